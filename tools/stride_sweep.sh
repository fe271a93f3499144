#!/bin/bash
# channel stride of the IQ block against step time (HBM page / channel effects): rows of 196 608 samples (1.5 MiB) and 49 152 (384 KiB)
run() { python bench.py "$@" --no-cpu --no-others 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['step_frac'])"; }
for rep in 1 2; do
for pad in 0 65536 327680 589824 69632; do echo -n "1024x96 pad $pad stride KiB $(( (196608+pad)*8/1024 )): "; run --stride-pad $pad --steps 100 --warmup 20; done
for pad in 0 16384 81920 212992; do echo -n "8192x24 pad $pad stride KiB $(( (49152+pad)*8/1024 )): "; run --channels 8192 --tiles 24 --stride-pad $pad --steps 60 --warmup 15; done
for pad in 0 16384; do echo -n "mix 4096x24 pad $pad: "; run --mix --channels 4096 --tiles 24 --stride-pad $pad --steps 100 --warmup 20; done
done
