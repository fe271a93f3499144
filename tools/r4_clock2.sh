#!/bin/bash
# clocks and package power under the float headline, the 16-bit headline, and 512 float channels (two workgroups per CU: chain-bound)
sample() { for i in $(seq 1 $1); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed 's/.*: *//' | tr '\n' ' '; echo; sleep 0.25; done; }
for cfg in "" "--iq16" "--channels 512" "--channels 4096"; do
  echo "== bench.py $cfg (steps 12000)"
  ( sleep 2.0; sample 8 ) > /tmp/smi.log &
  python bench.py --no-cpu --no-others $cfg --steps 12000 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'])"
  wait; cat /tmp/smi.log
done
