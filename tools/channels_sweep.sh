#!/bin/bash
# throughput vs channels per launch: how the launch ramp/tail, part-filled last generations and the framer amortise
# usage: channels_sweep.sh [tiles] (default 96 = 196608 samples per channel)
T=${1:-96}
for c in 512 1024 1100 1250 1280 1536 2048 2500 3072 4096; do
  echo -n "channels=$c tiles=$T: "
  python bench.py --no-cpu --no-others --channels $c --tiles $T --steps 100 --warmup 20 | grep -o '"ms_per_step[^,]*,\|"step_frac"[^,]*,' | tr '\n' ' '; echo
done
