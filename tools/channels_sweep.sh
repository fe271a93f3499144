#!/bin/bash
# throughput vs channels per launch (196608 samples each): how the launch ramp/tail and the framer amortise
for c in 512 1024 2048 3072 4096; do
  echo -n "channels=$c: "
  python bench.py --no-cpu --channels $c --steps 100 --warmup 20 | grep -o '"value[^,]*,\|"kernel_ms[^}]*}\|"frac"[^,]*,' | tr '\n' ' '; echo
done
