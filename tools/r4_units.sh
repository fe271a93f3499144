#!/bin/bash
# experiment: one-class pipelined batches cut into SONDE_UNITS launch units on their own streams
for shape in "1250 24" "1280 96" "1024 96" "8192 24" "1100 96"; do
  set -- $shape
  for u in 0 1 2 3 4 6 8; do
    if [ $u = 0 ]; then fl=0; un=1; else fl=4; un=$u; fi
    echo -n "channels=$1 tiles=$2 flags=$fl units=$un: "
    SONDE_UNITS=$un python bench.py --no-cpu --no-others --channels $1 --tiles $2 --flags $fl --steps 100 --warmup 20 2>/dev/null | grep -o '"ms_per_step[^,]*,\|"step_frac"[^,]*,' | tr '\n' ' '; echo
  done
done
