#!/bin/bash
# A/B timing of library variants inside ONE gpurun call (same box, interleaved): tools/ab_bench.sh ab/lib_a.so ab/lib_b.so ...
REPS=${REPS:-3}
for rep in $(seq 1 $REPS); do
  for lib in "$@"; do
    echo -n "$lib rep$rep: "
    SONDE_MI355_LIB=$PWD/$lib python bench.py --steps 20 --warmup 3 --no-cpu 2>&1 | tail -1 | grep -o '"kernel_ms[^}]*}'
  done
done
