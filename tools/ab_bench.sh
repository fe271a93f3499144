#!/bin/bash
# A/B timing of library variants inside ONE gpurun call (same box, interleaved): tools/ab_bench.sh ab/lib_a.so ab/lib_b.so ...
REPS=${REPS:-3}
for rep in $(seq 1 $REPS); do
  for lib in "$@"; do
    echo -n "$lib rep$rep: "
    SONDE_MI355_LIB=$PWD/$lib python bench.py --steps ${STEPS:-100} --warmup 20 --no-cpu 2>&1 | tail -1 | python -c "
import sys, json
try:
    j = json.loads(sys.stdin.read())
    print('ms_per_step', j['ms_per_step'], 'kernel_ms', j['kernel_ms'], 'step_frac', j['roofline'].get('step_frac'), 'frames', j.get('frames_per_step_steady'))
except Exception as e:
    print('FAILED', e)
"
  done
done
