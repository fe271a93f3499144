#!/bin/bash
# A/B timing of library variants inside ONE gpurun call (same box, interleaved): tools/ab_bench.sh ab/lib_a.so[:bench args] ...
REPS=${REPS:-3}
for rep in $(seq 1 $REPS); do
  for spec in "$@"; do
    lib=${spec%%:*}; extra=""; [[ "$spec" == *:* ]] && extra=${spec#*:}
    echo -n "$spec rep$rep: "
    SONDE_MI355_LIB=$PWD/$lib python bench.py --steps ${STEPS:-100} --warmup 20 --no-cpu $extra 2>&1 | tail -1 | python -c "
import sys, json
try:
    j = json.loads(sys.stdin.read())
    print('ms_per_step', j['ms_per_step'], 'kernel_ms', j['kernel_ms'].get('demod'), j['kernel_ms'].get('framer_fec'), 'step_frac', j['roofline'].get('step_frac'), 'frames', j.get('frames_per_step_steady'))
except Exception as e:
    print('FAILED', e)
"
  done
done
