#!/bin/bash
# interleaved A/B on ONE box: rounds per pair of tiles (tm2: a quarter less VALU work; results differ from the SPEC: speed only) against the shipped kernel (cur)
for rep in 1 2 3; do for v in cur tm2; do
  export SONDE_MI355_LIB=$PWD/ab/lib_$v.so
  for shape in "1024 96" "4096 96" "8192 24"; do set -- $shape
    python bench.py --no-cpu --no-others --channels $1 --tiles $2 --steps 150 --warmup 40 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', $1, $2, d['ms_per_step'], d['roofline']['step_frac'])"
  done
done; done 2>&1 | tee gpurun_out/r4_s_ab_tm2.txt
