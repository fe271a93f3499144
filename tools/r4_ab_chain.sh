#!/bin/bash
# A/B: bit-ring append on round wave 1 (w1), the same with a third register set at three workgroups per CU (w1s3), against the committed kernel (base)
mkdir -p gpurun_out
for v in w1 w1s3; do
  SONDE_MI355_LIB=$PWD/ab/lib_$v.so timeout 600 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_parity.py tests/test_gpu_other_sondes.py -m gpu -q -x 2>&1 | tail -2
done
for rep in 1 2; do for v in base w1 w1s3; do
  export SONDE_MI355_LIB=$PWD/ab/lib_$v.so
  for shape in "1024 96" "768 96" "512 96" "4096 96" "8192 24"; do set -- $shape
    python bench.py --no-cpu --no-others --channels $1 --tiles $2 --steps 100 --warmup 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', $1, $2, d['ms_per_step'], d['roofline']['step_frac'])"
  done
done; done 2>&1 | tee gpurun_out/r4_o_ab_chain.txt
