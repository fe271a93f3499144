#!/bin/bash
# interleaved A/B on ONE box: head (two raw register sets, re-requested after the tile's arithmetic) against oneset (complex64 4:1 class: one raw
# set, reduced to partial sums on arrival and re-requested there)
SONDE_MI355_LIB=$PWD/ab/lib_oneset.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_shapes.py tests/test_gpu_fuzz_parity.py tests/test_gpu_full_size.py -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -2
for rep in 1 2 3; do for v in head oneset; do
  export SONDE_MI355_LIB=$PWD/ab/lib_$v.so
  for shape in "1024 96" "4096 96" "8192 24" "1250 24" "512 96"; do set -- $shape
    python bench.py --no-cpu --no-others --channels $1 --tiles $2 --steps 150 --warmup 40 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', $1, $2, d['ms_per_step'], d['roofline']['step_frac'])"
  done
  python bench.py --mix --channels 4096 --tiles 24 --flags 4 --steps 100 --warmup 20 --no-cpu 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[0]); print('$v mix4096', d['ms_per_step'], d['roofline']['step_frac'])"
  python bench.py --no-cpu --no-others --channels 1024 --tiles 96 --ebn0 9 --steps 100 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v 9dB 1024 96', d['ms_per_step'], d['roofline']['step_frac'])"
done; done
