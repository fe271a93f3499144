#!/bin/bash
# interleaved A/B on ONE box: head (conditional prefetch loads in the tile loop: the compiler waits for vmcnt(3..0)) against uncond
# (steady-state loads unconditional: vmcnt(7..4))
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_shapes.py tests/test_gpu_iq16.py tests/test_gpu_fuzz_parity.py -m gpu -x -q 2>&1 | tail -3
for rep in 1 2 3; do for v in head uncond; do
  export SONDE_MI355_LIB=$PWD/ab/lib_$v.so
  for shape in "1024 96" "4096 96" "8192 24" "512 96"; do set -- $shape
    python bench.py --no-cpu --no-others --channels $1 --tiles $2 --steps 150 --warmup 40 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', $1, $2, d['ms_per_step'], d['roofline']['step_frac'])"
  done
  python bench.py --mix --channels 4096 --tiles 24 --flags 4 --steps 100 --warmup 20 --no-cpu 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[0]); print('$v mix4096', d['ms_per_step'], d['roofline']['step_frac'])"
  python bench.py --iq16 --no-cpu --channels 1024 --tiles 96 --steps 150 --warmup 40 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v cs16 1024 96', d['ms_per_step'], d['value'], d['step_frac'])"
  python bench.py --iq16 --no-cpu --channels 4096 --tiles 96 --steps 100 --warmup 20 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v cs16 4096 96', d['ms_per_step'], d['value'], d['step_frac'])"
done; done
