#!/bin/bash
# interleaved A/B on ONE box: cur (shipped) | join (prologue loads requested together, utype) | joinpl (the same + 13 kernel-argument dwords preloaded into SGPRs)
for v in join joinpl; do
  SONDE_MI355_LIB=$PWD/ab/lib_$v.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_shapes.py tests/test_channelizer.py -m gpu -x -q 2>&1 | tail -3 | sed "s/^/$v: /"
done 2>&1 | tee gpurun_out/r4_u_parity.txt
for rep in 1 2 3; do for v in cur join joinpl; do
  export SONDE_MI355_LIB=$PWD/ab/lib_$v.so
  for shape in "1024 96" "4096 96" "8192 24" "1250 24"; do set -- $shape
    python bench.py --no-cpu --no-others --channels $1 --tiles $2 --steps 150 --warmup 40 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', $1, $2, d['ms_per_step'], d['roofline']['step_frac'])"
  done
  python bench.py --mix --channels 4096 --tiles 24 --flags 4 --steps 100 --warmup 20 --no-cpu 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[0]); print('$v mix4096', d['ms_per_step'], d['roofline']['step_frac'])"
  for S in 1 8; do python bench.py --wideband --wb-streams $S --steps 60 --warmup 20 --no-cpu 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v wb S=$S', d['ms_per_step'], d['value'], d['kernel_ms'])"; done
done; done 2>&1 | tee gpurun_out/r4_u_ab_join.txt
