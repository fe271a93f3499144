#!/bin/bash
# 16-bit IQ input: parity tests, then gen (8-byte loads, the float path's lane mapping) against d4 (16-byte loads = one decimated sample each, integer sums)
timeout 900 python -m pytest tests/test_gpu_iq16.py tests/test_node.py -m gpu -x -q 2>&1 | tail -5
for rep in 1 2; do for v in iq16gen iq16d4; do
  export SONDE_MI355_LIB=$PWD/ab/lib_$v.so
  for shape in "1024 96" "4096 96" "8192 24" "1250 24"; do set -- $shape
    python bench.py --iq16 --no-cpu --channels $1 --tiles $2 --steps 150 --warmup 40 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', $1, $2, d['ms_per_step'], d['value'], d['step_frac'])"
  done
  python bench.py --iq16 --mix --channels 4096 --tiles 24 --flags 4 --steps 100 --warmup 20 --no-cpu 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v mix4096', d['ms_per_step'], d['value'], d['step_frac'])"
done; done
