#!/bin/bash
# 16-bit IQ input: parity tests, then interleaved A/B of library builds: tools/r4_iq16.sh A B (names of ab/lib_<name>.so)
timeout 900 python -m pytest tests/test_gpu_iq16.py tests/test_node.py -m gpu -x -q 2>&1 | tail -3
VARS="$@"; for rep in 1 2 3; do for v in $VARS; do
  export SONDE_MI355_LIB=$PWD/ab/lib_$v.so
  for shape in "1024 96" "4096 96" "8192 24"; do set -- $shape
    python bench.py --iq16 --no-cpu --channels $1 --tiles $2 --steps 150 --warmup 40 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', $1, $2, d['ms_per_step'], d['value'], d['step_frac'])"
  done
done; done
