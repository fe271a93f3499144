#!/bin/bash
# round 3: mixed batch (config 3), launch-unit pieces x pipelined / joined
mkdir -p gpurun_out
for ch in 1 2 3 4 6 8; do for fl in 4 0; do
  SONDE_MIX_CHUNKS=$ch python bench.py --mix --channels 4096 --tiles 24 --no-cpu --flags $fl --steps 200 --warmup 50 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('chunks $ch flags $fl ms', d['ms_per_step'], 'frac', d['roofline']['step_frac'], d['kernel_ms'].get('per_class'))"
done; done
