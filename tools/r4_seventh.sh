#!/bin/bash
# probe: rounds per pair of tiles ((IQ, 4, 8) class): parity at the full-size shapes + speed
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_bench_shapes.py -m gpu -q -x ) > gpurun_out/r4_j_tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r4_j_tests.log
for i in 1 2; do
python bench.py --no-cpu --no-others --steps 200 --warmup 50 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', d['ms_per_step'], d['roofline']['step_frac'], d['kernel_ms']['demod_hip_events_raw'])"
python bench.py --no-cpu --no-others --channels 4096 --tiles 96 --steps 60 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('4096x96', d['ms_per_step'], d['roofline']['step_frac'])"
python bench.py --no-cpu --no-others --channels 8192 --tiles 24 --steps 60 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('8192x24', d['ms_per_step'], d['roofline']['step_frac'])"
python bench.py --no-cpu --no-others --ebn0 9 --steps 100 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('9dB', d['ms_per_step'], d['roofline']['step_frac'])"
done
