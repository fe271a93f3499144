#!/bin/bash
# round 4, fifth GPU call: node host + scatter_rows + host-path tests, the default bench line with the e2e entry, sensitivity table
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_node.py tests/test_shard_native.py tests/test_gpu_bench_shapes.py::test_host_path_at_the_target_shape tests/test_gpu_bench_ranks.py tests/test_gpu_b1.py tests/test_channelizer.py -m gpu -q ) > gpurun_out/r4_h_tests.log 2>&1
echo "tests rc=$?"; tail -12 gpurun_out/r4_h_tests.log
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r4_h_bench.json 2> gpurun_out/r4_h_bench.err
echo "bench rc=$?"; tail -2 gpurun_out/r4_h_bench.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r4_h_bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_frac'])
print(d['other_configs']['rt1250_host_e2e'])
P
python tools/sensitivity.py > gpurun_out/r4_sensitivity.md 2>/dev/null; cat gpurun_out/r4_sensitivity.md
