#!/bin/bash
# round 4, first GPU call: the new same-shape parity tests, the channel-count sweep (baseline for item 3), the default bench line
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp; cd - >/dev/null
( time timeout 900 python -m pytest tests/test_gpu_bench_shapes.py "tests/test_channelizer.py::test_fused_channelizer_frames_equal_oracle" -m gpu -x -q ) > gpurun_out/r4_a_newtests.log 2>&1
echo "newtests rc=$?"
tail -5 gpurun_out/r4_a_newtests.log
( bash tools/channels_sweep.sh 96; bash tools/channels_sweep.sh 24 ) > gpurun_out/r4_a_sweep.txt 2>&1
cat gpurun_out/r4_a_sweep.txt
( time timeout 600 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r4_a_bench.json 2> gpurun_out/r4_a_bench.err
echo "bench rc=$?"; tail -3 gpurun_out/r4_a_bench.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r4_a_bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['roofline'])
for k,v in d['other_configs'].items(): print(k, v['ms_per_step'], v.get('step_frac'))
print('low_snr', d['low_snr']['ms_per_step'])
P
