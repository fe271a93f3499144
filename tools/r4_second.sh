#!/bin/bash
# round 4, second GPU call: the whole -m gpu suite on the AFC SPEC, the default bench line, the yardstick table (HIP path), a fuzz campaign
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r4_d_pytest.log 2>&1
echo "pytest rc=$?"; tail -6 gpurun_out/r4_d_pytest.log
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r4_d_bench.json 2> gpurun_out/r4_d_bench.err
echo "bench rc=$?"; tail -3 gpurun_out/r4_d_bench.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r4_d_bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_frac'], d['roofline']['kernel_us_trace'])
for k,v in d['other_configs'].items(): print(k, v['ms_per_step'], v.get('step_frac'))
print('low_snr', d['low_snr']['ms_per_step'], 'contig', d['contiguous_layout']['ms_per_step'])
print(d['cpu_baseline']['value'], d['cpu_baseline']['conventional'])
P
( time python tools/yardstick_study.py --gpu ) > gpurun_out/r4_d_yardstick.md 2> gpurun_out/r4_d_yardstick.err
tail -3 gpurun_out/r4_d_yardstick.err
( time python tools/fuzz_campaign.py 400 ) > gpurun_out/r4_d_fuzz.log 2>&1
tail -4 gpurun_out/r4_d_fuzz.log
