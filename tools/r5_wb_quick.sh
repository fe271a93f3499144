#!/bin/bash
# round-5 quick wideband check: channelizer parity tests, wideband bench lines (1 and 8 streams), kernel trace of the 8-stream step
export TMPDIR=/tmp
R=$PWD
TAG=${TAG:-r5_wbq}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_channelizer.py -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; tail -3 gpurun_out/${TAG}_pytest.log
python bench.py --wideband --steps 100 --warmup 20 --no-cpu > gpurun_out/${TAG}_wb1.json 2> gpurun_out/${TAG}.err
python bench.py --wideband --wb-streams 8 --steps 100 --warmup 20 --no-cpu > gpurun_out/${TAG}_wb8.json 2>> gpurun_out/${TAG}.err
cd /tmp && rm -rf /tmp/p_wb8
rocprofv3 --kernel-trace --stats -d /tmp/p_wb8 -o trace -- python $R/bench.py --wideband --wb-streams 8 --steps 100 --warmup 20 --no-cpu > /tmp/w8.log 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/p_wb1 -o trace -- python $R/bench.py --wideband --steps 100 --warmup 20 --no-cpu > /tmp/w1.log 2>&1
cd $R
python tools/rocprof_summary.py $(find /tmp/p_wb8 -name '*.db') $(find /tmp/p_wb1 -name '*.db') > gpurun_out/${TAG}_rocprof.csv 2>> gpurun_out/${TAG}.err
cat gpurun_out/${TAG}_rocprof.csv
for f in wb1 wb8; do python -c "
import json,sys; d=json.load(open('gpurun_out/${TAG}_'+'$f'+'.json')); print('$f', d['ms_per_step'], d['value'], d.get('roofline',{}).get('step_frac'), d['kernel_ms'])"; done
