#!/bin/bash
export TMPDIR=/tmp
R=$PWD
TAG=${TAG:-r2}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_channelizer.py -m gpu -x -q 2>&1 | tail -3
python bench.py --wideband --steps 100 --warmup 20 --no-cpu > gpurun_out/${TAG}_wb_bench.json 2> gpurun_out/wb.err
python bench.py --wideband --wb-streams 8 --steps 50 --warmup 10 --no-cpu > gpurun_out/${TAG}_wb8_bench.json 2>> gpurun_out/wb.err
cd /tmp && rm -rf /tmp/p_wb
rocprofv3 --kernel-trace --stats -d /tmp/p_wb -o trace -- python $R/bench.py --wideband --steps 100 --warmup 20 --no-cpu > /tmp/w1.log 2>&1
cd $R
python tools/rocprof_summary.py $(find /tmp/p_wb -name '*.db') > gpurun_out/${TAG}_wb_rocprof.csv 2>> gpurun_out/rocprof.err
tail -n 3 gpurun_out/wb.err
cat gpurun_out/${TAG}_wb_bench.json gpurun_out/${TAG}_wb8_bench.json gpurun_out/${TAG}_wb_rocprof.csv
