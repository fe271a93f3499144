#!/bin/bash
# round-2 quick GPU check: GPU tests, the bench line, and the rocprofv3 kernel trace of the bench command
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
tail -5 gpurun_out/pytest_gpu.log
python bench.py ${BENCH_ARGS:---no-cpu} > gpurun_out/bench.json 2> gpurun_out/bench.err
cat gpurun_out/bench.json
cd /tmp && rm -rf /tmp/p_trace
rocprofv3 --kernel-trace --stats -d /tmp/p_trace -o trace -- python $R/bench.py --steps 200 --warmup 50 --no-cpu > /tmp/b1.log 2>&1
cd $R
python tools/rocprof_summary.py $(find /tmp/p_trace -name '*.db') > gpurun_out/rocprof.csv 2> gpurun_out/rocprof.err
cat gpurun_out/rocprof.csv
