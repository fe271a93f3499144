#!/bin/bash
# kernel trace of the default bench (20 steps) -> per-kernel averages on stdout
export TMPDIR=/tmp
R=$PWD
cd /tmp && rm -rf /tmp/p_trace
rocprofv3 --kernel-trace --stats -d /tmp/p_trace -o trace -- python $R/bench.py --steps 200 --warmup 50 --no-cpu > /tmp/b1.log 2>&1
cd $R
python tools/rocprof_summary.py $(find /tmp/p_trace -name '*.db')
tail -1 /tmp/b1.log | grep -o '"kernel_ms[^}]*}'
