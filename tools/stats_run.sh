#!/bin/bash
# rocprofv3's own --stats summary (CSV) of the default bench command -> gpurun_out/kernel_stats.csv
export TMPDIR=/tmp
R=$PWD
cd /tmp && rm -rf /tmp/p_stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -o st -- python $R/bench.py --no-cpu > /tmp/st.log 2>&1
cd $R
mkdir -p gpurun_out
find /tmp/p_stats -name '*stats*.csv' | head -5
cp $(find /tmp/p_stats -name '*kernel_stats.csv' | head -1) gpurun_out/kernel_stats.csv
cat gpurun_out/kernel_stats.csv
tail -1 /tmp/st.log | cut -c1-300
