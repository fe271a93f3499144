#!/bin/bash
# configs 3 (mixed) and 4 (wideband): bench line + rocprofv3 kernel trace each -> gpurun_out/${TAG}_{mix,wb}_*
export TMPDIR=/tmp
R=$PWD
TAG=${TAG:-r2}
mkdir -p gpurun_out
python bench.py --mix --channels 4096 --tiles 24 --no-cpu > gpurun_out/${TAG}_mix_bench.json 2> gpurun_out/mix.err
python bench.py --mix --no-cpu > gpurun_out/${TAG}_mix96_bench.json 2>> gpurun_out/mix.err
python bench.py --channels 8192 --tiles 24 --no-cpu > gpurun_out/${TAG}_c5shard_bench.json 2>> gpurun_out/mix.err     # config 5: one GPU's shard, T = 1 s
python bench.py --channels 4096 --tiles 96 --steps 100 --warmup 20 --no-cpu > gpurun_out/${TAG}_4096x96_bench.json 2>> gpurun_out/mix.err
python bench.py --wideband --steps 100 --warmup 20 --no-cpu > gpurun_out/${TAG}_wb_bench.json 2> gpurun_out/wb.err
python bench.py --wideband --wb-streams 8 --steps 50 --warmup 10 --no-cpu > gpurun_out/${TAG}_wb8_bench.json 2>> gpurun_out/wb.err
cd /tmp && rm -rf /tmp/p_mix /tmp/p_wb
rocprofv3 --kernel-trace --stats -d /tmp/p_mix -o trace -- python $R/bench.py --mix --channels 4096 --tiles 24 --steps 100 --warmup 20 --no-cpu > /tmp/m1.log 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/p_wb -o trace -- python $R/bench.py --wideband --steps 100 --warmup 20 --no-cpu > /tmp/w1.log 2>&1
cd $R
python tools/rocprof_summary.py $(find /tmp/p_mix -name '*.db') > gpurun_out/${TAG}_mix_rocprof.csv 2> gpurun_out/rocprof.err
python tools/rocprof_summary.py $(find /tmp/p_wb -name '*.db') > gpurun_out/${TAG}_wb_rocprof.csv 2>> gpurun_out/rocprof.err
tail -2 gpurun_out/mix.err gpurun_out/wb.err
cat gpurun_out/${TAG}_mix_bench.json gpurun_out/${TAG}_wb_bench.json gpurun_out/${TAG}_wb8_bench.json gpurun_out/${TAG}_mix_rocprof.csv gpurun_out/${TAG}_wb_rocprof.csv
