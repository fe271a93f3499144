#!/bin/bash
# the round's evidence run (one box; TAG=r6 by default): smoke, the GPU suite, the default bench line and the line with the driver's arguments, the rocprofv3
# kernel trace + stats of the headline command, separate --pmc passes (HBM traffic, SQ, LDS), the wideband traces and counters, the
# yardstick and sensitivity tables from the HIP path.  Results land in gpurun_out/ (copy to profiles/).
export TMPDIR=/tmp
R=$PWD
TAG=${TAG:-r6}
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > gpurun_out/${TAG}_smoke.log 2>&1
( time timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/${TAG}_pytest_gpu.log 2>&1
tail -3 gpurun_out/${TAG}_smoke.log gpurun_out/${TAG}_pytest_gpu.log
( time python bench.py ) > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
( time python bench.py --steps 20 --warmup 5 ) > gpurun_out/${TAG}_bench_driver_args.json 2>> gpurun_out/${TAG}_bench.err
grep real gpurun_out/${TAG}_bench.err
cd /tmp && rm -rf /tmp/p_*
rocprofv3 --kernel-trace --stats -d /tmp/p_trace -o trace -- python $R/bench.py --steps 200 --warmup 50 --no-cpu --no-others > /tmp/b1.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d /tmp/p_fetch -o fetch -- python $R/bench.py --steps 5 --warmup 2 --ramp-ms 60 --no-cpu --no-others > /tmp/b2.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/p_write -o write -- python $R/bench.py --steps 5 --warmup 2 --ramp-ms 60 --no-cpu --no-others > /tmp/b3.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d /tmp/p_sq -o sq -- python $R/bench.py --steps 5 --warmup 2 --ramp-ms 60 --no-cpu --no-others > /tmp/b4.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU -d /tmp/p_lds -o lds -- python $R/bench.py --steps 5 --warmup 2 --ramp-ms 60 --no-cpu --no-others > /tmp/b5.log 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/p_wb -o trace -- python $R/bench.py --wideband --steps 100 --warmup 20 --no-cpu > /tmp/w1.log 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/p_wb8 -o trace -- python $R/bench.py --wideband --wb-streams 8 --steps 100 --warmup 20 --no-cpu > /tmp/w8.log 2>&1
cd $R
python tools/rocprof_summary.py $(find /tmp/p_trace -name '*.db') > gpurun_out/${TAG}_rocprof.csv 2> gpurun_out/rocprof.err
python tools/rocprof_summary.py $(find /tmp/p_fetch /tmp/p_write /tmp/p_sq /tmp/p_lds -name '*.db') > gpurun_out/${TAG}_counters.csv 2>> gpurun_out/rocprof.err
python tools/rocprof_summary.py $(find /tmp/p_wb -name '*.db') $(find /tmp/p_wb8 -name '*.db') > gpurun_out/${TAG}_wb_rocprof.csv 2>> gpurun_out/rocprof.err
cp $(find /tmp/p_trace -name '*kernel_stats.csv' | head -1) gpurun_out/${TAG}_kernel_stats.csv 2>/dev/null
TAG=${TAG} bash tools/wb_counters.sh > /dev/null 2>&1
python tools/yardstick_study.py --gpu > gpurun_out/${TAG}_yardstick.md 2> gpurun_out/${TAG}_yardstick.err
python tools/sensitivity.py > gpurun_out/${TAG}_sensitivity.md 2> gpurun_out/${TAG}_sensitivity.err
cat gpurun_out/${TAG}_bench_driver_args.json | head -c 3000; echo; cat gpurun_out/${TAG}_rocprof.csv; grep -v read_probe gpurun_out/${TAG}_counters.csv; cat gpurun_out/${TAG}_wb_rocprof.csv
