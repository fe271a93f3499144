#!/bin/bash
# round-2 evidence run: GPU tests, smoke, the full bench line (with the CPU baseline), the rocprofv3 kernel trace of the
# bench command, and separate --pmc passes (HBM traffic, SQ, LDS).  Results land in gpurun_out/ (copy to profiles/).
export TMPDIR=/tmp
R=$PWD
TAG=${TAG:-r2}
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > gpurun_out/smoke.log 2>&1
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
tail -3 gpurun_out/smoke.log gpurun_out/pytest_gpu.log
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/${TAG}_bench_driver_args.json 2>> gpurun_out/bench.err
cd /tmp && rm -rf /tmp/p_*
rocprofv3 --kernel-trace --stats -d /tmp/p_trace -o trace -- python $R/bench.py --steps 200 --warmup 50 --no-cpu > /tmp/b1.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d /tmp/p_fetch -o fetch -- python $R/bench.py --steps 5 --warmup 2 --ramp-ms 60 --no-cpu > /tmp/b2.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/p_write -o write -- python $R/bench.py --steps 5 --warmup 2 --ramp-ms 60 --no-cpu > /tmp/b3.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d /tmp/p_sq -o sq -- python $R/bench.py --steps 5 --warmup 2 --ramp-ms 60 --no-cpu > /tmp/b4.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU -d /tmp/p_lds -o lds -- python $R/bench.py --steps 5 --warmup 2 --ramp-ms 60 --no-cpu > /tmp/b5.log 2>&1
cd $R
python tools/rocprof_summary.py $(find /tmp/p_trace -name '*.db') > gpurun_out/${TAG}_rocprof.csv 2> gpurun_out/rocprof.err
python tools/rocprof_summary.py $(find /tmp/p_fetch /tmp/p_write /tmp/p_sq /tmp/p_lds -name '*.db') > gpurun_out/${TAG}_counters.csv 2>> gpurun_out/rocprof.err
cp $(find /tmp/p_trace -name '*kernel_stats.csv' | head -1) gpurun_out/${TAG}_kernel_stats.csv 2>/dev/null
cat gpurun_out/${TAG}_bench.json gpurun_out/${TAG}_bench_driver_args.json gpurun_out/${TAG}_rocprof.csv; grep -v read_probe gpurun_out/${TAG}_counters.csv
