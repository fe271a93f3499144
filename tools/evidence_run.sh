set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > gpurun_out/smoke.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/p_trace -o trace -- python $R/bench.py --steps 200 --warmup 50 --no-cpu > /tmp/b1.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d /tmp/p_fetch -o fetch -- python $R/bench.py --steps 5 --warmup 2 --ramp-ms 60 --no-cpu > /tmp/b2.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/p_write -o write -- python $R/bench.py --steps 5 --warmup 2 --ramp-ms 60 --no-cpu > /tmp/b3.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d /tmp/p_sq -o sq -- python $R/bench.py --steps 5 --warmup 2 --ramp-ms 60 --no-cpu > /tmp/b4.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU -d /tmp/p_lds -o lds -- python $R/bench.py --steps 5 --warmup 2 --ramp-ms 60 --no-cpu > /tmp/b5.log 2>&1
cd $R
python tools/rocprof_summary.py $(find /tmp/p_trace /tmp/p_fetch /tmp/p_write /tmp/p_sq /tmp/p_lds -name '*.db') > gpurun_out/rocprof.csv 2> gpurun_out/rocprof.err
tail -3 /tmp/b1.log /tmp/b4.log > gpurun_out/blogs.txt
cat gpurun_out/smoke.log | tail -2; tail -2 gpurun_out/pytest_gpu.log; cat gpurun_out/bench.json; cat gpurun_out/rocprof.csv
