#!/bin/bash
# issue-side counters of the 16-bit IQ launch (1024 x 96 tiles) beside the float launch: is it VALU-bound?
export TMPDIR=/tmp; R=$PWD; cd /tmp && rm -rf /tmp/q_*
for v in f32 i16; do
  X=""; [ $v = i16 ] && X="--iq16"
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d /tmp/q_sq_$v -o sq -- python $R/bench.py $X --steps 5 --warmup 2 --ramp-ms 60 --no-cpu --no-others > /tmp/q1.log 2>&1
  rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU -d /tmp/q_lds_$v -o lds -- python $R/bench.py $X --steps 5 --warmup 2 --ramp-ms 60 --no-cpu --no-others > /tmp/q2.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM -d /tmp/q_any_$v -o any -- python $R/bench.py $X --steps 5 --warmup 2 --ramp-ms 60 --no-cpu --no-others > /tmp/q3.log 2>&1
  rocprofv3 --kernel-trace --stats -d /tmp/q_tr_$v -o trace -- python $R/bench.py $X --steps 100 --warmup 20 --no-cpu --no-others > /tmp/q4.log 2>&1
done
cd $R
for v in f32 i16; do echo "== $v"; python tools/rocprof_summary.py $(find /tmp/q_sq_$v /tmp/q_lds_$v /tmp/q_any_$v -name '*.db') 2>/dev/null | grep -v read_probe; python tools/rocprof_summary.py $(find /tmp/q_tr_$v -name '*.db') 2>/dev/null | grep demod; done > gpurun_out/r4_y_iq16_counters.csv
cat gpurun_out/r4_y_iq16_counters.csv
