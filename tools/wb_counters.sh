#!/bin/bash
# SQ / TCC counters of the filter-bank and decoder kernels of the wideband configuration (8 streams, 1 block per submit)
export TMPDIR=/tmp
R=$PWD
out=gpurun_out/${TAG:-r6}_wb_counters.csv
: > $out
i=0
for pass in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); d=/tmp/p_wbc_$i
  cd /tmp && rm -rf $d
  rocprofv3 --pmc $pass -d $d -o pmc -- python $R/bench.py --wideband --wb-streams 8 --steps 5 --warmup 2 --ramp-ms 60 --no-cpu > /tmp/pmc.log 2>&1 || tail -5 /tmp/pmc.log
  cd $R
  python tools/rocprof_summary.py $(find $d -name '*.db') | grep -v "read_probe\|^db," >> $out
done
cat $out
