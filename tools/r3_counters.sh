#!/bin/bash
# round 3: SQ counters of every demod class alone (1024 channels x 96 tiles): VALU issue share, LDS, wait buckets
# usage (GPU box): TAG=r3 tools/r3_counters.sh   -> gpurun_out/${TAG}_class_counters.csv
export TMPDIR=/tmp
R=$PWD
TAG=${TAG:-r3}
mkdir -p gpurun_out
out=gpurun_out/${TAG}_class_counters.csv
: > $out
for t in 3 1 0; do
  i=0
  for pass in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD"; do
    i=$((i+1))
    d=/tmp/p_t${t}_$i
    cd /tmp && rm -rf $d
    if [ $t = 0 ]; then A="--blocks 1"; else A="--sonde-type $t"; fi
    rocprofv3 --pmc $pass -d $d -o pmc -- python $R/bench.py $A --steps 5 --warmup 2 --ramp-ms 60 --no-cpu --no-others > /tmp/pmc.log 2>&1 || tail -5 /tmp/pmc.log
    cd $R
    python tools/rocprof_summary.py $(find $d -name '*.db') | grep -v "read_probe\|^db," >> $out
  done
done
cat $out
