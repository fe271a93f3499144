#!/bin/bash
# every float class alone -- step times (4096 channels x 24 tiles and 1024 x 96) and SQ counters (LDS bank conflicts, VALU, waits)
# usage (GPU box): TAG=r6 tools/class_counters.sh  -> gpurun_out/${TAG}_class_counters.csv, ${TAG}_classes.txt
export TMPDIR=/tmp
R=$PWD
TAG=${TAG:-r6}
mkdir -p gpurun_out
: > gpurun_out/${TAG}_classes.txt
for t in 0 1 3; do
  for shape in "4096 24" "1024 96"; do
    set -- $shape
    if [ $t = 0 ]; then A=""; else A="--sonde-type $t"; fi
    python bench.py $A --channels $1 --tiles $2 --no-cpu --no-others --steps 100 --warmup 20 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('type $t $1 x $2', d['ms_per_step'], d['roofline']['step_frac'], d['kernel_ms'].get('demod'))" | tee -a gpurun_out/${TAG}_classes.txt
  done
done
for fl in 1; do for t in 0 3; do
  python bench.py --sonde-type $t --flags $fl --channels 1024 --tiles 96 --no-cpu --no-others --steps 100 --warmup 20 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('type $t WIDE 1024 x 96', d['ms_per_step'], d['roofline']['step_frac'])" | tee -a gpurun_out/${TAG}_classes.txt
done; done
out=gpurun_out/${TAG}_class_counters.csv
: > $out
for cfg in "3 0" "0 0" "3 1" "0 1"; do
  set -- $cfg
  i=0
  for pass in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD"; do
    i=$((i+1)); d=/tmp/p_t$1_$2_$i
    cd /tmp && rm -rf $d
    rocprofv3 --pmc $pass -d $d -o pmc -- python $R/bench.py --sonde-type $1 --flags $2 --blocks 1 --steps 5 --warmup 2 --ramp-ms 60 --no-cpu --no-others > /tmp/pmc.log 2>&1 || tail -5 /tmp/pmc.log
    cd $R
    python tools/rocprof_summary.py $(find $d -name '*.db') | grep -v "read_probe\|^db," | sed "s/^/type$1_flags$2,/" >> $out
  done
done
cat $out
