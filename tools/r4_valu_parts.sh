#!/bin/bash
# What each part of kernel A's per-tile work costs: the (complex64, 4:1) class with one part switched off at a time (results are then WRONG:
# counters and times only).  var0 as shipped; var1 no sync search (K4, round wave 3); var2 no in-loop clean-frame decoder (round wave 2);
# var3 no bit-ring append (round wave 1); var4 no mid-symbol FIR (the Gardner term's second interpolation); var5 no AFC rotation.
export TMPDIR=/tmp; R=$PWD; cd /tmp && rm -rf /tmp/vp_*
for v in 0 1 2 3 4 5; do
  export SONDE_MI355_LIB=$R/ab/lib_var$v.so
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE -d /tmp/vp_$v -o c -- python $R/bench.py --steps 5 --warmup 2 --ramp-ms 60 --no-cpu --no-others > /tmp/vp$v.log 2>&1
done
cd $R
for v in 0 1 2 3 4 5; do echo "== var$v"; python tools/rocprof_summary.py $(find /tmp/vp_$v -name '*.db') 2>/dev/null | grep demod | sed 's/.*>,//' ; done
