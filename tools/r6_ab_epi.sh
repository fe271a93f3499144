#!/bin/bash
# round 6: the fixed-length framers' frame decoders in the demod kernel's epilogue (SONDE_FIXED_EPI=1) against the stand-alone kernel (0):
# per-type and mixed step times at the default flags (joined at every submit) and late-joined.  usage (GPU box): tools/r6_ab_epi.sh
export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/r6_ab_epi.txt
: > $out
run() {  # label, env, args
  for rep in 1 2; do
    env $2 python bench.py $3 --no-cpu --no-others --steps 100 --warmup 20 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('$1 $2 rep$rep', d['ms_per_step'], d['roofline']['step_frac'], d['config'].get('launch_units'), d['config'].get('join'))" | tee -a $out
  done
}
for epi in 0 1; do
  run "dfm4096x24" "SONDE_FIXED_EPI=$epi" "--sonde-type 1 --channels 4096 --tiles 24"
  run "m10_4096x24" "SONDE_FIXED_EPI=$epi" "--sonde-type 3 --channels 4096 --tiles 24"
  run "mix4096 flags0" "SONDE_FIXED_EPI=$epi" "--mix --flags 0"
  run "mix4096 flags32" "SONDE_FIXED_EPI=$epi" "--mix --flags 32"
done
run "rs41_4096x24" "X=1" "--channels 4096 --tiles 24"
run "rt1250 flags0" "X=1" "--channels 1250 --tiles 24"
run "rt1250 flags32" "X=1" "--channels 1250 --tiles 24 --flags 32"
cat $out
