#!/bin/bash
# round 6: time-sliced demod launches (SONDE_SEG = segments per channel; 1 = unsliced) at the default flags and late-joined
export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/r6_ab_seg.txt
: > $out
run() {  # label, env, args
  for rep in 1 2; do
    env $2 python bench.py $3 --no-cpu --no-others --steps 100 --warmup 20 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('$1 $2 rep$rep', d['ms_per_step'], d['roofline']['step_frac'], d['config'].get('launch_units'), d['config'].get('join'))" | tee -a $out
  done
}
for seg in 1 2 3 4; do
  run "rt1250x24 f0" "SONDE_SEG=$seg" "--channels 1250 --tiles 24"
  run "ch1280x96 f0" "SONDE_SEG=$seg" "--channels 1280 --tiles 96"
  run "rs41_4096x24 f0" "SONDE_SEG=$seg" "--channels 4096 --tiles 24"
  run "shard8192x24 f0" "SONDE_SEG=$seg" "--channels 8192 --tiles 24"
  run "mix4096 f0" "SONDE_SEG=$seg" "--mix --flags 0"
  run "mix4096 f32" "SONDE_SEG=$seg" "--mix --flags 32"
  run "m10_4096x24 f0" "SONDE_SEG=$seg" "--sonde-type 3 --channels 4096 --tiles 24"
done
for seg in 1 2 4 8; do
  run "headline1024x96 f0" "SONDE_SEG=$seg" "--channels 1024 --tiles 96"
  run "ch1280x96 f0" "SONDE_SEG=$seg" "--channels 1280 --tiles 96"
done
cat $out
