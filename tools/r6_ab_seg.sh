#!/bin/bash
# round 6: time-sliced demod launches (bench.py --time-slices n = SondeBatchConfig.time_slices: segments per channel and submit; 1 = unsliced,
# 0 = the library's choice) at the default flags.  usage (GPU box): tools/r6_ab_seg.sh -> gpurun_out/r6_ab_seg.txt
export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/r6_ab_seg.txt
: > $out
run() {  # label, args
  for rep in 1 2; do
    python bench.py $2 --no-cpu --no-others --steps 100 --warmup 20 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('$1 rep$rep', d['ms_per_step'], d['roofline']['step_frac'], d['config'].get('launch_units'), d['config'].get('join'))" | tee -a $out
  done
}
for seg in 1 2 4 0; do
  run "rt1250x24 slices=$seg" "--channels 1250 --tiles 24 --time-slices $seg"
  run "ch1280x96 slices=$seg" "--channels 1280 --tiles 96 --time-slices $seg"
  run "rs41_4096x24 slices=$seg" "--channels 4096 --tiles 24 --time-slices $seg"
  run "headline1024x96 slices=$seg" "--channels 1024 --tiles 96 --time-slices $seg"
done
cat $out
