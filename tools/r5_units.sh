#!/bin/bash
# round 5: the part-filled-generation shapes at the DEFAULT flags (launch units joined one submit late), with SONDE_FLAG_JOIN (16: every
# submit joined: rounds 1-4's default) and SONDE_FLAG_PIPELINE (4) beside them
for shape in "1250 24" "1280 96" "1024 96" "8192 24" "1537 24"; do
  set -- $shape
  for fl in 0 16 4; do
    python bench.py --no-cpu --no-others --channels $1 --tiles $2 --flags $fl --steps 100 --warmup 20 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('rs41 $1 x $2 flags $fl', d['ms_per_step'], d['roofline']['step_frac'])"
  done
done
for fl in 0 16 4; do
  python bench.py --mix --channels 4096 --tiles 24 --no-cpu --flags $fl --steps 100 --warmup 20 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('mix 4096 x 24 flags $fl', d['ms_per_step'], d['roofline']['step_frac'])"
done
