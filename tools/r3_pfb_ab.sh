#!/bin/bash
# round 3: what each phase of the 8-step filter-bank kernel costs (A/B builds in ab/: make OUT=../../ab/lib_X.so EXTRA=-DPFB_AB_X)
for v in BASE NOSTAGE NOFOLD NOFFT NOSTORE; do
  if [ $v = BASE ]; then unset SONDE_MI355_LIB; else export SONDE_MI355_LIB=$PWD/ab/lib_$v.so; fi
  for S in 1 8; do python bench.py --wideband --wb-streams $S --steps 60 --warmup 20 --no-cpu 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v S', d['config']['streams_per_gpu'], d['ms_per_step'], d['kernel_ms'])"; done
done
