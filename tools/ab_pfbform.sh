#!/bin/bash
# filter-bank forms side by side (SONDE_PFB_FORM), optional library builds: tools/ab_pfbform.sh "8 82" [libname ...]
forms=$1; shift
libs=${@:-default}
for rep in 1 2; do for v in $libs; do for f in $forms; do
  if [ $v = default ]; then unset SONDE_MI355_LIB; else export SONDE_MI355_LIB=$PWD/ab/lib_$v.so; fi
  for S in 1 8; do for B in 1 4; do SONDE_PFB_FORM=$f python bench.py --wideband --wb-streams $S --wb-blocks $B --steps 60 --warmup 20 --no-cpu 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v form $f S=$S B=$B', d['ms_per_step'], d['value'], d['kernel_ms'])"; done; done
done; done; done
