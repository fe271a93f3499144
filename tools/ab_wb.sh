#!/bin/bash
# interleaved A/B of library builds on the wideband configuration: tools/ab_wb.sh A B [...] (names of tools/ab_libs/lib_<name>.so: tools/mkvariant.sh)
for rep in 1 2; do for v in "$@"; do
  export SONDE_MI355_LIB=$PWD/tools/ab_libs/lib_$v.so
  for S in 1 8; do python bench.py --wideband --wb-streams $S --steps 60 --warmup 20 --no-cpu 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v S=$S', d['ms_per_step'], d['value'], d['kernel_ms'])"; done
done; done
