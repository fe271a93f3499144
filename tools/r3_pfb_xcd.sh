#!/bin/bash
# filter bank: step groups mapped so that an XCD's resident workgroups are neighbours in time (default) against launch order
for rep in 1 2; do for v in XCD NOXCD; do
  export SONDE_MI355_LIB=$PWD/ab/lib_$v.so
  for S in 1 8; do for B in 1 4; do python bench.py --wideband --wb-streams $S --wb-blocks $B --steps 60 --warmup 20 --no-cpu 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v S=$S B=$B', d['ms_per_step'], d['value'], d['kernel_ms'])"; done; done
done; done
