#!/bin/bash
# interleaved A/B of library builds on the mixed batch (config 3), joined and pipelined, and on single-type batches of the
# fixed-length sondes: tools/ab_mix.sh A B [...] (names of ab/lib_<name>.so)
p() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[0]); print('$1', d['ms_per_step'], d['roofline']['step_frac'], d['kernel_ms'].get('demod'), d['kernel_ms'].get('framer_fec'))"; }
for rep in 1 2; do for v in "$@"; do
  export SONDE_MI355_LIB=$PWD/tools/ab_libs/lib_$v.so
  python bench.py --mix --channels 4096 --tiles 24 --flags 0 --steps 100 --warmup 20 --no-cpu 2>/dev/null | p "$v joined   "
  python bench.py --mix --channels 4096 --tiles 24 --flags 4 --steps 100 --warmup 20 --no-cpu 2>/dev/null | p "$v pipelined"
  python bench.py --sonde-type 1 --steps 60 --warmup 15 --no-cpu --no-others 2>/dev/null | p "$v DFM 1024x96"
  python bench.py --sonde-type 3 --steps 60 --warmup 15 --no-cpu --no-others 2>/dev/null | p "$v M10 1024x96"
done; done
