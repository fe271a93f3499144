#!/bin/bash
run() { SONDE_MI355_LIB=$1 python bench.py $2 --no-cpu --no-others 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_ms']['demod'])"; }
for cfg in "--sonde-type 3 --steps 100 --warmup 20" "--mix --channels 4096 --tiles 24"; do
  for i in 1 2 3; do echo "PREV $cfg: $(run $PWD/tools/ab_libs/lib_PREV.so "$cfg")  NEW: $(run $PWD/sdrpp_radiosonde_amd/libsonde_mi355.so "$cfg")"; done
done
