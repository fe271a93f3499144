#!/bin/bash
# interleaved A/B of two builds of the library (run-to-run noise on one box is 1-2 %: single runs cannot rank close variants)
# usage: tools/ab_repeat.sh ab/lib_A.so ab/lib_B.so [reps]
A=$1; B=$2; REPS=${3:-3}
run() { SONDE_MI355_LIB=$1 python bench.py $2 --no-cpu --no-others 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for cfg in "--steps 200 --warmup 40" "--steps 200 --warmup 40 --ebn0 9" "--channels 8192 --tiles 24" "--channels 4096 --tiles 96 --steps 60 --warmup 15" "--mix --channels 4096 --tiles 24"; do
  ra=""; rb=""
  for i in $(seq $REPS); do ra="$ra $(run $PWD/$A "$cfg")"; rb="$rb $(run $PWD/$B "$cfg")"; done
  python -c "
import statistics as st
a=[float(x) for x in '$ra'.split()]; b=[float(x) for x in '$rb'.split()]
print('$cfg | A', a, 'median', st.median(a), '| B', b, 'median', st.median(b), '| B/A', round(st.median(b)/st.median(a),4))"
done
