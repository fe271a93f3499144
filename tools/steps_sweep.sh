#!/bin/bash
# ms per step of the headline against the length of the timed region and of the warmup (the driver types --steps 20 --warmup 5)
for a in "20 5" "200 50" "20 5" "20 200" "40 5" "100 5" "200 5" "20 5"; do
  set -- $a
  echo -n "steps=$1 warmup=$2: "
  python bench.py --no-cpu --no-others --steps $1 --warmup $2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_ms']['demod'], d['roofline']['step_frac'])"
done
