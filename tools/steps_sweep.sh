#!/bin/bash
for a in "20 3" "200 3" "200 50" "2000 3" "2000 200" "20 200" "20 1000"; do
  set -- $a
  echo -n "steps=$1 warmup=$2: "
  python bench.py --no-cpu --steps $1 --warmup $2 | grep -o '"value[^,]*,\|"kernel_ms[^}]*}' | tr '\n' ' '; echo
done
