#!/bin/bash
# experiment: kernel time vs channel stride padding (samples)
for p in "$@"; do
  python bench.py --no-cpu --stride-pad $p --steps 30 > /tmp/o.json
  python -c "import sys,json; d=json.load(open('/tmp/o.json')); print(sys.argv[1], d['kernel_ms'], d['ms_per_step'])" $p
done
