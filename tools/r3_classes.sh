#!/bin/bash
# round 3: every class alone (1024 x 96 tiles), the mix at both block lengths, and the mix's kernel trace
export TMPDIR=/tmp
R=$PWD
TAG=${TAG:-r3}
mkdir -p gpurun_out
for t in 0 1 3; do
  python bench.py --sonde-type $t --no-cpu --steps 100 --warmup 20 > gpurun_out/${TAG}_type${t}_bench.json 2>> gpurun_out/${TAG}_classes.err
done
python bench.py --mix --channels 4096 --tiles 24 --no-cpu > gpurun_out/${TAG}_mix_bench.json 2>> gpurun_out/${TAG}_classes.err
python bench.py --mix --no-cpu > gpurun_out/${TAG}_mix96_bench.json 2>> gpurun_out/${TAG}_classes.err
cd /tmp && rm -rf /tmp/p_mix
rocprofv3 --kernel-trace --stats -d /tmp/p_mix -o trace -- python $R/bench.py --mix --channels 4096 --tiles 24 --steps 100 --warmup 20 --no-cpu > /tmp/m1.log 2>&1
cd $R
python tools/rocprof_summary.py $(find /tmp/p_mix -name '*.db') > gpurun_out/${TAG}_mix_rocprof.csv 2> gpurun_out/rocprof.err
python tools/rocprof_timeline.py $(find /tmp/p_mix -name '*.db') > gpurun_out/${TAG}_mix_timeline.txt 2>> gpurun_out/rocprof.err
for f in gpurun_out/${TAG}_type*_bench.json gpurun_out/${TAG}_mix_bench.json gpurun_out/${TAG}_mix96_bench.json; do python - "$f" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[1], d["ms_per_step"], d["value"], d["roofline"]["step_frac"], d["kernel_ms"])
PY
done
cat gpurun_out/${TAG}_mix_rocprof.csv; tail -30 gpurun_out/${TAG}_mix_timeline.txt
