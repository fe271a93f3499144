#!/bin/bash
# clocks and power under (a) kernel A's headline, (b) the pure-read microbenchmark of the same shape (is the launch power- or clock-limited?)
cd tools/ubench && hipcc --offload-arch=gfx950 -O3 -o stride_pattern stride_pattern.hip 2>/dev/null; cd ../..
sample() { for i in $(seq 1 $1); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|socclk|Power" | sed 's/.*: *//' | tr '\n' ' '; echo; sleep 0.2; done; }
echo "== idle"; sample 3
echo "== kernel A headline (20000 steps)"
( sleep 1.5; sample 12 ) > /tmp/smi_a.log &
python bench.py --no-cpu --no-others --steps 20000 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], d['roofline']['step_frac'])"
wait; cat /tmp/smi_a.log
echo "== pure read, same shape (stride_pattern)"
( sleep 1.0; sample 12 ) > /tmp/smi_b.log &
./tools/ubench/stride_pattern > /tmp/sp.log 2>&1
wait; cat /tmp/smi_b.log; grep "2048 KiB depth 2\|4 workgroups per CU depth 3\|2 workgroups per CU depth 2" /tmp/sp.log
