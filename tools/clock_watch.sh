#!/bin/bash
# sample clocks/power while the bench loops (is kernel A clock- or power-limited on this box?)
( for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | tr '\n' ' '; echo; sleep 0.25; done ) > /tmp/smi.log &
python bench.py --no-cpu --steps 20000 --warmup 3 | grep -o '"kernel_ms[^}]*}\|"value[^,]*,'
wait
sort /tmp/smi.log | uniq -c | sort -rn | head -12
