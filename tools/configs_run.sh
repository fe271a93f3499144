#!/bin/bash
# throughput of the other BASELINE configurations (information for DESIGN.md section 6; the headline is the default run)
python bench.py --no-cpu --mix --channels 4096 --tiles 24 --steps 100 | tee gpurun_out/bench_mix4096.json
python bench.py --no-cpu --channels 8192 --tiles 24 --steps 100 | tee gpurun_out/bench_8192x24.json
python bench.py --no-cpu --channels 4096 --tiles 96 --steps 50 | tee gpurun_out/bench_4096x96.json
