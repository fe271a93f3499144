#!/bin/bash
# wideband front-end: overlapped (default) against serial submits, 1 and 8 streams, 1 / 4 / 8 blocks per submit
mkdir -p gpurun_out
out=gpurun_out/r3_wb_overlap.txt
: > $out
for S in 1 8; do for B in 1 4 8; do for M in "" "--wb-overlap"; do
  python bench.py --wideband --wb-streams $S --wb-blocks $B $M --steps 100 --warmup 20 --no-cpu 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('S=$S B=$B mode=${M:-serial}', 'ms/step', d['ms_per_step'], 'us/stream-block', round(d['ms_per_step']*1e3/($S*$B),2), 'Msps', d['value'], d['kernel_ms'], 'frames', d['frames_per_step'])" >> $out
done; done; done
cat $out
