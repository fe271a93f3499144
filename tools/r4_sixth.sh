#!/bin/bash
# round 4, sixth GPU call: 4-step filter-bank workgroups for small launches: parity + A/B
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_channelizer.py -m gpu -q ) > gpurun_out/r4_i_chan.log 2>&1
echo "chan rc=$?"; tail -6 gpurun_out/r4_i_chan.log
for rep in 1 2; do for s in 8 4; do for cfg in "1 1" "1 2" "2 1" "8 1"; do set -- $cfg
  SONDE_PFB_S=$s python bench.py --wideband --wb-streams $1 --wb-blocks $2 --steps 100 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('S=$s wb', $1, $2, d['ms_per_step'], d['value'], d['kernel_ms'])"
done; done; done 2>&1 | tee gpurun_out/r4_i_ab.txt
