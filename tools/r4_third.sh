#!/bin/bash
# round 4, third GPU call: the 20 kS/s phase-output filter bank: channelizer tests, whole suite, wideband bench lines, campaign
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_channelizer.py -m gpu -q ) > gpurun_out/r4_e_chan.log 2>&1
echo "chan rc=$?"; tail -15 gpurun_out/r4_e_chan.log
for cfg in "1 1" "8 1" "8 4"; do set -- $cfg
  python bench.py --wideband --wb-streams $1 --wb-blocks $2 --steps 100 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wb', $1, $2, d['ms_per_step'], d['value'], d['kernel_ms'])"
done
( time timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/r4_e_pytest.log 2>&1
echo "pytest rc=$?"; tail -8 gpurun_out/r4_e_pytest.log
( time python tools/wb_campaign.py 60 ) > gpurun_out/r4_e_wbcampaign.log 2>&1
tail -3 gpurun_out/r4_e_wbcampaign.log
