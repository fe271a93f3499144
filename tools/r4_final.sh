#!/bin/bash
# round 4 closing run: evidence set (TAG) + campaigns + yardstick table on the final sources
TAG=${TAG:-r4_v2} bash tools/r4_evidence.sh > gpurun_out/${TAG:-r4_v2}_evidence.log 2>&1
tail -30 gpurun_out/${TAG:-r4_v2}_evidence.log
bash tools/r4_wb_counters.sh > /dev/null 2>&1; grep -E "FETCH_SIZE|WRITE_SIZE" gpurun_out/r4_wb_counters.csv
( time python tools/wb_campaign.py 60 ) > gpurun_out/r4_wb_campaign.log 2>&1; tail -2 gpurun_out/r4_wb_campaign.log
( time python tools/fuzz_campaign.py 400 ) > gpurun_out/r4_fuzz_campaign.log 2>&1; tail -3 gpurun_out/r4_fuzz_campaign.log
( time python tools/yardstick_study.py --gpu ) > gpurun_out/r4_yardstick.md 2> gpurun_out/r4_yardstick.err; tail -2 gpurun_out/r4_yardstick.err
