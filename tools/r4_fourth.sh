#!/bin/bash
# round 4, fourth GPU call: A/B of the bins-path decoder at 8 waves per SIMD, the overlap option, wideband counters, the rest of the suite
mkdir -p gpurun_out
bash tools/ab_wb.sh base w8 2>&1 | tee gpurun_out/r4_f_ab.txt
unset SONDE_MI355_LIB
for ov in 0 1; do for cfg in "8 1" "8 4"; do set -- $cfg
  python bench.py --wideband --wb-streams $1 --wb-blocks $2 --steps 100 --warmup 20 $( [ $ov = 1 ] && echo --wb-overlap ) 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('overlap', $ov, 'wb', $1, $2, d['ms_per_step'], d['value'], d['kernel_ms'])"
done; done 2>&1 | tee -a gpurun_out/r4_f_ab.txt
bash tools/r4_wb_counters.sh > /dev/null 2>&1
grep -E "FETCH_SIZE|WRITE_SIZE" gpurun_out/r4_wb_counters.csv
( time timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/r4_f_pytest.log 2>&1
echo "pytest rc=$?"; tail -6 gpurun_out/r4_f_pytest.log
