#!/bin/bash
# usage: tools/pmc_run.sh "<counters pass 1>" "<counters pass 2>" ...   (one rocprofv3 --pmc pass each)
export TMPDIR=/tmp
R=$PWD
i=0
for pass in "$@"; do
  i=$((i+1))
  cd /tmp && rm -rf /tmp/p_pmc$i
  rocprofv3 --pmc $pass -d /tmp/p_pmc$i -o pmc -- python $R/bench.py --steps 5 --warmup 2 --ramp-ms 60 --no-cpu > /tmp/pmc$i.log 2>&1 || tail -5 /tmp/pmc$i.log
  cd $R
  python tools/rocprof_summary.py $(find /tmp/p_pmc$i -name '*.db') | grep -v read_probe
done
