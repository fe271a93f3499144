#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  SONDE_MI355_LIB=$PWD/$lib rocprofv3 --kernel-trace --stats -d gpurun_out/abk_$(basename $lib .so) -o t -- python bench.py --steps 10 --warmup 2 --no-cpu > /dev/null 2>&1
done
