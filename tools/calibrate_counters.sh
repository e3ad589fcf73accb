#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on the env kernel's access patterns (tools/microbench/counter_calib.hip), separate --pmc passes
# with --kernel-trace only.   usage (via gpurun): bash tools/calibrate_counters.sh [n]   -> gpurun_out/calib/{FETCH_SIZE,WRITE_SIZE}
n=${1:-4194304}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/calib; mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$c -o p -- tools/microbench/counter_calib $n 5 > $out/$c.log 2>&1 < /dev/null
done
ls $out
