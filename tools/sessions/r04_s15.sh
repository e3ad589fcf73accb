#!/bin/bash
# round 4, GPU session 15: (a) A/B of the cache check: full cold path (k_check2), a one-instruction cold path (k_trap), no check (k_nocheck);
# (b) which launch of the throughput variant built with NPF16_PIN_MASK=7 faults (DESIGN's parked fault)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_s15; mkdir -p $out
python tools/microbench/ab_libs.py --rounds 3 --steps 100 k_check2 k_trap k_nocheck 2>&1 | tee $out/ab_cache_check3.log | tail -5
NPF16_LIB=tools/microbench/libs/k_pin7.so timeout 900 python tools/microbench/fault_probe.py throughput 2>&1 | tee $out/fault_probe_pin7.log
