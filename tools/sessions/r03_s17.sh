#!/bin/bash
# round 3, GPU session 17: phase stamps of the 32-row controller kernel
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_s17; mkdir -p $out
NPF16_LIB=tools/microbench/libs/act_trace.so timeout 300 python tools/microbench/actor_phases.py > $out/phases.log 2>&1; grep -v Warn $out/phases.log
