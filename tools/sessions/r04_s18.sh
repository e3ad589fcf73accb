#!/bin/bash
# round 4, session 18: soaks on the final library: SingleCombat dual family vs the other kernels; the env kernel's pair vs throughput variants
cd $GRAFT_REPO_ROOT; out=gpurun_out/r04_s18; mkdir -p $out
timeout 600 python tools/microbench/soak_combat_dual.py 3000 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | tee $out/soak_combat_dual.log
timeout 600 python tools/microbench/soak_pair.py 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | tee $out/soak_pair.log
