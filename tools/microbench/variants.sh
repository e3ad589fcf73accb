#!/bin/bash
# A/B timing of kernel builds INSIDE ONE gpurun session (box-to-box spread is +-3-5 %, so variants are only ever compared
# within a session).  Build the variants in the build container, e.g.
#     NPF16_EXTRA_FLAGS="-DNPF16_EXP=1" python -c "from neuralplane_amd import build; build.build_hip(force=True)"
#     cp neuralplane_amd/csrc/libneuralplane_hip.so tools/microbench/libs/exp1.so      (libs/ is git-ignored)
# then:  gpurun -- 'bash tools/microbench/variants.sh exp0 exp1 ...'
# Timing-only switches: NPF16_EXP bit 1 = no observation noise, 2 = no Overload re-evaluation, 4 = no MLP evaluation, 8 = hardware sin/cos/pow instead of the fp64 sequences
# (any NPF16_EXP value also selects the per-class asm statements);
# NPF16_PHASE_ASM=0 = one asm statement per class; NPF16_BLOCK, NPF16_MINWAVES, NPF16_STAGGER_CYCLES, NPF16_COMBAT_MINWAVES.
cd $GRAFT_REPO_ROOT
run() { timeout 120 python bench.py --steps 100 --warmup 5 --no-cpu-baseline $2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', 'kernel_ms', round(d['roofline']['kernel_avg_ms'],4), 'steps/s', '%.3e'%d['value'])"; }
cp neuralplane_amd/csrc/libneuralplane_hip.so /tmp/keep.so
for rep in 1 2; do
  for v in "$@"; do
    cp tools/microbench/libs/$v.so neuralplane_amd/csrc/libneuralplane_hip.so; touch neuralplane_amd/csrc/libneuralplane_hip.so
    run "$v mlp" ""; run "$v tables" "--aero-1d-tables 1"
  done
done
cp /tmp/keep.so neuralplane_amd/csrc/libneuralplane_hip.so
