#!/bin/bash
# round 4, session 16: targeted probes of v_mfma_f32_16x16x32_bf16 (operands from tools/microbench/mfma_bf16_probe.py gen)
cd $GRAFT_REPO_ROOT; out=gpurun_out/r04_s16; mkdir -p $out
for f in tools/microbench/mfma_bf16_probe*_in.bin; do
  b=$(basename $f _in.bin)
  timeout 60 tools/microbench/mfma_bf16_model --in $f $out/${b}_out.bin
done
