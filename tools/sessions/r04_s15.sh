#!/bin/bash
# round 4, session 15: dump of one v_mfma_f32_16x16x32_bf16 per trial (operands + result) for the arithmetic-model comparison on the CPU
cd $GRAFT_REPO_ROOT; out=gpurun_out/r04_s15; mkdir -p $out
timeout 60 tools/microbench/mfma_bf16_model $out/mfma_bf16_model.bin 96
