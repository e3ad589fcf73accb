#!/bin/bash
# round 4, session 20: stack of fixed-point Linear + ReLU layers with quantiser and epilogue on the device (DESIGN 14 b); timing builds without
# the quantiser / the epilogue (wrong results) price the phases
cd $GRAFT_REPO_ROOT; out=gpurun_out/r04_s20; mkdir -p $out
(timeout 60 tools/microbench/i8_mlp_stack 1; timeout 60 tools/microbench/i8_mlp_stack 9
 for v in DSKIP_QUANT DSKIP_EPI DSKIP_QUANTDSKIP_EPI; do echo "timing build $v:"; timeout 60 tools/microbench/i8_mlp_stack$v 2 | tail -1; done) 2>&1 | tee $out/i8_mlp_stack.log
