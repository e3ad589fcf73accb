#!/bin/bash
# round 4, session 17: block-fixed-point dense layer on the i8 matrix pipe (DESIGN 14 b): exactness against int64 and cycles per layer
cd $GRAFT_REPO_ROOT; out=gpurun_out/r04_s17; mkdir -p $out
timeout 60 tools/microbench/i8_dense_layer 9 2>&1 | tee $out/i8_dense_layer.log
timeout 60 tools/microbench/mfma_rates 2>&1 | tee $out/mfma_rates.log
