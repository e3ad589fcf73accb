#!/bin/bash
# round 3, GPU session 15: one long delay for half of a one-generation grid of the pair variant (two sub-grids with a phase offset)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_s15; mkdir -p $out
for lib in k_head k_b8_2x16k k_b8_2x32k k_b0_2x32k k_b0_3x16k; do NPF16_LIB=tools/microbench/libs/$lib.so timeout 400 python tools/microbench/mid_n.py --variants auto --out $out/mid_$lib.json 100000 114688 131072 163840 196608 262144 > $out/mid_$lib.log 2>&1; grep "N=" $out/mid_$lib.log | sed "s/^/$lib /"; done
