#!/bin/bash
# round 3, GPU session 21: the region between one generation and the headline size (131 072 .. 1e6) — auto variant and the pair
# variant pinned to two / three waves per SIMD, with the first-generation de-phasing applied from 2 generations (shipped), from 1
# generation (b_mg1) and to every grid (b_mg0).  One session, so the columns compare.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_s21; mkdir -p $out
sizes="131072 163840 196608 229376 262144 327680 393216 458752 524288 655360 786432 1000000"
for lib in a_base b_mg1 b_mg0; do
  NPF16_LIB=tools/microbench/libs/$lib.so timeout 300 python tools/microbench/mid_n.py --variants auto --steps 200 --out $out/${lib}_auto.json $sizes 2>/dev/null | grep "N=" | sed "s/^/$lib auto  /"
  for pw in 2 3; do
    NPF16_PAIR_WAVES=$pw NPF16_LIB=tools/microbench/libs/$lib.so timeout 300 python tools/microbench/mid_n.py --variants pair --steps 200 --out $out/${lib}_pair$pw.json $sizes 2>/dev/null | grep "N=" | sed "s/^/$lib pair$pw /"
  done
done > $out/table.log 2>&1
cat $out/table.log
