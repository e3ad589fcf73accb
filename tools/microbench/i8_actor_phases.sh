#!/bin/bash
# builds tools/microbench/i8_actor_phases.hip with parts of the call removed and times each (see the .hip); run from the repo root on a GPU box
cd $(dirname $0)/../..
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -mllvm -disable-machine-licm"
for v in "0 0" "1 0" "2 0" "3 0" "16 0" "0 8" "4 0" "8 0" "12 0" "19 8" "31 8"; do   # 4: no LayerNorm exchange, 8: no quantiser arithmetic (round 6)
  set -- $v
  hipcc $F -DNPACT8_EXP=$1 -DNPACT_EXP=$2 tools/microbench/i8_actor_phases.hip -o /tmp/i8ph_$1_$2 2>/dev/null && for n in ${N:-8192}; do /tmp/i8ph_$1_$2 $n; done   # N="8192 262144": several batch sizes from one build
done
