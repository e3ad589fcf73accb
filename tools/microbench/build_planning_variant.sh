#!/bin/bash
# usage: build_variant.sh name flags...   -> tools/microbench/libs/<name>.so (planning TU rebuilt with the flags, linked with the shipped objects of the other translation units)
name=$1; shift
cd /root/repo/neuralplane_amd/csrc
mkdir -p ../../tools/microbench/libs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -mllvm -disable-machine-licm "$@" -c -o ../../tools/microbench/libs/$name.o np_planning.hip 2>/dev/null && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o ../../tools/microbench/libs/$name.so $(ls build/*.o | grep -v np_planning.o) ../../tools/microbench/libs/$name.o && echo built $name
