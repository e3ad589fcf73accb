#!/bin/bash
# experimental build of the HIP extension into tools/microbench/libs/<name>.so (git-ignored; selected through NPF16_LIB by ab_libs.py)
#   usage: tools/microbench/build_lib.sh name [-DFLAG ...]
name=$1; shift
here=$(cd $(dirname $0)/../.. && pwd)
mkdir -p $here/tools/microbench/libs
cd $here/neuralplane_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -shared -mllvm -disable-machine-licm "$@" \
  -o $here/tools/microbench/libs/$name.so np_*.hip && echo built $name   # every translation unit of neuralplane_amd/build.py::SOURCES
