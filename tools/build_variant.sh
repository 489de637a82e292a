#!/bin/bash
# build a variant of libcodeps_hip.so with extra -D flags on conv.hip only:  tools/build_variant.sh <name> <flags...>
# -> codeps_amd/variants/libcodeps_hip_<name>.so   (select with CODEPS_HIP_LIB)
set -e
name=$1; shift
cd "$(dirname "$0")/../codeps_amd/csrc"
mkdir -p ../variants build
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-result "$@" -c conv.hip -o build/conv_$name.o
objs=$(ls build/*.o | grep -v "build/conv\.hip\.o\|build/conv_")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libcodeps_hip_$name.so build/conv_$name.o $objs -Wl,--no-undefined
echo built $name
