#!/bin/bash
# build a variant of libcodeps_hip.so with extra -D flags on conv.hip and chain1x1.hip only:  tools/build_variant.sh <name> <flags...>
# -> codeps_amd/variants/libcodeps_hip_<name>.so   (select with CODEPS_HIP_LIB)
#    tools/build_variant.sh stamps -DCODEPS_CLOCK_STAMPS   = the diagnostic build with in-kernel clock stamps (tools/clock_cycles.py)
set -e
name=$1; shift
cd "$(dirname "$0")/../codeps_amd/csrc"
mkdir -p ../variants build
for f in conv chain1x1; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-result "$@" -c $f.hip -o build/var_${f}_$name.o &
done
wait
objs=$(ls build/*.o | grep -v "build/conv\.hip\.o\|build/chain1x1\.hip\.o\|build/conv_\|build/var_\|build/diag\|build/conv1x1s\|build/core_diag")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libcodeps_hip_$name.so build/var_conv_$name.o build/var_chain1x1_$name.o $objs -Wl,--no-undefined
echo built $name
