#!/bin/bash
# A/B builds of the library: tools/build_variant.sh <name> [-Dflags...]  ->  build/ab/<name>/libzkgpu.so  (run with ZKGPU_LIB=...)
set -e
N=$1; shift
mkdir -p build/ab/$N
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -fvisibility=hidden -Wl,--version-script=zkvm_amd/csrc/zkgpu.map "$@" -o build/ab/$N/libzkgpu.so zkvm_amd/csrc/zkgpu.hip
echo build/ab/$N/libzkgpu.so
