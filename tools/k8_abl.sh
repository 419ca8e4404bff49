#!/bin/bash
# Dev tool: timing-only builds of csrc/ff_fused.hip with parts of the tile removed (-DK8_ABL_*; wrong results), linked
# against the product's other objects into csrc/build/abl/libvdx_hip_<tag>.so (git-ignored, travels to the GPU box).  Usage: tools/k8_abl.sh TAG "-DK8_ABL_X ..."
set -e
cd "$(dirname "$0")/.."
CS=$(ls -d dec*/csrc)
mkdir -p $CS/build/abl
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 $2 -c $CS/ff_fused.hip -o $CS/build/abl/ff_$1.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $CS/build/abl/ff_$1.o $(ls $CS/build/*.o | grep -v ff_fused.o) -ldl -o $CS/build/abl/libvdx_hip_$1.so
