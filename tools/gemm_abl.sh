#!/bin/bash
# Dev tool: a candidate / timing build of csrc/gemm.hip (e.g. "-DVDX_GEMM_PLAIN_LOOP": the K loop without the rolling
# fragment prefetch) linked with the product's other objects as csrc/build/abl/libvdx_<tag>.so (git-ignored, travels to the
# GPU box); tools/gemm_lab.py replays the product's launches on it.  Usage: tools/gemm_abl.sh TAG "-DFLAGS"   (after `make`)
set -e
cd "$(dirname "$0")/.."
CS=$(ls -d dec*/csrc)
mkdir -p $CS/build/abl
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 $2 -c $CS/gemm.hip -o $CS/build/abl/gemm_$1.o
OBJS=$(ls $CS/build/*.o | grep -v "/gemm.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $CS/build/abl/gemm_$1.o $OBJS -ldl -o $CS/build/abl/libvdx_$1.so
rm -f $CS/build/abl/gemm_$1.o
