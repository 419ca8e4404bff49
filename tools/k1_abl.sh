#!/bin/bash
# Dev tool: candidate builds of the K1 kernel (csrc/conv_fused.hip or another source file with the same entry points) as
# csrc/build/abl/libk1_<tag>.so (git-ignored, travels to the GPU box), timed side by side by tools/k1_lab.py.
# Usage: tools/k1_abl.sh TAG [SOURCE.hip] ["-DFLAGS"]
set -e
cd "$(dirname "$0")/.."
CS=$(ls -d dec*/csrc)
mkdir -p $CS/build/abl
SRC=${2:-$CS/conv_fused.hip}
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -I$CS -Iinclude $3 -c $SRC -o $CS/build/abl/k1_$1.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $CS/build/abl/k1_$1.o $CS/build/common.o -ldl -o $CS/build/abl/libk1_$1.so
rm -f $CS/build/abl/k1_$1.o
