#!/bin/bash
# norm_bench under a kernel trace: per-kernel times by shape.  Usage (GPU box): bash tools/norm_bench.sh <tag>
set -o pipefail
OUT=gpurun_out/${1:-nb}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 tools/norm_bench.py > $OUT/whole.txt 2>&1
python tools/norm_bench.py --trace $OUT/trace > $OUT/kernels.txt 2>&1
rm -rf $OUT/trace
cat $OUT/whole.txt | grep -v amdgpu.ids; cat $OUT/kernels.txt
