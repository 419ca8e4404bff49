#!/bin/bash
# Dev tool: diagnostic / candidate builds of csrc/flash.hip alone (-DFL_*; the ablation builds give WRONG results by
# construction) as csrc/build/abl/libflash_<tag>.so (git-ignored, travels to the GPU box), loaded side by side by
# tools/flash_lab.py.  Usage: tools/flash_abl.sh TAG "-DFL_X ..."   |   tools/flash_abl.sh --all  (the r04 table's set)
set -e
cd "$(dirname "$0")/.."
CS=$(ls -d dec*/csrc)
mkdir -p $CS/build/abl
one() {
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 $2 -c $CS/flash.hip -o $CS/build/abl/flash_$1.o 2>/dev/null
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $CS/build/abl/flash_$1.o $CS/build/common.o -ldl -o $CS/build/abl/libflash_$1.so
    rm -f $CS/build/abl/flash_$1.o
}
if [ "$1" == "--all" ]; then
    one base "" &
    one stamps "-DFL_STAMPS" &
    one noexp "-DFL_ABL_NOEXP" &
    one nosum "-DFL_ABL_NOSUM" &
    one nomax "-DFL_ABL_NOMAX" &
    one nodma "-DFL_ABL_NODMA" &
    wait
    one nokread "-DFL_ABL_NOKREAD" &
    one novread "-DFL_ABL_NOVREAD" &
    one noreads "-DFL_ABL_NOKREAD -DFL_ABL_NOVREAD" &
    one nobar "-DFL_ABL_NODMA -DFL_ABL_NOBAR" &
    one novalu "-DFL_ABL_NOEXP -DFL_ABL_NOSUM -DFL_ABL_NOMAX" &
    one mfmaonly "-DFL_ABL_NOEXP -DFL_ABL_NOSUM -DFL_ABL_NOMAX -DFL_ABL_NODMA -DFL_ABL_NOBAR -DFL_ABL_NOKREAD -DFL_ABL_NOVREAD" &
    wait
else
    one "$1" "$2"
fi
