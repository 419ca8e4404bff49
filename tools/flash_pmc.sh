#!/bin/bash
# Dev tool (GPU box): rocprofv3 --pmc passes over tools/flash_one.py, one counter group per pass (8 SQ slots), summed per
# kernel by tools/pmc_rows.py.  Usage: tools/flash_pmc.sh OUTDIR
set -o pipefail
OUT=${1:-gpurun_out/flash_pmc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P="python3 tools/flash_one.py 4"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $OUT/p1 -- $P > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/p2 -- $P > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS --output-format csv -d $OUT/p3 -- $P > $OUT/p3.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d $OUT/p4 -- $P > $OUT/p4.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/p5 -- $P > $OUT/p5.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/p6 -- $P > $OUT/p6.log 2>&1
python3 tools/pmc_rows.py $OUT flash_attn > $OUT/rows.txt
cat $OUT/rows.txt
rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 $OUT/p5 $OUT/p6
