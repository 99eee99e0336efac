#!/bin/bash
# PMC passes for one conv shape (GPU box).  usage: tools/pmc_conv.sh <shape> <cfg> <outdir>
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SHAPE=$1; CFG=$2; OUT=$3
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS -d $OUT/p1 -o p1 --output-format csv -- python tools/run_one_conv.py $SHAPE $CFG 3 > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU -d $OUT/p2 -o p2 --output-format csv -- python tools/run_one_conv.py $SHAPE $CFG 3 > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d $OUT/p3 -o p3 --output-format csv -- python tools/run_one_conv.py $SHAPE $CFG 3 > $OUT/p3.log 2>&1
ls $OUT/p1 $OUT/p2
