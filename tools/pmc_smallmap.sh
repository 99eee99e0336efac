#!/bin/bash
# PMC passes for one conv shape (GPU box).  usage: tools/pmc_smallmap.sh <shape> <cfg-name-or-id> <outdir>
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SHAPE=$1; CFG=$2; OUT=$3
mkdir -p $OUT
P="python tools/run_one_conv.py $SHAPE $CFG 3"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS -d $OUT/p1 -o p1 --output-format csv -- $P > $OUT/p1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU -d $OUT/p2 -o p2 --output-format csv -- $P > $OUT/p2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC -d $OUT/p3 -o p3 --output-format csv -- $P > $OUT/p3.log 2>&1
python - <<PY
import csv, collections
for pth in ("p1","p2","p3"):
    rows=list(csv.DictReader(open("$OUT/%s/%s_counter_collection.csv"%(pth,pth))))
    agg=collections.defaultdict(list)
    for r in rows:
        if "conv_smallmap" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    for k,v in agg.items(): print(pth, k, "%.4g"%(sum(v)/len(v)))
    print(pth, "last duration us", dur)
PY
