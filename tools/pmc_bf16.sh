#!/bin/bash
# PMC passes for one bf16 conv shape (GPU box).  usage: tools/pmc_bf16.sh <outdir> Cin Cout S variant [G]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=$1; shift
mkdir -p $OUT
P="python3 tools/run_one_bf16.py $*"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS -d $OUT/p1 -o p1 --output-format csv -- $P > $OUT/p1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU -d $OUT/p2 -o p2 --output-format csv -- $P > $OUT/p2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC -d $OUT/p3 -o p3 --output-format csv -- $P > $OUT/p3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum -d $OUT/p4 -o p4 --output-format csv -- $P > $OUT/p4.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/p5 -o p5 --output-format csv -- $P > $OUT/p5.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/p6 -o p6 --output-format csv -- $P > $OUT/p6.log 2>&1
python3 - <<PY
import csv, collections, glob
for pth in ("p1","p2","p3","p4","p5","p6"):
    fs = glob.glob("$OUT/%s/**/*counter_collection.csv" % pth, recursive=True)
    if not fs: print(pth, "no csv"); continue
    rows=list(csv.DictReader(open(fs[0])))
    agg=collections.defaultdict(list); dur=0
    for r in rows:
        if "conv_bf16" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    for k,v in agg.items(): print(pth, k, "%.5g"%(sum(v)/len(v)))
    print(pth, "last duration us", dur)
PY
