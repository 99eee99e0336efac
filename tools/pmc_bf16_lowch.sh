#!/bin/bash
# PMC passes on the general bf16 kernel for one low-channel layer (64 -> 64 at 512^2, bf16 activations).  usage: tools/pmc_bf16_lowch.sh [outdir]
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
OUT=${1:-gpurun_out/pmc_bf16_lowch}
mkdir -p $OUT
P="python3 tools/run_one_bf16.py 64 64 512 0"
export IO_BF16=1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $OUT/p1 -o p1 --output-format csv -- $P > $OUT/p1.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU -d $OUT/p2 -o p2 --output-format csv -- $P > $OUT/p2.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_BF16 -d $OUT/p3 -o p3 --output-format csv -- $P > $OUT/p3.log 2>&1
python3 - <<PY
import csv, collections, glob
for pth in ("p1","p2","p3"):
    fs = glob.glob("$OUT/%s/**/*counter_collection.csv" % pth, recursive=True)
    if not fs: print(pth, "no csv"); continue
    rows=list(csv.DictReader(open(fs[0])))
    agg=collections.defaultdict(list); dur=0; name=""
    for r in rows:
        if "conv_bf16" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"])); name=r["Kernel_Name"][:90]
            dur=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    for k,v in agg.items(): print(pth, k, "%.5g"%(sum(v)/len(v)))
    print(pth, name, "last duration us", dur)
PY
