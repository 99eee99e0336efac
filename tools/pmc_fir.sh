#!/bin/bash
# PMC passes for the blur kernel on one plane shape (GPU box).  Every pass runs under `timeout`: a TCC / TCP counter pass aborted inside
# rocprofv3 (signal 6) in round 4 and then sat until the box limit.  usage: tools/pmc_fir.sh <outdir> B C S [bf16]
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
OUT=$1; shift
mkdir -p $OUT
P="python3 tools/run_one_fir.py $*"
timeout 180 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE -d $OUT/p1 -o p1 --output-format csv -- $P > $OUT/p1.log 2>&1
timeout 180 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU -d $OUT/p2 -o p2 --output-format csv -- $P > $OUT/p2.log 2>&1
timeout 180 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_WAVES_EQ_64 SQ_LEVEL_WAVES -d $OUT/p3 -o p3 --output-format csv -- $P > $OUT/p3.log 2>&1
python3 - <<PY
import csv, collections, glob, sys
missing = False
print("== fir_tile_kernel  $*")
for pth in ("p1","p2","p3"):
    fs = glob.glob("$OUT/%s/**/*counter_collection.csv" % pth, recursive=True)
    if not fs: print(pth, "no csv"); missing = True; continue
    rows=list(csv.DictReader(open(fs[0])))
    agg=collections.defaultdict(list); dur=[]
    for r in rows:
        if "fir_tile" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur.append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
    for k,v in agg.items(): print(pth, k, "%.5g"%(sum(v)/len(v)))
    print(pth, "duration us (min over dispatches)", min(dur) if dur else None)
sys.exit(1 if missing else 0)
PY
