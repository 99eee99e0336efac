#!/bin/bash
# PMC passes for one layer on a pipelined conv configuration (GPU box).  usage: tools/pmc_pipe.sh <outdir> <shape> <config name | tuned>
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
OUT=$1; shift
mkdir -p $OUT
P="python3 tools/run_one_pipe.py $*"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS -d $OUT/p1 -o p1 --output-format csv -- $P > $OUT/p1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU -d $OUT/p2 -o p2 --output-format csv -- $P > $OUT/p2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC -d $OUT/p3 -o p3 --output-format csv -- $P > $OUT/p3.log 2>&1
python3 - <<PY
import csv, collections, glob, sys
missing = False
print("== conv kernel of  $*")
vals = {}
for pth in ("p1","p2","p3"):
    fs = glob.glob("$OUT/%s/**/*counter_collection.csv" % pth, recursive=True)
    if not fs: print(pth, "no csv"); missing = True; continue
    rows=list(csv.DictReader(open(fs[0])))
    agg=collections.defaultdict(list); dur=[]; name=""
    for r in rows:
        if "vspconv::" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"])); name=r["Kernel_Name"][:70]
            dur.append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
    for k,v in agg.items():
        vals[k] = sum(v)/len(v); print(pth, k, "%.5g"%vals[k])
    print(pth, name, "duration us (min over dispatches)", min(dur) if dur else None)
if "SQ_VALU_MFMA_BUSY_CYCLES" in vals and "GRBM_GUI_ACTIVE" in vals:
    print("MFMA pipe busy: %.1f %%" % (100 * vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (128 * vals["GRBM_GUI_ACTIVE"])))
if "SQ_LDS_BANK_CONFLICT" in vals:
    print("LDS conflict / active: %.3f" % (vals["SQ_LDS_BANK_CONFLICT"] / max(vals["SQ_LDS_IDX_ACTIVE"], 1)))
for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
    if k in vals and "SQ_WAVE_CYCLES" in vals: print(k, "/ wave cycles: %.2f" % (vals[k] / vals["SQ_WAVE_CYCLES"]))
sys.exit(1 if missing else 0)   # a pass without a counter file is a failed run, not evidence
PY
