#!/bin/bash
# PMC passes for one Winograd conv shape (GPU box).  usage: tools/pmc_wino.sh <outdir> Cin Cout S [G]
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
OUT=$1; shift
mkdir -p $OUT
P="python3 tools/run_one_wino.py $*"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS -d $OUT/p1 -o p1 --output-format csv -- $P > $OUT/p1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU -d $OUT/p2 -o p2 --output-format csv -- $P > $OUT/p2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC -d $OUT/p3 -o p3 --output-format csv -- $P > $OUT/p3.log 2>&1
python3 - <<PY
import csv, collections, glob, sys
missing = False
print("== Winograd kernels  $*  WINO=${WINO:-2}")
vals = collections.defaultdict(dict)
for pth in ("p1","p2","p3"):
    fs = glob.glob("$OUT/%s/**/*counter_collection.csv" % pth, recursive=True)
    if not fs: print(pth, "no csv"); missing = True; continue
    rows=list(csv.DictReader(open(fs[0])))
    agg=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(list)
    for r in rows:
        if "wino" in r["Kernel_Name"] and "weight" not in r["Kernel_Name"]:
            import re
            name = re.search(r"(\\w*wino\\w*)", r["Kernel_Name"]).group(1)
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[name].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
    for name in agg:
        for k,v in agg[name].items():
            vals[name][k] = sum(v)/len(v); print(pth, name, k, "%.5g"%vals[name][k])
        print(pth, name, "duration us (min over dispatches)", min(dur[name]))
for name, v in vals.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v:   # GRBM_GUI_ACTIVE is summed over the 8 XCDs, the SQ counters over 1024 SIMDs
        print(name, "MFMA pipe busy: %.1f %%" % (100 * v["SQ_VALU_MFMA_BUSY_CYCLES"] / (128 * v["GRBM_GUI_ACTIVE"])))
    if "SQ_LDS_BANK_CONFLICT" in v and v.get("SQ_LDS_IDX_ACTIVE"):
        print(name, "LDS conflict / active: %.3f" % (v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"]))
    if "SQ_WAIT_INST_ANY" in v and "SQ_WAVE_CYCLES" in v:
        print(name, "waves waiting / wave cycles: %.2f" % (v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"]))
sys.exit(1 if missing else 0)   # a pass without a counter file is a failed run, not evidence
PY
