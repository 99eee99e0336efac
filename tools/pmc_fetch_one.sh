#!/bin/bash
# FETCH_SIZE / WRITE_SIZE / L2 hit counters of ONE bf16 conv shape under a VSP_CONV_DBG ablation (GPU box; ablation library).
# usage: tools/pmc_fetch_one.sh <tag> <dbg> Cin Cout S variant [G]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; DBG=$2; shift; shift
OUT=gpurun_out/pmc_fetch_$TAG; mkdir -p $OUT
export VSPBFR_HIP_LIB=$GRAFT_REPO_ROOT/vspbfr_amd/lib/libvspbfr_hip_ablate.so VSP_CONV_DBG=$DBG VSP_TUNE=1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/f -o f --output-format csv -- python3 tools/run_one_bf16.py $* > $OUT/f.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum -d $OUT/t -o t --output-format csv -- python3 tools/run_one_bf16.py $* > $OUT/t.log 2>&1
python3 - <<PY
import csv, glob, collections
print("== $TAG dbg=$DBG args: $*  IO_BF16=${IO_BF16:-0}")
for d in ("f", "t"):
    fs = glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True)
    if not fs: print(d, "no csv"); continue
    agg = collections.defaultdict(list); dur = []
    for r in csv.DictReader(open(fs[0])):
        if "conv_bf16" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"])); dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in agg.items(): print("  ", k, "%.5g" % (sum(v) / len(v)), ("= %.3f GB (x2 KiB calibration)" % (sum(v) / len(v) * 2048 / 1e9)) if k == "FETCH_SIZE" else "")
    if dur: print("   duration us", min(dur))
PY
