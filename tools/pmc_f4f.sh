#!/bin/bash
# Round 5: PMC passes on the fused F(4x4) kernel (conv_wino4f.hip).  usage: tools/pmc_f4f.sh <outdir> Cin Cout S
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
OUT=$1; shift
mkdir -p $OUT
export WINO=5
P="python3 tools/run_one_wino.py $*"
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_IFETCH"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o p$i --output-format csv -- $P > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, collections, glob
for i in range(1, 5):
    fs = glob.glob("$OUT/p%d/**/*counter_collection.csv" % i, recursive=True)
    if not fs: print("p%d" % i, "no csv"); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "wino4f_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items(): print("p%d" % i, k, "%.6g" % (sum(v) / len(v)), "(n=%d)" % len(v))
PY
rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4
