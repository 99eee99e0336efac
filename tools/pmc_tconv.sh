#!/bin/bash
# Round 5 (review item 8): PMC evidence for the transposed up-convolutions -- MFMA busy, LDS conflicts, instruction mix (tools/pmc_pipe.sh) and
# FETCH_SIZE / WRITE_SIZE / L2 hit rate (separate passes) on 128 -> 64 at 257^2, 64 -> 32 at 513^2, 512 -> 512 at 33^2.   usage: tools/pmc_tconv.sh <outdir>
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
OUT=${1:-gpurun_out/pmc_tconv}
mkdir -p $OUT
for shape in up256 up512 up32; do
  bash tools/pmc_pipe.sh $OUT/$shape $shape tuned > $OUT/$shape.txt 2>&1
  P="python3 tools/run_one_pipe.py $shape tuned"
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum -d $OUT/$shape/p4 -o p4 --output-format csv -- $P > $OUT/$shape/p4.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_MISS_sum -d $OUT/$shape/p5 -o p5 --output-format csv -- $P > $OUT/$shape/p5.log 2>&1
  python3 - <<PY >> $OUT/$shape.txt
import csv, collections, glob
for pth in ("p4", "p5"):
    fs = glob.glob("$OUT/$shape/%s/**/*counter_collection.csv" % pth, recursive=True)
    if not fs: print(pth, "no csv"); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "vspconv::" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items(): print(pth, k, "%.5g" % (sum(v) / len(v)), "(FETCH_SIZE / WRITE_SIZE in KiB; FETCH x2 on gfx950 for wide coalesced reads)")
PY
  rm -rf $OUT/$shape
done
cat $OUT/up256.txt $OUT/up512.txt $OUT/up32.txt > $OUT/pmc_tconv.txt
grep -v "^p[123] SQ_INSTS\|^p[123] SQ_ACTIVE" $OUT/pmc_tconv.txt
