#!/bin/bash
# Round 5: ablation series + PMC passes of the register-resident-U Winograd kernel (conv_wino_rs.hip) on one layer.
# usage: tools/rs_probe.sh <outdir> Cin Cout S G      (needs build/abl/libvspbfr_rsabl.so: tools/build_abl.sh conv_wino_rs.hip VSP_WINO_ABLATE libvspbfr_rsabl)
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
OUT=$1; shift
mkdir -p $OUT
export WINO_FORM=3
if [ -f build/abl/libvspbfr_rsabl.so ]; then
  for dbg in 0 1 2 3 4 8 12 16 32 64 15 31 47; do
    VSPBFR_HIP_LIB=build/abl/libvspbfr_rsabl.so VSP_TUNE=1 VSP_CONV_DBG=$dbg timeout 120 python3 tools/wino_ablate.py $* 2>&1 | grep -v amdgpu.ids
  done > $OUT/ablate.log
  cat $OUT/ablate.log
fi
WINO=2 bash tools/pmc_wino.sh $OUT $* > $OUT/pmc.txt 2>&1
P="python3 tools/run_one_wino.py $*"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum -d $OUT/p4 -o p4 --output-format csv -- $P > $OUT/p4.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_MISS_sum -d $OUT/p5 -o p5 --output-format csv -- $P > $OUT/p5.log 2>&1
python3 - <<PY >> $OUT/pmc.txt
import csv, collections, glob
for pth in ("p4", "p5"):
    fs = glob.glob("$OUT/%s/**/*counter_collection.csv" % pth, recursive=True)
    if not fs: print(pth, "no csv"); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "wino" in r["Kernel_Name"] and "weight" not in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items(): print(pth, k, "%.5g" % (sum(v) / len(v)), "(FETCH_SIZE / WRITE_SIZE in KiB; FETCH x2 on gfx950 for wide coalesced reads)")
PY
cat $OUT/pmc.txt | grep -v "^p[123] .*INST\|no csv" | tail -40
rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 $OUT/p5
