#!/bin/bash
# HBM traffic of the conv kernel family over one bench step (GPU box): separate PMC passes for FETCH_SIZE and WRITE_SIZE
# (TCC slots do not fit both), plus a calibration pass on kernels with known byte counts.
# usage: tools/pmc_bench.sh <outdir> [extra bench.py arguments, e.g. --conv-dtype bf16]
set -e
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=$1; shift; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o f --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-overlap "$@" > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o w --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-overlap "$@" > $OUT/write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/cal_fetch -o f --output-format csv -- python3 tools/pmc_calibrate.py > $OUT/cal_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/cal_write -o w --output-format csv -- python3 tools/pmc_calibrate.py > $OUT/cal_write.log 2>&1
python3 tools/pmc_traffic_summary.py $OUT
rm -f $OUT/*/*.db $OUT/*/*/*.db
