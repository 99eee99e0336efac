#!/bin/bash
# rocprofv3 kernel trace of bench.py; usage: tools/prof_bench.sh <tag> <bench args...>   (run on the GPU box from the repo root)
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o k -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/bench.log 2>&1
grep "^{\"metric\"" $OUT/bench.log | tail -1 > $OUT/bench_line.json
DB=$(ls $OUT/*.db $OUT/*/*.db 2>/dev/null | head -1)
python3 tools/rocpd_summary.py $DB $OUT/kernel_stats.md > /dev/null
ls $OUT
