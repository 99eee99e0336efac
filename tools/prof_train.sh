#!/bin/bash
# rocprofv3 kernel trace of one training iteration; usage: tools/prof_train.sh <tag> [B] [iters]   (on the GPU box, repo root)
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT -o k -- python3 tools/bench_train_step.py "$@" > $OUT/run.log 2>&1
DB=$(ls $OUT/*.db $OUT/*/*.db 2>/dev/null | head -1)
python3 tools/rocpd_summary.py $DB $OUT/kernel_stats.md > /dev/null
grep "^{" $OUT/run.log | tail -1 > $OUT/line.json
rm -f $OUT/*.db $OUT/*/*.db   # (tens of MB: gpurun copies at most 64 MiB back)
ls $OUT
