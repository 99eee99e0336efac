#!/bin/bash
# rocprofv3 kernel trace of the code_diffuser_train iteration; usage (GPU box, repo root): bash tools/prof_diffuser_train.sh [B]
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_diffuser_train
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT -o k -- python3 tools/bench_diffuser_train.py ${1:-16} > $OUT/run.log 2>&1
DB=$(ls $OUT/*.db $OUT/*/*.db 2>/dev/null | head -1)
python3 tools/rocpd_summary.py $DB $OUT/kernel_stats.md > /dev/null
rm -f $OUT/*.db $OUT/*/*.db
grep "^{" $OUT/run.log | tail -1 | cut -c1-200
