#!/bin/bash
# rocprofv3 kernel trace of stage A (e4e encoder) alone; usage (GPU box, repo root): bash tools/prof_encoder.sh
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_encoder
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT -o k -- python3 tools/run_encoder.py 4 > $OUT/run.log 2>&1
DB=$(ls $OUT/*.db $OUT/*/*.db 2>/dev/null | head -1)
python3 tools/rocpd_summary.py $DB $OUT/kernel_stats.md > /dev/null
rm -f $OUT/*.db $OUT/*/*.db
tail -2 $OUT/run.log
