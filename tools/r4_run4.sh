export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4e
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/p -o k -- python3 $GRAFT_REPO_ROOT/tools/bench_wino4.py 8,512,512,64 8,256,256,128 8,512,512,32 > $OUT/bench.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(ls $OUT/p/*.db $OUT/p/*/*.db 2>/dev/null | head -1)
python3 tools/rocpd_summary.py $DB $OUT/kernel_stats.md > /dev/null
head -30 $OUT/kernel_stats.md; cat $OUT/bench.log | grep -v "^/opt"
