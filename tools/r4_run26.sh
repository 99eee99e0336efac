mkdir -p gpurun_out/r4w
bash tools/prof_bench.sh r4_end_serial --no-overlap > /dev/null 2>&1
bash tools/prof_bench.sh r4_end_overlap > /dev/null 2>&1
head -30 gpurun_out/prof_r4_end_serial/kernel_stats.md | cut -c1-150
cut -c1-300 gpurun_out/prof_r4_end_overlap/bench_line.json
timeout 900 python tools/conv_breakdown.py 2>&1 | grep -v "^/opt" > gpurun_out/r4w/conv_breakdown.log; tail -3 gpurun_out/r4w/conv_breakdown.log
timeout 600 python bench.py --steps 20 --warmup 3 2>&1 | grep '^{' > gpurun_out/r4w/bench_default.json; cut -c1-200 gpurun_out/r4w/bench_default.json
