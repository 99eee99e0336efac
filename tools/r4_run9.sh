bash tools/prof_bench.sh r4_serial --no-overlap > /dev/null 2>&1
head -45 gpurun_out/prof_r4_serial/kernel_stats.md | cut -c1-160
cat gpurun_out/prof_r4_serial/bench_line.json | cut -c1-300
