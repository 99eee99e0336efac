bash tools/prof_bench.sh r4_c3_serial --preset c3 --no-overlap > /dev/null 2>&1
head -40 gpurun_out/prof_r4_c3_serial/kernel_stats.md | cut -c1-150
cat gpurun_out/prof_r4_c3_serial/bench_line.json | cut -c1-200
timeout 600 python bench.py --preset c3 --steps 10 --warmup 2 2>&1 | grep '^{' | tee gpurun_out/r4_bench_c3.json | cut -c1-200
