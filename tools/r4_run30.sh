mkdir -p gpurun_out/r4z
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | grep '^{' | cut -c1-170 | tee gpurun_out/r4z/bench_default.log
timeout 600 python bench.py --preset c3 --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | grep '^{' | cut -c1-170 | tee gpurun_out/r4z/bench_c3.log
