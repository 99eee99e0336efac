mkdir -p gpurun_out/r4i
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/r4i/gputest.log
timeout 600 python bench.py --steps 20 --warmup 3 2>&1 | grep '^{' | tee gpurun_out/r4i/bench_default.json
