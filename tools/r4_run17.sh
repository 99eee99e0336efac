mkdir -p gpurun_out/r4p
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r4p/gputest.log
timeout 600 python bench.py --steps 20 --warmup 3 2>&1 | grep '^{' > gpurun_out/r4p/bench_default.json; cut -c1-200 gpurun_out/r4p/bench_default.json
timeout 600 python bench.py --preset c4 --steps 10 --warmup 2 2>&1 | grep '^{' > gpurun_out/r4p/bench_c4.json; cut -c1-200 gpurun_out/r4p/bench_c4.json
timeout 900 python bench.py --preset c5 2>&1 | grep '^{' > gpurun_out/r4p/bench_c5.json; cut -c1-200 gpurun_out/r4p/bench_c5.json
