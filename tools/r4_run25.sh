mkdir -p gpurun_out/r4v
timeout 600 python bench.py --preset c4 --steps 10 --warmup 2 2>&1 | grep '^{' > gpurun_out/r4v/bench_c4.json; cut -c1-200 gpurun_out/r4v/bench_c4.json
timeout 900 python bench.py --preset c5 2>&1 | grep '^{' > gpurun_out/r4v/bench_c5.json; cut -c1-200 gpurun_out/r4v/bench_c5.json
timeout 1500 python -m pytest tests/test_hip_models.py -q -x -k "c4 or c5 or training" 2>&1 | tail -4
