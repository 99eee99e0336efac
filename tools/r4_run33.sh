mkdir -p gpurun_out/r4ac
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r4ac/gputest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/r4ac/smoke.log
timeout 600 python bench.py --steps 20 --warmup 3 2>&1 | grep '^{' > gpurun_out/r4ac/bench_default.json; cut -c1-200 gpurun_out/r4ac/bench_default.json
timeout 600 python bench.py --preset c3 --steps 10 --warmup 2 2>&1 | grep '^{' > gpurun_out/r4ac/bench_c3.json; cut -c1-200 gpurun_out/r4ac/bench_c3.json
timeout 600 python bench.py --preset c4 --steps 10 --warmup 2 2>&1 | grep '^{' > gpurun_out/r4ac/bench_c4.json; cut -c1-200 gpurun_out/r4ac/bench_c4.json
