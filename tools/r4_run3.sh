mkdir -p gpurun_out/r4d
timeout 600 python tools/bench_wino4.py 2>&1 | tee gpurun_out/r4d/bench_wino4.log
