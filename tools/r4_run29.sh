mkdir -p gpurun_out/r4z
timeout 300 python tools/bench_fir.py 2>&1 | grep -v "^/opt" | tee gpurun_out/r4z/bench_fir.log
timeout 600 python -m pytest tests/test_hip_ops.py -q -x -k "blur or fir or upfirdn" 2>&1 | tail -3
