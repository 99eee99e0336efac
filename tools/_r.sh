mkdir -p gpurun_out/r4af
timeout 300 python tools/bench_wino.py 2>&1 | grep "@" | tee gpurun_out/r4af/ro_lean2.log
timeout 900 python -m pytest tests/test_hip_ops.py -q -x -k "winograd" 2>&1 | tail -3
