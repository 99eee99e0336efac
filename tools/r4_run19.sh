mkdir -p gpurun_out/r4r
for rod in 0 1; do echo "VSP_WINO_ROD=$rod"; VSP_WINO_ROD=$rod timeout 300 python tools/bench_wino.py 2>&1 | grep "4x"; done | tee gpurun_out/r4r/rod.log
timeout 900 python -m pytest tests/test_hip_ops.py -q -x -k "winograd" 2>&1 | tail -3
