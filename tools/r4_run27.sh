mkdir -p gpurun_out/r4x
for nb in 2 1; do echo "VSP_WINO_RO_NB=$nb"; VSP_WINO_RO_NB=$nb timeout 300 python tools/bench_wino.py plain 2>&1 | grep "@"; done | tee gpurun_out/r4x/half_tile.log
timeout 900 python -m pytest tests/test_hip_ops.py -q -x -k "winograd" 2>&1 | tail -3
