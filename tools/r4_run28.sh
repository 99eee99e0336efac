mkdir -p gpurun_out/r4y
for w in 0 1; do echo "VSP_WINO_RO_WIDE=$w"; VSP_WINO_RO_WIDE=$w timeout 300 python tools/bench_wino.py plain 2>&1 | grep "@"; done | tee gpurun_out/r4y/wide_tile.log
VSP_WINO_RO_WIDE=1 timeout 900 python -m pytest tests/test_hip_ops.py -q -x -k "winograd" 2>&1 | tail -3
