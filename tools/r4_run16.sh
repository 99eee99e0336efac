mkdir -p gpurun_out/r4o
for dbg in 0 16384; do echo "VSP_CONV_DBG=$dbg"; VSP_CONV_DBG=$dbg timeout 300 python tools/bench_wino.py plain 2>&1 | grep "@"; done | tee gpurun_out/r4o/quad_epi.log
timeout 900 python -m pytest tests/test_hip_ops.py -q -x -k "winograd or conv2d_packed or epilogue or prologue" 2>&1 | tail -3
