mkdir -p gpurun_out/r4f
for ro in 0 1 2; do VSP_WINO_RO=$ro timeout 300 python tools/bench_wino.py > gpurun_out/r4f/bench_wino_ro$ro.log 2>&1; done
tail -n 16 gpurun_out/r4f/bench*.log
timeout 600 python -m pytest tests/test_hip_ops.py -q -x -k "winograd" 2>&1 | tail -3
timeout 300 python -m pytest tests/test_layout.py -q -x 2>&1 | tail -3
