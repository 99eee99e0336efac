mkdir -p gpurun_out/r4t
for n in 0 1; do echo "VSP_FIR_DWORDS=$n"; VSP_FIR_DWORDS=$n timeout 300 python tools/bench_fir.py 2>&1 | grep -v "^/opt" | grep bf16; done | tee gpurun_out/r4t/bench_fir_bf16.log
timeout 600 python -m pytest tests/test_hip_ops.py -q -x -k "blur or fir or upfirdn" 2>&1 | tail -3
