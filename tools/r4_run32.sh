mkdir -p gpurun_out/r4ab
for n in 0; do echo "VSP_FIR_NTB=$n (0 = persistent)"; VSP_FIR_NTB=$n timeout 300 python tools/bench_fir.py 2>&1 | grep -v "^/opt"; done | tee gpurun_out/r4ab/bench_fir_persistent2.log
