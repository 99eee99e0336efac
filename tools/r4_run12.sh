mkdir -p gpurun_out/r4l
for n in 1 2; do echo "VSP_FIR_NTB=$n"; VSP_FIR_NTB=$n timeout 300 python tools/bench_fir.py 2>&1 | grep -v "^/opt" | grep bf16; done | tee gpurun_out/r4l/bench_fir_bf16.log
for n in 1 2; do VSP_FIR_NTB=$n timeout 600 python bench.py --preset c3 --steps 10 --warmup 2 2>&1 | grep '^{' | cut -c1-160; done | tee gpurun_out/r4l/bench_c3_ntb.log
