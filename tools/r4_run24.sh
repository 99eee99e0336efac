mkdir -p gpurun_out/r4u
timeout 600 python tools/bench_wino4.py 16,512,512,32 16,256,256,32 16,256,512,32 16,128,128,64 16,256,256,64 16,512,512,64 16,128,256,64 4,512,512,64 4,256,256,128 4,128,128,256 2>&1 | grep -v "^/opt" | cut -c1-260 | tee gpurun_out/r4u/bench_wino4_b16.log
