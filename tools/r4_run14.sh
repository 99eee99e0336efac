mkdir -p gpurun_out/r4m
for dbg in 0 4096 8192 64; do echo "VSP_CONV_DBG=$dbg"; VSP_CONV_DBG=$dbg timeout 300 python tools/bench_wino4.py 8,512,512,64 8,256,256,128 8,128,128,256 2>&1 | grep B8 | cut -c1-250; done | tee gpurun_out/r4m/prio.log
for dbg in 0 4096; do echo "VSP_CONV_DBG=$dbg"; VSP_CONV_DBG=$dbg timeout 300 python tools/bench_wino.py plain 2>&1 | grep "@"; done | tee -a gpurun_out/r4m/prio.log
