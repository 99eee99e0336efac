mkdir -p gpurun_out/r4s
timeout 900 python tools/conv_breakdown_c3.py 2>&1 | grep -v "^/opt" | tee gpurun_out/r4s/conv_breakdown_c3.log | head -45
