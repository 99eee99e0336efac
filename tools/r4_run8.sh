mkdir -p gpurun_out/r4j
timeout 900 python tools/conv_breakdown.py 2>&1 | grep -v "^/opt" | tee gpurun_out/r4j/conv_breakdown.log | head -50
