mkdir -p gpurun_out/r4aa
( bash tools/pmc_fir.sh gpurun_out/r4aa/bf 16 32 1024 bf16; bash tools/pmc_fir.sh gpurun_out/r4aa/f32 8 32 1024 ) 2>&1 | grep -v "^/opt" | tee gpurun_out/r4aa/pmc_fir.txt
