mkdir -p gpurun_out/r4k
( WINO=4 bash tools/pmc_wino.sh gpurun_out/r4k/w4_512 512 512 64
  WINO=4 bash tools/pmc_wino.sh gpurun_out/r4k/w4_256 256 256 128 ) 2>&1 | grep -v "^/opt" | tee gpurun_out/r4k/pmc_wino4.txt | grep -E "==|busy|conflict|waiting|duration"
