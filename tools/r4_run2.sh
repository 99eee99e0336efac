mkdir -p gpurun_out/r4c
for ro in 1 2 4; do VSP_WINO_RO=$ro timeout 300 python tools/bench_wino.py plain > gpurun_out/r4c/bench_wino_ro$ro.log 2>&1; done
tail -n 10 gpurun_out/r4c/bench*.log
for shape in "512 512 64" "64 64 512"; do
  for dbg in 0 1 2 4 64 128 8 16 32 6 63 62; do
    VSP_WINO_RO=1 VSP_CONV_DBG=$dbg VSPBFR_HIP_LIB=$PWD/build/abl/libvspbfr_roabl.so timeout 120 python tools/wino_ablate.py $shape 2>&1 | grep dbg
  done
done | tee gpurun_out/r4c/ablate.log
