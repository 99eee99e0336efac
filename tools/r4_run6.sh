mkdir -p gpurun_out/r4h
timeout 600 python tools/bench_wino4.py 8,512,512,64 8,256,256,128 8,512,512,32 8,128,128,256 2,64,96,48 3,20,40,36 2>&1 | grep -v "^/opt" | tee gpurun_out/r4h/bench_wino4.log
for shape in "8 512 512 64" "8 256 256 128"; do
  for dbg in 0 16 32 96 33 34 36 40 38; do
    VSP_CONV_DBG=$dbg VSPBFR_HIP_LIB=$PWD/build/abl/libvspbfr_roabl.so timeout 120 python tools/wino4_ablate.py $shape 2>&1 | grep dbg
  done
done | tee gpurun_out/r4h/ablate4.log
