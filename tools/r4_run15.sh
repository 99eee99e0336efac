mkdir -p gpurun_out/r4n
for shape in "64 64 512" "32 32 1024" "512 512 64"; do
  for dbg in 0 512 63 575 62 574; do
    VSP_WINO_RO=1 VSP_CONV_DBG=$dbg VSPBFR_HIP_LIB=$PWD/build/abl/libvspbfr_roabl.so timeout 120 python tools/wino_ablate.py $shape 2>&1 | grep dbg
  done
done | tee gpurun_out/r4n/ablate_ro_epi.log
