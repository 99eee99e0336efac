mkdir -p gpurun_out/r4ad
for shape in "16 64 64 512" "16 32 32 1024" "16 128 128 256"; do
  for dbg in 0 1 2 4 6 7 1048576 2097152 3145728; do
    VSP_CONV_DBG=$dbg VSPBFR_HIP_LIB=$PWD/build/abl/libvspbfr_bf16abl.so timeout 120 python tools/bf16_lowch_ablate.py $shape 2>&1 | grep dbg
  done
done | tee gpurun_out/r4ad/bf16_lowch_ablate.log
