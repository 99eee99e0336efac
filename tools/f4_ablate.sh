#!/bin/bash
# Round 5: the fused F(4x4) kernel (conv_wino4f.hip) under its compile-time ablation switches; build the variants first:
#   for v in 1 2 4 8 16 32 5 21 29; do tools/build_abl.sh conv_wino4f.hip VSP_F4_ABL=$v f4abl_$v; done
# usage: tools/f4_ablate.sh <shape B,Cin,Cout,S> ...
cd "$(dirname "$0")/.."
echo "production:"; FORMS=F4f python3 tools/bench_wino4f.py "$@" 2>&1 | grep -v amdgpu.ids
for v in 1 2 4 8 16 32 5 21 29; do
  [ -f build/abl/f4abl_$v.so ] || continue
  echo "VSP_F4_ABL=$v (1 windows from L1, 2 no U staging, 4 no MFMAs, 8 no transform, 16 no epilogue, 32 no barrier):"
  VSPBFR_HIP_LIB=build/abl/f4abl_$v.so FORMS=F4f timeout 300 python3 tools/bench_wino4f.py "$@" 2>&1 | grep -v amdgpu.ids
done
