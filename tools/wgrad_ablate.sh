#!/bin/bash
# Ablation builds of the weight-gradient kernel (VSP_WG_ABL: 1 no result stores, 2 no MFMAs, 4 no staging after the first chunk) on the
# training shapes.  Builds variant libraries under build/abl/ HERE (they travel with the snapshot) when run with `build`, measures on the
# GPU box otherwise.  usage: bash tools/wgrad_ablate.sh build; gpurun -- bash tools/wgrad_ablate.sh
set -e
cd "$(dirname "$0")/.."
VARS="0 1 2 4 6"
if [ "$1" = build ]; then
  mkdir -p build/abl
  for v in $VARS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-gpu-rdc -DVSP_WG_ABL=$v ${WG_EXTRA} \
      -c vspbfr_amd/csrc/conv_wgrad.hip -o build/abl/conv_wgrad_$v.o
    objs=$(ls build/csrc/*.o | grep -v conv_wgrad.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/abl/libvspbfr_abl$v.so $objs build/abl/conv_wgrad_$v.o
  done
  exit 0
fi
mkdir -p gpurun_out
for v in $VARS; do
  echo "== VSP_WG_ABL=$v"
  VSPBFR_HIP_LIB=$PWD/build/abl/libvspbfr_abl$v.so python tools/bench_wgrad.py
done 2>&1 | tee gpurun_out/wgrad_ablate.log
