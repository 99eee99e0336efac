#!/bin/bash
# phase times of the persistent chain under the projection ablations (VSP_TP_ABL).  build: bash tools/tp_ablate.sh build; run on the GPU box without arguments
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  for a in 0 1 2 4; do bash tools/build_tp.sh -DVSP_TP_ABL=$a && mv build/abl/libvspbfr_tp.so build/abl/libvspbfr_tp$a.so; done
  exit 0
fi
for a in 0 1 2 4; do echo "VSP_TP_ABL=$a"; VSPBFR_HIP_LIB=$PWD/build/abl/libvspbfr_tp$a.so timeout 200 python tools/tacc_phase_times.py ${1:-4} 2>&1 | grep cluster; done
