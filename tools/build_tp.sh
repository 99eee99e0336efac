#!/bin/bash
# tuning build of the persistent chain with phase counters: build/abl/libvspbfr_tp.so   (usage: bash tools/build_tp.sh [extra -D flags])
set -e
cd "$(dirname "$0")/.."
mkdir -p build/abl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-gpu-rdc -DVSP_TP_TIMING "$@" -c vspbfr_amd/csrc/tacc_persist.hip -o build/abl/tacc_persist_tp.o
objs=$(ls build/csrc/*.o | grep -v tacc_persist.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/abl/libvspbfr_tp.so $objs build/abl/tacc_persist_tp.o
