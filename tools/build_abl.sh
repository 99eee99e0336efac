#!/bin/bash
# Tuning build: the library with ONE source compiled with an ablation macro (its VSP_CONV_DBG switches), as build/abl/<name>.so.
# usage: tools/build_abl.sh <source.hip> <MACRO> <name>      then   VSPBFR_HIP_LIB=build/abl/<name>.so VSP_CONV_DBG=<bits> python tools/...
set -e
cd "$(dirname "$0")/.."
src=$1; macro=$2; name=$3
mkdir -p build/abl
make -C vspbfr_amd/csrc -j8 > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -D$macro $(sed -n "s/^\$(OBJDIR)\/${src%.hip}.o: CXXFLAGS += //p" vspbfr_amd/csrc/Makefile) -c vspbfr_amd/csrc/$src -o build/abl/${src%.hip}.o
objs=$(ls build/csrc/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/abl/$name.so $objs build/abl/${src%.hip}.o
echo build/abl/$name.so
