#!/bin/bash
# Run tools/rv_stress.py on the production library (reference results) and on every assembly variant under build/rvasm (GPU box).
# usage: tools/rv_bisect.sh <outfile> [N]
OUT=$1; N=${2:-8}
python tools/rv_stress.py 2 > /dev/null 2>&1
for so in build/rvasm/*.so; do
  echo "=== $(basename $so .so)" >> $OUT
  VSPBFR_HIP_LIB=$so timeout 300 python tools/rv_stress.py $N 2>/dev/null | grep -E -B1 "in_scale only|modulated" | grep -v "^--" | sed -E 's/; pixel pairs.*//; s/^build.rvasm.//' >> $OUT
done
cat $OUT
