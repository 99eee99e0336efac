set -x
mkdir -p gpurun_out/r4a
for ro in 0 8 16; do VSP_WINO_RO=$ro timeout 300 python tools/bench_wino.py > gpurun_out/r4a/bench_wino_ro$ro.log 2>&1; done
tail -n 20 gpurun_out/r4a/*.log
