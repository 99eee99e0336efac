#!/bin/bash
# Round 5: the evidence runs behind the final numbers (GPU box, from the repo root): bench lines of the four configurations, per-layer
# conv breakdowns, kernel-trace summaries of the serial steps, PMC passes on the new kernels, HBM traffic of the conv family.
# usage: tools/r5_evidence.sh <tag>      -> gpurun_out/r5ev_<tag>/
TAG=${1:-final}
OUT=gpurun_out/r5ev_$TAG
mkdir -p $OUT
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_default.json
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_default_2.json
python bench.py --preset c3 2>/dev/null | tail -1 > $OUT/bench_c3.json
python bench.py --preset c4 2>/dev/null | tail -1 > $OUT/bench_c4.json
python bench.py --preset c5 2>/dev/null | tail -1 > $OUT/bench_c5.json
python tools/conv_breakdown.py 2>&1 | grep -v amdgpu.ids > $OUT/conv_breakdown.log
B=16 T=25 python tools/conv_breakdown_c3.py 2>&1 | grep -v amdgpu.ids > $OUT/conv_breakdown_c3.log
bash tools/prof_bench.sh r5ev_serial --no-overlap > /dev/null 2>&1
cp gpurun_out/prof_r5ev_serial/kernel_stats.md $OUT/kernel_stats_b8_t50_serial.md
bash tools/prof_bench.sh r5ev_c3_serial --no-overlap --preset c3 > /dev/null 2>&1
cp gpurun_out/prof_r5ev_c3_serial/kernel_stats.md $OUT/kernel_stats_c3_serial.md
bash tools/rs_probe.sh $OUT/pmc_dil64 64 64 512 4 > $OUT/pmc_dil64.txt 2>&1
WINO_FORM=0 bash tools/pmc_wino.sh $OUT/pmc_dil128 128 128 256 4 > $OUT/pmc_dil128.txt 2>&1
WINO_FORM=0 bash tools/pmc_wino.sh $OUT/pmc_dil512 512 512 64 4 > $OUT/pmc_dil512.txt 2>&1
WINO=5 bash tools/pmc_wino.sh $OUT/pmc_f4f_64 64 64 512 > $OUT/pmc_f4f_64.txt 2>&1
WINO=5 bash tools/pmc_wino.sh $OUT/pmc_f4f_32 32 32 1024 > $OUT/pmc_f4f_32.txt 2>&1
bash tools/pmc_f4f.sh $OUT/pmc_f4f_64b 64 64 512 > $OUT/pmc_f4f_64_issue.txt 2>&1
bash tools/pmc_bench.sh $OUT/traffic > $OUT/traffic.log 2>&1
rm -rf gpurun_out/prof_r5ev_serial/*.db gpurun_out/prof_r5ev_c3_serial/*.db
ls $OUT
