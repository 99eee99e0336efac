#!/bin/bash
# Round 6: the evidence runs behind the final numbers (GPU box, from the repo root): bench lines of the configurations (the default line carries
# configs[2] and the configs[3] share as extra_configs), per-layer conv breakdowns, kernel-trace summaries of the serial steps, what the side
# stream costs (tools/bench_hidden_cost.py, overlap_timeline.py, bench_side_load.py), PMC passes on the bf16 kernel with per-image weights,
# HBM traffic of the conv family for configs[1] and configs[2].
# usage: tools/r6_evidence.sh <tag>      -> gpurun_out/r6ev_<tag>/
TAG=${1:-final}
OUT=gpurun_out/r6ev_$TAG
mkdir -p $OUT
python bench.py --steps 20 2>/dev/null | tail -1 > $OUT/bench_default.json
python bench.py --steps 20 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_default_2.json
python bench.py --preset c3 --steps 20 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_c3.json
python bench.py --preset c4 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_c4.json
python bench.py --preset c5 2>/dev/null | tail -1 > $OUT/bench_c5.json
python tools/conv_breakdown.py 2>&1 | grep -v amdgpu.ids > $OUT/conv_breakdown.log
python tools/conv_breakdown_c3.py 2>&1 | grep -v amdgpu.ids > $OUT/conv_breakdown_c3.log
python tools/bench_hidden_cost.py --preset c2 2>/dev/null > $OUT/hidden_cost_c2.json
python tools/bench_hidden_cost.py --preset c3 2>/dev/null > $OUT/hidden_cost_c3.json
for m in h b ab; do python tools/overlap_timeline.py --split $m 2>/dev/null > $OUT/timeline_c2_$m.txt; done
python tools/overlap_timeline.py --preset c3 --split h 2>/dev/null > $OUT/timeline_c3_h.txt
python tools/bench_side_load.py 2>/dev/null > $OUT/side_load.txt
bash tools/prof_bench.sh r6ev_serial --no-overlap --no-extra > /dev/null 2>&1
cp gpurun_out/prof_r6ev_serial/kernel_stats.md $OUT/kernel_stats_b8_t50_serial.md
bash tools/prof_bench.sh r6ev_c3_serial --no-overlap --preset c3 > /dev/null 2>&1
cp gpurun_out/prof_r6ev_c3_serial/kernel_stats.md $OUT/kernel_stats_c3_serial.md
B=16 IO_BF16=1 bash tools/pmc_bf16.sh $OUT/pmc_bf16_512 512 512 64 0 > $OUT/pmc_bf16_512_modw.txt 2>&1
B=16 IO_BF16=1 VSP_TUNE=1 VSP_BF16_MODW=0 bash tools/pmc_bf16.sh $OUT/pmc_bf16_512s 512 512 64 0 > $OUT/pmc_bf16_512_shared.txt 2>&1
B=16 IO_BF16=1 bash tools/pmc_bf16.sh $OUT/pmc_bf16_g128 128 128 256 0 4 > $OUT/pmc_bf16_dil128_modw.txt 2>&1
B=16 IO_BF16=1 VSP_TUNE=1 VSP_BF16_MODW=0 bash tools/pmc_bf16.sh $OUT/pmc_bf16_g128s 128 128 256 0 4 > $OUT/pmc_bf16_dil128_shared.txt 2>&1
bash tools/pmc_bench.sh $OUT/traffic > $OUT/traffic.log 2>&1
bash tools/pmc_bench.sh $OUT/traffic_c3 --preset c3 > $OUT/traffic_c3.log 2>&1
rm -rf gpurun_out/prof_r6ev_serial/*.db gpurun_out/prof_r6ev_c3_serial/*.db $OUT/pmc_*/p*/ $OUT/traffic*/fetch $OUT/traffic*/write $OUT/traffic*/cal_*
ls $OUT
