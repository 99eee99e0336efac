#!/bin/bash
# round 3: the evidence runs behind the final numbers (GPU box): kernel-trace summaries of the serial and the overlapped step, PMC
# passes of the pipelined conv kernels on their layers, HBM traffic of the conv family, the default bench line
mkdir -p gpurun_out/r3_ev
bash tools/prof_bench.sh r3_serial_final --no-overlap > /dev/null 2>&1
bash tools/prof_bench.sh r3_overlap_final > /dev/null 2>&1
( bash tools/pmc_pipe.sh gpurun_out/r3_ev/pmc_stem stem 4x2x2x4x4k1p3o1r5
  bash tools/pmc_pipe.sh gpurun_out/r3_ev/pmc_down64 down64 4x2x2x4x4k1p3o2r5
  bash tools/pmc_pipe.sh gpurun_out/r3_ev/pmc_up128 up128 1x8x2x4x8k1p3o2r5t
  bash tools/pmc_pipe.sh gpurun_out/r3_ev/pmc_dil512 dil512 4x2x1x8x4k1p3o4r8fd ) > gpurun_out/r3_ev/pmc_pipe.txt 2>&1
bash tools/pmc_bench.sh gpurun_out/r3_ev/traffic > gpurun_out/r3_ev/traffic.log 2>&1
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r3_ev/bench_default_steps20.json
python bench.py 2>/dev/null | tail -1 > gpurun_out/r3_ev/bench_default.json
ls gpurun_out/r3_ev gpurun_out/prof_r3_serial_final gpurun_out/prof_r3_overlap_final
