#!/bin/bash
# round 3: Winograd kernel after the LDS re-pitch -- parity tests, layer timings, PMC on the three judged shapes (GPU box)
mkdir -p gpurun_out/r3_wino
python -m pytest tests/test_hip_ops.py -q -m gpu -k "winograd" -x > gpurun_out/r3_wino/pytest.log 2>&1; tail -3 gpurun_out/r3_wino/pytest.log
python tools/bench_wino.py > gpurun_out/r3_wino/bench_wino.log 2>&1; cat gpurun_out/r3_wino/bench_wino.log
for s in "512 512 64" "128 128 256" "64 64 512"; do
  bash tools/pmc_wino.sh gpurun_out/r3_wino/pmc_$(echo $s | tr ' ' '_') $s >> gpurun_out/r3_wino/pmc.txt 2>&1
done
cat gpurun_out/r3_wino/pmc.txt
