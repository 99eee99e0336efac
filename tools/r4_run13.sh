bash tools/pmc_bench.sh gpurun_out/r4_traffic_c2 > gpurun_out/r4_traffic_c2.log 2>&1
tail -40 gpurun_out/r4_traffic_c2.log
rm -rf gpurun_out/r4_traffic_c2/fetch gpurun_out/r4_traffic_c2/write gpurun_out/r4_traffic_c2/cal_fetch gpurun_out/r4_traffic_c2/cal_write
