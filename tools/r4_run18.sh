mkdir -p gpurun_out/r4q
for f in "" "--graphs"; do timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $f 2>&1 | grep '^{' | cut -c1-170; done | tee gpurun_out/r4q/graphs.log
