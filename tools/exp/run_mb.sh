ZKFHE_TRACE=1 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --steady-seconds 0 2> gpurun_out/trace20.log | tail -1 | cut -c1-90
grep -c trace gpurun_out/trace20.log
