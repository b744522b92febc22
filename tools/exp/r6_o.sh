#!/bin/bash
# round 6, GPU pass 14: the wave's stragglers (a proof whose phase-0 commitment comes back after 50 ms: profiles/r6_probes.md section 7) against the
# number of hardware queues: 16 / 20 / 21 / 22 / 23, six runs each, alternating
set -u
OUT=gpurun_out/r6o; mkdir -p $OUT
for rep in 1 2 3 4 5 6; do
  for q in 16 20 21 22 23; do
    GPU_MAX_HW_QUEUES=$q ZKFHE_TRACE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic-pass --steady-seconds 0 > $OUT/wave_q${q}_$rep.json 2> $OUT/wave_q${q}_$rep.err
  done
done
python tools/exp/wave_trace.py $OUT
