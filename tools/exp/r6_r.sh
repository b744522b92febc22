#!/bin/bash
# round 6, GPU pass 17: the tail of the driver's wave with finer marks in the last phase (evaluations back, evaluations absorbed, SHPLONK h committed)
set -u
OUT=gpurun_out/r6r; mkdir -p $OUT
for rep in 1 2 3 4; do
  ZKFHE_TRACE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic-pass --steady-seconds 0 > $OUT/wave_$rep.json 2> $OUT/wave_$rep.err
done
ZKFHE_TRACE=1 python bench.py --steps 4 --streams 1 --no-cpu-baseline --no-traffic-pass --steady-seconds 0 --announce off > $OUT/single.json 2> $OUT/single.err
python tools/exp/wave_trace.py $OUT | cut -c1-200
