#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r4lc2
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests/test_gpu_prover.py -m gpu -x -q > $OUT/tests.log 2>&1
grep -E "passed|failed" $OUT/tests.log | tail -2
for cfg in k16 k19; do
  timeout 300 python bench.py --config $cfg --steps 4 --warmup 1 --streams 1 --transcript blake2b --steady-seconds 0 --no-cpu-baseline > $OUT/${cfg}.json 2> $OUT/${cfg}.err
  echo "$cfg $(grep -o '"ms_per_step": [0-9.]*' $OUT/${cfg}.json)"
done
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/k13_wave$i.json 2> $OUT/k13_wave$i.err
echo "k13 wave $(grep -o '"value": [0-9.]*' $OUT/k13_wave$i.json | head -1) $(grep -o '"steady_state_proofs_per_s": [0-9.]*' $OUT/k13_wave$i.json)"
done
timeout 300 python bench.py --steps 96 --warmup 16 --no-cpu-baseline --steady-seconds 0 > $OUT/k13_96.json 2> $OUT/k13_96.err
echo "k13 96 $(grep -o '"value": [0-9.]*' $OUT/k13_96.json | head -1)"
timeout 300 python bench.py --steps 8 --warmup 2 --streams 1 --transcript blake2b --no-cpu-baseline --steady-seconds 0 > $OUT/k13_single.json 2> $OUT/k13_single.err
echo "k13 single $(grep -o '"ms_per_step": [0-9.]*' $OUT/k13_single.json | head -1)"
