#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r4suite3
mkdir -p $OUT
cd $REPO
timeout 1800 python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1
grep -E "passed|failed" $OUT/tests.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"
for cfg in k16 k19; do
  timeout 300 python bench.py --config $cfg --steps 4 --warmup 1 --streams 1 --transcript blake2b --steady-seconds 0 --no-cpu-baseline > $OUT/${cfg}.json 2> $OUT/${cfg}.err
  echo "$cfg $(grep -o '"ms_per_step": [0-9.]*' $OUT/${cfg}.json)"
done
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/k13_wave.json 2> $OUT/k13_wave.err
echo "k13 wave $(grep -o '"value": [0-9.]*' $OUT/k13_wave.json | head -1) $(grep -o '"steady_state_proofs_per_s": [0-9.]*' $OUT/k13_wave.json)"
