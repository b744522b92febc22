#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r4prefix
mkdir -p $OUT
cd $REPO
python -m pytest tests/test_gpu_prover.py -m gpu -x -q -k "config4 or config5 or k16 or k19" 2>&1 | grep -E "passed|failed|error" > $OUT/tests.log
for cfg in k16 k19; do
  python bench.py --config $cfg --steps 4 --warmup 1 --streams 1 --transcript blake2b --steady-seconds 0 --no-cpu-baseline > $OUT/${cfg}.json 2>/dev/null
done
python -c "
import json
for c in ('k16','k19'):
    d=json.load(open('gpurun_out/r4prefix/%s.json'%c)); print(c, round(d['ms_per_step'],2), d['config']['per_proof_latency_ms'])" > $OUT/summary.txt
