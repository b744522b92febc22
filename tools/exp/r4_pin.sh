#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r4pin
mkdir -p $OUT
cd $REPO
for i in 1 2 3 4; do
for p in 0 1; do
  ZKFHE_HASH_PIN=$p timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --steady-seconds 0 > $OUT/p${p}_$i.json 2> $OUT/p${p}_$i.err
  python3 -c "
import json; d=json.loads(open('$OUT/p${p}_$i.json').read().strip().splitlines()[-1]); c=d['config']; print('pin=$p', round(d['value'],1), round(c['per_proof_latency_ms']['witness_host'],1), round(c['host_cpu_ms_per_proof'],1))"
done
done
