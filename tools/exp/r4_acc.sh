#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r4acc
mkdir -p $OUT
cd $REPO
run() {
  local name=$1; shift
  env "$@" timeout 300 python bench.py --config k19 --steps 4 --warmup 1 --streams 1 --transcript blake2b --steady-seconds 0 --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.err
  python3 -c "
import json; d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],1), d['roofline'].get('avg_launch_ms'), d['roofline']['int_alu']['frac'])"
}
run base A=1
run e32 ZKFHE_TASK_E=32
run e48 ZKFHE_TASK_E=48
run b16 ZKFHE_ACC_BLOCKS=16
run b64 ZKFHE_ACC_BLOCKS=64
run base2 A=1
