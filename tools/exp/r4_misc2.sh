#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r4misc2
mkdir -p $OUT
cd $REPO
run() {
  local name=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline $ARGS > $OUT/$name.json 2> $OUT/$name.err
  echo "$name $(grep -o '"ms_per_step": [0-9.]*' $OUT/$name.json) $(grep -o '"value": [0-9.]*' $OUT/$name.json | head -1) $(grep -o '"steady_state_proofs_per_s": [0-9.]*' $OUT/$name.json)"
}
ARGS="--steps 20 --warmup 5"
for i in 1 2 3; do
  run wave_def_$i A=1
  run wave_bi8_$i ZKFHE_BI_CHUNK=8
done
ARGS="--config k19 --steps 4 --warmup 1 --streams 1 --transcript blake2b --steady-seconds 0"
run k19 A=1
ARGS="--config k19 --steps 3 --warmup 1 --streams 1 --steady-seconds 0"
run k19_poseidon A=1
ARGS="--config k16 --steps 6 --warmup 1 --streams 1 --steady-seconds 0"
run k16_poseidon A=1
