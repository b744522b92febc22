#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r4misc
mkdir -p $OUT
cd $REPO
run() { # name, env..., -- args
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steady-seconds 0 --no-cpu-baseline $ARGS > $OUT/$name.json 2> $OUT/$name.err
  echo "$name $(grep -o '"ms_per_step": [0-9.]*' $OUT/$name.json) $(grep -o '"value": [0-9.]*' $OUT/$name.json | head -1)"
}
ARGS="--config k19 --steps 4 --warmup 1 --streams 1 --transcript blake2b"
run k19_base A=1
run k19_early ZKFHE_EARLY_P1=1
run k19_bi8 ZKFHE_BI_CHUNK=8
ARGS="--config k16 --steps 6 --warmup 1 --streams 1 --transcript blake2b"
run k16_base A=1
run k16_early ZKFHE_EARLY_P1=1
run k16_bi8 ZKFHE_BI_CHUNK=8
ARGS="--steps 96 --warmup 16"
run k13_96_bi8 A=1
run k13_96_bi16 ZKFHE_BI_CHUNK=16
run k13_96_bi8b A=1
run k13_96_bi16b ZKFHE_BI_CHUNK=16
ARGS="--steps 8 --warmup 2 --streams 1 --transcript blake2b"
run k13_single_bi8 A=1
run k13_single_bi16 ZKFHE_BI_CHUNK=16
