#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r4k19
mkdir -p $OUT
cd $REPO
for cfg in k16 k19; do
  ZKFHE_TRACE=1 ZKFHE_TRACE0=1 python bench.py --config $cfg --steps 2 --warmup 1 --streams 1 --transcript blake2b --steady-seconds 0 --no-cpu-baseline > $OUT/${cfg}.json 2> $OUT/${cfg}.trace
done
