#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r4streams2
mkdir -p $OUT
cd $REPO
for s in 20 10 12 14 16 20 10 12; do
  timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --streams $s --steady-seconds 0 > $OUT/s$s.json 2> $OUT/s$s.err
  echo "streams=$s $(grep -o '"value": [0-9.]*' $OUT/s$s.json | head -1)"
done
