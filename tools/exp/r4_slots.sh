#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r4slots
mkdir -p $OUT
cd $REPO
for i in 1 2 3; do
for sl in 0 16 14 12; do
  ZKFHE_HASH_SLOTS=$sl timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --steady-seconds 0 > $OUT/s${sl}_$i.json 2> $OUT/s${sl}_$i.err
  echo "slots=$sl $(grep -o '"value": [0-9.]*' $OUT/s${sl}_$i.json | head -1)"
done
done
