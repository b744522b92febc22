#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r3q5
mkdir -p $OUT
cd $REPO
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "msm" 2>&1 | tail -3 > $OUT/tests.log
for wl in "96 full" "240 small" "240 mixed"; do
  echo "== $wl" >> $OUT/msm_calls.txt
  BITS=13 python tools/exp/msm_table_bench.py 13 $wl 2>/dev/null >> $OUT/msm_calls.txt
done
ZKFHE_DEBUG_NPART=1 python bench.py --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --steady-seconds 0 --transcript blake2b > /dev/null 2> $OUT/npart.log
python bench.py --steps 8 --streams 1 --transcript blake2b --no-cpu-baseline --steady-seconds 0 > $OUT/bench_single_blake2b.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2>/dev/null
python bench.py --transcript blake2b --no-cpu-baseline > $OUT/bench_blake2b.json 2>/dev/null
