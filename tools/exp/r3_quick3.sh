#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r3q3
mkdir -p $OUT
cd $REPO
python -m pytest tests/test_gpu_parity.py tests/test_gpu_prover.py -m gpu -x -q -k "msm or k13 or toy" 2>&1 | tail -3 > $OUT/tests.log
for wl in "96 full" "240 small" "1 full"; do
  echo "== $wl" >> $OUT/msm_calls.txt
  BITS=13 python tools/exp/msm_table_bench.py 13 $wl 2>/dev/null >> $OUT/msm_calls.txt
done
ZKFHE_TRACE=1 ZKFHE_TRACE0=1 python bench.py --steps 1 --warmup 1 --streams 1 --no-cpu-baseline --steady-seconds 0 --transcript blake2b > $OUT/trace13.json 2> $OUT/trace13.txt
ZKFHE_TRACE=1 ZKFHE_TRACE0=1 python bench.py --config k19 --steps 1 --warmup 1 --streams 1 --steady-seconds 0 --transcript blake2b > $OUT/trace19.json 2> $OUT/trace19.txt
python bench.py --steps 8 --streams 1 --transcript blake2b --no-cpu-baseline --steady-seconds 0 > $OUT/bench_single_blake2b.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2>/dev/null
