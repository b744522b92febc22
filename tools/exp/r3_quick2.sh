#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r3q2
mkdir -p $OUT
cd $REPO
python -m pytest tests/test_batch_gloo.py -m gpu -x -q > $OUT/sharded_tests.log 2>&1
for wl in "96 full" "240 small" "240 mixed" "1 full"; do
  echo "== $wl" >> $OUT/msm_calls.txt
  BITS=13 python tools/exp/msm_table_bench.py 13 $wl 2>/dev/null >> $OUT/msm_calls.txt
done
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "msm" 2>&1 | tail -3 > $OUT/msm_tests.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2>/dev/null
