#!/bin/bash
# round 5: the wide table MSM as two kernels (lists, then additions) against the fused kernel (ZKFHE_MSM_FUSED=1)
set -u
ulimit -c 0
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r5l
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "msm" > $OUT/msm_tests.txt 2>&1; echo "rc=$?" >> $OUT/msm_tests.txt
if ! grep -q "rc=0" $OUT/msm_tests.txt; then exit 1; fi
timeout 600 python -m pytest tests/test_gpu_prover.py -m gpu -x -q -k "bfv_in_k13 or twelve or toy" > $OUT/prover_tests.txt 2>&1; echo "rc=$?" >> $OUT/prover_tests.txt
export ZKFHE_TABLE_GB=160
for f in 0 1; do
  for wl in "96 full" "240 small" "240 mixed" "136 full"; do
    echo "== fused=$f $wl" >> $OUT/msm_calls.txt
    ZKFHE_MSM_FUSED=$f BITS=15 python tools/exp/msm_table_bench.py 13 $wl 2>/dev/null >> $OUT/msm_calls.txt
  done
done
for rep in 1 2; do
  for f in 0 1; do
    ZKFHE_MSM_FUSED=$f python bench.py --steps 20 --warmup 5 --no-cpu-baseline >> $OUT/driver_fused$f.json 2>> $OUT/driver_fused$f.err
    ZKFHE_MSM_FUSED=$f python bench.py --no-cpu-baseline >> $OUT/b96_fused$f.json 2>> $OUT/b96_fused$f.err
  done
done
for f in 0 1; do
  ZKFHE_MSM_FUSED=$f python bench.py --steps 8 --streams 1 --transcript blake2b --no-cpu-baseline --steady-seconds 0 > $OUT/single_fused$f.json 2>/dev/null
done
