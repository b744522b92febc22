#!/bin/bash
# quick r3 measurements: MSM calls, one-proof timeline, sharded tests
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r3q
mkdir -p $OUT
cd $REPO
python -m pytest tests/test_batch_gloo.py -m gpu -x -q 2>&1 | tail -5 > $OUT/sharded_tests.log
for wl in "96 full" "240 small" "240 mixed" "1 full"; do
  echo "== $wl" >> $OUT/msm_calls.txt
  BITS=13 python tools/exp/msm_table_bench.py 13 $wl 2>/dev/null >> $OUT/msm_calls.txt
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_b
rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o r -- python $REPO/bench.py --steps 4 --warmup 1 --streams 1 --no-cpu-baseline --transcript blake2b --steady-seconds 0 > $OUT/b_single_bench.json 2> $OUT/b_err.log
python $REPO/tools/last_proof_stats.py /tmp/prof_b/r_results.db > $OUT/b_single_last_proof.txt
python $REPO/tools/last_proof_timeline.py /tmp/prof_b/r_results.db 10 > $OUT/b_single_timeline.txt
cd $REPO
ZKFHE_TRACE=1 python bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --steady-seconds 0 > $OUT/trace_bench.json 2> $OUT/trace_poseidon.txt
