#!/bin/bash
# two-level sort: parity tests, then k16 / k19 proofs with the one-pass sort (ZKFHE_SORT=1) and the two-level sort
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r4sort2
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "msm" > $OUT/tests.log 2>&1
tail -5 $OUT/tests.log
for cfg in k16 k19; do
  for s in 1 0; do
    ZKFHE_SORT=$s python bench.py --config $cfg --steps 4 --warmup 1 --streams 1 --transcript blake2b --steady-seconds 0 --no-cpu-baseline > $OUT/${cfg}_sort$s.json 2> $OUT/${cfg}_sort$s.err
    grep -o '"ms_per_step": [0-9.]*' $OUT/${cfg}_sort$s.json
  done
done
cd /tmp && export TMPDIR=/tmp
for cfg in k16 k19; do
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$cfg -o r -- python $REPO/bench.py --config $cfg --steps 2 --warmup 1 --streams 1 --transcript blake2b --steady-seconds 0 --no-cpu-baseline > $OUT/${cfg}_prof.json 2> $OUT/${cfg}_prof.err
  python $REPO/tools/last_proof_timeline.py /tmp/prof_$cfg/r_results.db 200 > $OUT/${cfg}_timeline.txt 2>&1
done
