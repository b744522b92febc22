#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r4lk
mkdir -p $OUT
cd $REPO
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lookup" > $OUT/tests.log 2>&1
tail -2 $OUT/tests.log
for cfg in k16 k19; do
  timeout 300 python bench.py --config $cfg --steps 4 --warmup 1 --streams 1 --transcript blake2b --steady-seconds 0 --no-cpu-baseline > $OUT/${cfg}.json 2> $OUT/${cfg}.err
  echo "$cfg $(grep -o '"ms_per_step": [0-9.]*' $OUT/${cfg}.json)"
done
cd /tmp && export TMPDIR=/tmp
for cfg in k16 k19; do
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$cfg -o r -- python $REPO/bench.py --config $cfg --steps 2 --warmup 1 --streams 1 --transcript blake2b --steady-seconds 0 --no-cpu-baseline > $OUT/${cfg}_prof.json 2> $OUT/${cfg}_prof.err
  python $REPO/tools/last_proof_timeline.py /tmp/prof_$cfg/r_results.db 150 > $OUT/${cfg}_timeline.txt 2>&1
done
