#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r4sort8
mkdir -p $OUT
cd $REPO
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "msm" > $OUT/tests.log 2>&1
tail -3 $OUT/tests.log
cd /tmp && export TMPDIR=/tmp
for cfg in k16 k19; do
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$cfg -o r -- python $REPO/bench.py --config $cfg --steps 2 --warmup 1 --streams 1 --transcript blake2b --steady-seconds 0 --no-cpu-baseline > $OUT/${cfg}_prof.json 2> $OUT/${cfg}_prof.err
  python $REPO/tools/last_proof_timeline.py /tmp/prof_$cfg/r_results.db 200 > $OUT/${cfg}_timeline.txt 2>&1
done
cd $REPO
for w in 14; do
  ZKFHE_WINDOW_BITS=$w timeout 300 python bench.py --config k16 --steps 4 --warmup 1 --streams 1 --transcript blake2b --steady-seconds 0 --no-cpu-baseline > $OUT/k16_w$w.json 2> $OUT/k16_w$w.err
  echo "k16 w=$w $(grep -o '"ms_per_step": [0-9.]*' $OUT/k16_w$w.json)"
done
timeout 300 python bench.py --config k19 --steps 4 --warmup 1 --streams 1 --transcript blake2b --steady-seconds 0 --no-cpu-baseline > $OUT/k19.json 2> $OUT/k19.err
echo "k19 $(grep -o '"ms_per_step": [0-9.]*' $OUT/k19.json)"
