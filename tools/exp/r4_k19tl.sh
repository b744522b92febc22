#!/bin/bash
# host trace + kernel timeline (idle gaps) of a lone k = 16 / k = 19 proof
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r4k19tl
mkdir -p $OUT
cd $REPO
for cfg in k16 k19; do
  ZKFHE_TRACE=1 ZKFHE_TRACE0=1 python bench.py --config $cfg --steps 3 --warmup 1 --streams 1 --transcript blake2b --steady-seconds 0 --no-cpu-baseline > $OUT/${cfg}.json 2> $OUT/${cfg}.trace
done
cd /tmp && export TMPDIR=/tmp
for cfg in k16 k19; do
  rocprofv3 --kernel-trace -d /tmp/prof_$cfg -o r -- python $REPO/bench.py --config $cfg --steps 2 --warmup 1 --streams 1 --transcript blake2b --steady-seconds 0 --no-cpu-baseline > $OUT/${cfg}_prof.json 2> $OUT/${cfg}_prof.err
  python $REPO/tools/last_proof_timeline.py /tmp/prof_$cfg/r_results.db 200 > $OUT/${cfg}_timeline.txt 2>&1
done
