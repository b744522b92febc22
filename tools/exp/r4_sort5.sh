#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r4sort5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in k16 k19; do
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$cfg -o r -- python $REPO/bench.py --config $cfg --steps 2 --warmup 1 --streams 1 --transcript blake2b --steady-seconds 0 --no-cpu-baseline > $OUT/${cfg}_prof.json 2> $OUT/${cfg}_prof.err
  python $REPO/tools/last_proof_timeline.py /tmp/prof_$cfg/r_results.db 200 > $OUT/${cfg}_timeline.txt 2>&1
done
