#!/bin/bash
# GPU occupancy over the driver's wave of 20 proofs (2 ms bins), x8 hashing off and on
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r4wave
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in off on; do
  rm -rf /tmp/prof_w_$mode
  if [ $mode = off ]; then export ZKFHE_POSEIDON_X8=0; else unset ZKFHE_POSEIDON_X8; fi
  rocprofv3 --kernel-trace -d /tmp/prof_w_$mode -o r -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --steady-seconds 0 > $OUT/${mode}_bench.json 2> $OUT/${mode}_err.log
  python $REPO/tools/busy_bins.py /tmp/prof_w_$mode/r_results.db 400 2 > $OUT/${mode}_bins.txt
done
