#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r3big
mkdir -p $OUT
cd $REPO
for k in k16 k19; do
  timeout 900 python bench.py --config $k --steps 4 --streams 1 --transcript blake2b --steady-seconds 0 > $OUT/bench_${k}_blake2b.json 2>/dev/null
  timeout 900 python bench.py --config $k --steps 4 --streams 1 --steady-seconds 0 > $OUT/bench_${k}_poseidon.json 2>/dev/null
done
timeout 900 python bench.py --config k16 --steps 8 --warmup 2 --transcript blake2b --steady-seconds 0 > $OUT/bench_k16_2streams.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
for k in k16 k19; do
  rm -rf /tmp/prof_$k
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$k -o r -- python $REPO/bench.py --config $k --steps 2 --warmup 1 --streams 1 --transcript blake2b --steady-seconds 0 > $OUT/prof_${k}_bench.json 2> $OUT/prof_${k}_err.log
  python $REPO/tools/last_proof_stats.py /tmp/prof_$k/r_results.db > $OUT/${k}_last_proof.txt
done
