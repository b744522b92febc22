#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r4dif4b
mkdir -p $OUT
cd $REPO
for v in 8 4 8 4 8 4; do
  ZKFHE_NTT_LDS_PASS=$v timeout 300 python bench.py --config k19 --steps 6 --warmup 1 --streams 1 --transcript blake2b --steady-seconds 0 --no-cpu-baseline > $OUT/k19_$v.json 2> $OUT/k19_$v.err
  echo "k19 per-thread=$v $(grep -o '"ms_per_step": [0-9.]*' $OUT/k19_$v.json)"
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_k19 -o r -- python $REPO/bench.py --config k19 --steps 2 --warmup 1 --streams 1 --transcript blake2b --steady-seconds 0 --no-cpu-baseline > $OUT/k19_prof.json 2> $OUT/k19_prof.err
python $REPO/tools/last_proof_timeline.py /tmp/prof_k19/r_results.db 200 2>&1 | grep -E "k_dif|span" > $OUT/k19_timeline.txt
cat $OUT/k19_timeline.txt
