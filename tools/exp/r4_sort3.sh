#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r4sort3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for f in 0 1 2; do
  ZKFHE_EXPF=$f timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_$f -o r -- python $REPO/tools/exp/sort_probe.py 19 32 > $OUT/p_$f.log 2>&1
  python $REPO/tools/rocpd_stats.py /tmp/prof_$f/r_results.db 2>&1 | grep -E "chist|cscatter|k_msm_fine|accumulate|k_msm_hist|k_msm_scatter" > $OUT/p_tl_$f.txt
done
ZKFHE_SORT=1 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_old -o r -- python $REPO/tools/exp/sort_probe.py 19 32 > $OUT/p_old.log 2>&1
python $REPO/tools/rocpd_stats.py /tmp/prof_old/r_results.db 2>&1 | grep -E "chist|cscatter|k_msm_fine|accumulate|k_msm_hist|k_msm_scatter" > $OUT/p_tl_old.txt
