#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r4sort4
mkdir -p $OUT
cd $REPO
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "msm" > $OUT/tests.log 2>&1
tail -3 $OUT/tests.log
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_n -o r -- python $REPO/tools/exp/sort_probe.py 19 32 > $OUT/p_new.log 2>&1
python $REPO/tools/rocpd_stats.py /tmp/prof_n/r_results.db 2>&1 | grep -E "chist|cscatter|k_msm_fine|accumulate|k_msm_hist|k_msm_scatter" > $OUT/p_tl_new.txt
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_n16 -o r -- python $REPO/tools/exp/sort_probe.py 16 64 > $OUT/p_new16.log 2>&1
python $REPO/tools/rocpd_stats.py /tmp/prof_n16/r_results.db 2>&1 | grep -E "chist|cscatter|k_msm_fine|accumulate|k_msm_hist|k_msm_scatter" > $OUT/p_tl_new16.txt
ZKFHE_SORT=1 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_o16 -o r -- python $REPO/tools/exp/sort_probe.py 16 64 > $OUT/p_old16.log 2>&1
python $REPO/tools/rocpd_stats.py /tmp/prof_o16/r_results.db 2>&1 | grep -E "chist|cscatter|k_msm_fine|accumulate|k_msm_hist|k_msm_scatter" > $OUT/p_tl_old16.txt
cd $REPO
for cfg in k16 k19; do
  timeout 300 python bench.py --config $cfg --steps 4 --warmup 1 --streams 1 --transcript blake2b --steady-seconds 0 --no-cpu-baseline > $OUT/${cfg}.json 2> $OUT/${cfg}.err
  grep -o '"ms_per_step": [0-9.]*' $OUT/${cfg}.json
done
