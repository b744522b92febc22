#!/bin/bash
# round 5: the quarter-column 2^13 kernel against the half-column one (tests, micro-benchmark, the driver's command), and a k = 19
# kernel profile with the four-step passes
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r5e
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ntt or coset or 2_13" > $OUT/ntt_tests.txt 2>&1; echo "rc=$?" >> $OUT/ntt_tests.txt
for v in quarter half; do
  ZKFHE_NTT13=$v python tools/exp/ntt13_bench.py > $OUT/ntt13_bench_$v.txt 2>&1
done
timeout 600 python -m pytest tests/test_gpu_prover.py -m gpu -x -q -k "bfv_in_k13 or twelve or toy" > $OUT/prover_tests.txt 2>&1; echo "rc=$?" >> $OUT/prover_tests.txt
for v in quarter half quarter half; do
  ZKFHE_NTT13=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline >> $OUT/driver_$v.json 2>> $OUT/driver_$v.err
done
for v in quarter half; do
  ZKFHE_NTT13=$v python bench.py --no-cpu-baseline > $OUT/b96_$v.json 2>> $OUT/b96_$v.err
  ZKFHE_NTT13=$v python bench.py --config k19 --steps 4 --warmup 1 --streams 1 --transcript blake2b --steady-seconds 0 --no-cpu-baseline > $OUT/k19_$v.json 2> $OUT/k19_$v.err
  ZKFHE_NTT13=$v python bench.py --config k16 --steps 6 --warmup 1 --streams 1 --transcript blake2b --steady-seconds 0 --no-cpu-baseline > $OUT/k16_$v.json 2> $OUT/k16_$v.err
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_k19
rocprofv3 --kernel-trace --stats -d /tmp/prof_k19 -o r -- python $REPO/bench.py --config k19 --steps 2 --warmup 1 --streams 1 --transcript blake2b --steady-seconds 0 --no-cpu-baseline > $OUT/k19_prof.json 2> $OUT/k19_prof.err
python $REPO/tools/rocpd_stats.py /tmp/prof_k19/r_results.db > $OUT/k19_kernel_stats.md
python $REPO/tools/last_proof_stats.py /tmp/prof_k19/r_results.db > $OUT/k19_last_proof.txt
