#!/bin/bash
# round 5: quarter-column 2^13 kernel, second attempt (coset tables sized right; start skew of the first resident workgroups)
set -u
ulimit -c 0   # a faulting kernel must not fill the box's disk with a core file
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r5f
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ntt or coset or 2_13" > $OUT/ntt_tests.txt 2>&1; echo "rc=$?" >> $OUT/ntt_tests.txt
if ! grep -q "rc=0" $OUT/ntt_tests.txt; then exit 1; fi
for v in half quarter; do
  for sk in 0 2 4 6; do
    [ $v = half ] && [ $sk != 0 ] && continue
    ZKFHE_NTT13=$v ZKFHE_NTT13_SKEW=$sk python tools/exp/ntt13_bench.py > $OUT/ntt13_bench_${v}_$sk.txt 2>&1
  done
done
timeout 600 python -m pytest tests/test_gpu_prover.py -m gpu -x -q -k "bfv_in_k13 or twelve or toy" > $OUT/prover_tests.txt 2>&1; echo "rc=$?" >> $OUT/prover_tests.txt
for rep in 1 2; do
  for v in half quarter; do
    ZKFHE_NTT13=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline >> $OUT/driver_$v.json 2>> $OUT/driver_$v.err
  done
  ZKFHE_NTT13=quarter ZKFHE_NTT13_SKEW=4 python bench.py --steps 20 --warmup 5 --no-cpu-baseline >> $OUT/driver_quarter_s4.json 2>> $OUT/driver_quarter_s4.err
  ZKFHE_QUOTIENT=groups ZKFHE_NTT13=half python bench.py --steps 20 --warmup 5 --no-cpu-baseline >> $OUT/driver_half_groups.json 2>> $OUT/driver_half_groups.err
done
for v in half quarter; do
  ZKFHE_NTT13=$v python bench.py --no-cpu-baseline > $OUT/b96_$v.json 2>> $OUT/b96_$v.err
done
ZKFHE_QUOTIENT=groups ZKFHE_NTT13=half python bench.py --no-cpu-baseline > $OUT/b96_half_groups.json 2>> $OUT/b96_half_groups.err
ZKFHE_NTT13=half python bench.py --config k19 --steps 4 --warmup 1 --streams 1 --transcript blake2b --steady-seconds 0 --no-cpu-baseline > $OUT/k19_blocks.json 2> $OUT/k19_blocks.err
ZKFHE_QUOTIENT=groups ZKFHE_NTT13=half python bench.py --config k19 --steps 4 --warmup 1 --streams 1 --transcript blake2b --steady-seconds 0 --no-cpu-baseline > $OUT/k19_groups.json 2> $OUT/k19_groups.err
ZKFHE_NTT13=half python bench.py --config k16 --steps 6 --warmup 1 --streams 1 --transcript blake2b --steady-seconds 0 --no-cpu-baseline > $OUT/k16_blocks.json 2> $OUT/k16_blocks.err
ZKFHE_QUOTIENT=groups ZKFHE_NTT13=half python bench.py --config k16 --steps 6 --warmup 1 --streams 1 --transcript blake2b --steady-seconds 0 --no-cpu-baseline > $OUT/k16_groups.json 2> $OUT/k16_groups.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_k19
ZKFHE_NTT13=half rocprofv3 --kernel-trace --stats -d /tmp/prof_k19 -o r -- python $REPO/bench.py --config k19 --steps 2 --warmup 1 --streams 1 --transcript blake2b --steady-seconds 0 --no-cpu-baseline > $OUT/k19_prof.json 2> $OUT/k19_prof.err
python $REPO/tools/rocpd_stats.py /tmp/prof_k19/r_results.db > $OUT/k19_kernel_stats.md
python $REPO/tools/last_proof_stats.py /tmp/prof_k19/r_results.db > $OUT/k19_last_proof.txt
