#!/bin/bash
# Round-2 profiles (run on the MI355X box through gpurun; results are copied into gpurun_out/r2/ and from there into profiles/).
#   bash tools/profile_r2.sh
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
prof() {  # name, then the bench arguments
  local name=$1; shift
  rm -rf /tmp/prof_$name
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o r -- python $REPO/bench.py "$@" > $OUT/${name}_bench.json 2> $OUT/${name}_err.log
  python $REPO/tools/rocpd_stats.py /tmp/prof_$name/r_results.db > $OUT/${name}_kernel_stats.md
}
# (a) the default bench command (what the driver runs), 16 proofs in flight, Poseidon transcript
prof a_default --no-cpu-baseline
# (b) one proof in flight, Blake2b transcript: per-kernel costs, the timeline of the last proof, MSM kernels per call
prof b_single --steps 4 --warmup 1 --streams 1 --no-cpu-baseline --transcript blake2b --steady-seconds 0
python $REPO/tools/last_proof_stats.py /tmp/prof_b_single/r_results.db > $OUT/b_single_last_proof.txt
python $REPO/tools/last_proof_timeline.py /tmp/prof_b_single/r_results.db 10 > $OUT/b_single_timeline.txt
# (c) k = 16 and k = 19, one proof in flight
prof c_k16 --config k16 --steps 3 --warmup 1 --streams 1 --transcript blake2b --steady-seconds 0
python $REPO/tools/last_proof_stats.py /tmp/prof_c_k16/r_results.db > $OUT/c_k16_last_proof.txt
prof d_k19 --config k19 --steps 2 --warmup 1 --streams 1 --transcript blake2b --steady-seconds 0
python $REPO/tools/last_proof_stats.py /tmp/prof_d_k19/r_results.db > $OUT/d_k19_last_proof.txt
# (d) HBM traffic: one counter per pass (FETCH_SIZE and WRITE_SIZE do not fit together), kernel trace only
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$ctr
  rocprofv3 --pmc $ctr --kernel-trace -d /tmp/pmc_$ctr -o r -- python $REPO/bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --steady-seconds 0 > /dev/null 2> $OUT/pmc_${ctr}_err.log
  python $REPO/tools/pmc_stats.py /tmp/pmc_$ctr/r_results.db > $OUT/pmc_$ctr.txt
done
for k in k16 k19; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${k}_$ctr
    rocprofv3 --pmc $ctr --kernel-trace -d /tmp/pmc_${k}_$ctr -o r -- python $REPO/bench.py --config $k --steps 1 --warmup 1 --streams 1 --transcript blake2b --steady-seconds 0 > /dev/null 2> $OUT/pmc_${k}_${ctr}_err.log
    python $REPO/tools/pmc_stats.py /tmp/pmc_${k}_$ctr/r_results.db > $OUT/pmc_${k}_$ctr.txt
  done
done
# (e) micro-benchmarks and the un-profiled bench lines
cd $REPO
python tools/microbench.py > $OUT/microbench.json 2> $OUT/microbench_err.log
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default_err.log
python bench.py --steps 20 --no-cpu-baseline > $OUT/bench_steps20.json 2>/dev/null
python bench.py --transcript blake2b --no-cpu-baseline > $OUT/bench_blake2b.json 2>/dev/null
python bench.py --steps 8 --streams 1 --transcript blake2b --no-cpu-baseline --steady-seconds 0 > $OUT/bench_single_blake2b.json 2>/dev/null
python bench.py --steps 8 --streams 1 --no-cpu-baseline --steady-seconds 0 > $OUT/bench_single_poseidon.json 2>/dev/null
for k in k16 k19; do
  python bench.py --config $k --steps 4 --streams 1 --transcript blake2b --steady-seconds 0 > $OUT/bench_${k}_blake2b.json 2>/dev/null
  python bench.py --config $k --steps 4 --streams 1 --steady-seconds 0 > $OUT/bench_${k}_poseidon.json 2>/dev/null
done
ls -la $OUT
