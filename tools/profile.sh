#!/bin/bash
# The round's profiles: kernel traces, PMC passes, micro-benchmarks and un-profiled bench lines of every configuration.
# Run on the MI355X box through gpurun; results land in gpurun_out/r<round>/ and are summarised into profiles/r<round>_*.md by
# tools/make_profiles.py --round <round>.      bash tools/profile.sh <round> [sections, default "a b c d e f g"]
# (One script for every round: rounds 2-5 each kept a copy of this file that differed by the directory name.)
set -u
ulimit -c 0   # a faulting kernel must not fill the box's disk with a core file
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
RT=r${1:?usage: profile.sh <round> [sections]}
SECTIONS=${2:-a b c d e f g}
has() { [[ " $SECTIONS " == *" $1 "* ]]; }
OUT=$REPO/gpurun_out/$RT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
prof() {  # name, then the bench arguments
  local name=$1; shift
  rm -rf /tmp/prof_$name
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o r -- python $REPO/bench.py --no-traffic-pass "$@" > $OUT/${name}_bench.json 2> $OUT/${name}_err.log
  python $REPO/tools/rocpd_stats.py /tmp/prof_$name/r_results.db > $OUT/${name}_kernel_stats.md
}
# (a) the driver's command: Poseidon transcript, 20 steps, 5 warm-up -> one wave of 20 concurrent proofs
if has a; then
  prof a_driver --steps 20 --warmup 5 --no-cpu-baseline
  python $REPO/tools/busy_bins.py /tmp/prof_a_driver/r_results.db 400 2 > $OUT/a_driver_bins.txt
fi
# (b) one proof in flight, Blake2b transcript: per-kernel costs, the timeline of the last proof
if has b; then
  prof b_single --steps 4 --warmup 1 --streams 1 --no-cpu-baseline --transcript blake2b --steady-seconds 0
  python $REPO/tools/last_proof_stats.py /tmp/prof_b_single/r_results.db > $OUT/b_single_last_proof.txt
  python $REPO/tools/last_proof_timeline.py /tmp/prof_b_single/r_results.db 10 > $OUT/b_single_timeline.txt
fi
# (c) k = 16 and k = 19, one proof in flight
if has c; then
  prof c_k16 --config k16 --steps 3 --warmup 1 --streams 1 --transcript blake2b --steady-seconds 0
  python $REPO/tools/last_proof_stats.py /tmp/prof_c_k16/r_results.db > $OUT/c_k16_last_proof.txt
  prof d_k19 --config k19 --steps 2 --warmup 1 --streams 1 --transcript blake2b --steady-seconds 0
  python $REPO/tools/last_proof_stats.py /tmp/prof_d_k19/r_results.db > $OUT/d_k19_last_proof.txt
fi
# (d) counters, one or two per pass, kernel trace only: HBM traffic and VALU instructions of a proof
if has d; then
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$ctr
    rocprofv3 --pmc $ctr --kernel-trace -d /tmp/pmc_$ctr -o r -- python $REPO/bench.py --no-traffic-pass --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --steady-seconds 0 > /dev/null 2> $OUT/pmc_${ctr}_err.log
    python $REPO/tools/pmc_stats.py /tmp/pmc_$ctr/r_results.db > $OUT/pmc_$ctr.txt
  done
  for ctr in FETCH_SIZE WRITE_SIZE; do
    for cfg in k16 k19; do
      rm -rf /tmp/pmc_${cfg}_$ctr
      rocprofv3 --pmc $ctr --kernel-trace -d /tmp/pmc_${cfg}_$ctr -o r -- python $REPO/bench.py --no-traffic-pass --config $cfg --steps 1 --warmup 1 --streams 1 --transcript blake2b --no-cpu-baseline --steady-seconds 0 > /dev/null 2> $OUT/pmc_${cfg}_${ctr}_err.log
      python $REPO/tools/pmc_stats.py /tmp/pmc_${cfg}_$ctr/r_results.db > $OUT/pmc_${cfg}_$ctr.txt
    done
  done
  rm -rf /tmp/pmc_valu
  rocprofv3 --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES --kernel-trace -d /tmp/pmc_valu -o r -- python $REPO/bench.py --no-traffic-pass --steps 6 --warmup 1 --streams 1 --no-cpu-baseline --transcript blake2b --steady-seconds 0 > /dev/null 2> $OUT/pmc_valu_err.log
  python $REPO/tools/pmc_per_proof.py /tmp/pmc_valu/r_results.db 10 > $OUT/pmc_valu_per_proof.md
  # (d2) the 2^13 tile alone (tools/exp/ntt13_bench.py: 256 columns out of place, inverse, 4 coset rows): time, traffic, LDS
  cd $REPO
  python tools/exp/ntt13_bench.py > $OUT/ntt13_bench.txt 2>&1
  cd /tmp
  for ctr in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVE_CYCLES"; do
    tag=$(echo $ctr | cut -d' ' -f1)
    rm -rf /tmp/pmc_n_$tag
    rocprofv3 --pmc $ctr --kernel-trace -d /tmp/pmc_n_$tag -o r -- python $REPO/tools/exp/ntt13_bench.py > /dev/null 2>&1
    python $REPO/tools/pmc_per_launch.py /tmp/pmc_n_$tag/r_results.db k_ntt13 >> $OUT/ntt13_pmc.txt
  done
  rm -rf /tmp/prof_n
  rocprofv3 --kernel-trace --stats -d /tmp/prof_n -o r -- python $REPO/tools/exp/ntt13_bench.py > /dev/null 2>&1
  python $REPO/tools/kernel_resources.py /tmp/prof_n/r_results.db k_ntt13 > $OUT/ntt13_resources.txt 2>&1
fi
# (e) one MSM call through the table path
if has e; then
  cd $REPO
  for wl in "96 full" "240 small" "240 mixed" "1 full" "3 full"; do
    echo "== $wl" >> $OUT/msm_calls.txt
    BITS=0,13 python tools/exp/msm_table_bench.py 13 $wl 2>/dev/null >> $OUT/msm_calls.txt
  done
fi
# (f) micro-benchmarks and the un-profiled bench lines
if has f; then
  cd $REPO
  python tools/microbench.py --big > $OUT/microbench.json 2> $OUT/microbench_err.log
  python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver_err.log
  python bench.py --no-traffic-pass --no-cpu-baseline > $OUT/bench_default.json 2>/dev/null
  python bench.py --no-traffic-pass --transcript blake2b --no-cpu-baseline > $OUT/bench_blake2b.json 2>/dev/null
  python bench.py --no-traffic-pass --transcript blake2b --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_blake2b_20.json 2>/dev/null
  python bench.py --no-traffic-pass --steps 8 --streams 1 --transcript blake2b --no-cpu-baseline --steady-seconds 0 > $OUT/bench_single_blake2b.json 2>/dev/null
  python bench.py --no-traffic-pass --steps 8 --streams 1 --announce off --no-cpu-baseline --steady-seconds 0 > $OUT/bench_single_poseidon.json 2>/dev/null
  # configs[3] / [4]: Blake2b (no host hashing to speak of), then the reference's Poseidon transcript -- one proof alone WITHOUT announcing
  # its input (the first challenge waits for the sequential sponge over 5 N + 1 public inputs: host-bound latency), the same with every
  # input announced one proof ahead (zkfhe_bfv_pk_prehash: the sponge of proof i + 1 runs on a host thread while proof i is on the GPU),
  # and 2 / 3 proofs in flight
  for k in k16 k19; do
    python bench.py --no-traffic-pass --config $k --steps 4 --streams 1 --transcript blake2b --steady-seconds 0 > $OUT/bench_${k}_blake2b.json 2>/dev/null
    python bench.py --no-traffic-pass --config $k --steps 6 --streams 1 --announce off --steady-seconds 0 > $OUT/bench_${k}_poseidon.json 2>/dev/null
    python bench.py --no-traffic-pass --config $k --steps 9 --warmup 2 --streams 1 --steady-seconds 0 > $OUT/bench_${k}_poseidon_announced.json 2>/dev/null
    for st in 2 3; do
      python bench.py --no-traffic-pass --config $k --steps 9 --warmup 2 --streams $st --steady-seconds 0 > $OUT/bench_${k}_poseidon_s$st.json 2>/dev/null
    done
    python bench.py --no-traffic-pass --config $k --steps 9 --warmup 2 --streams 3 --announce off --steady-seconds 0 > $OUT/bench_${k}_poseidon_s3_plain.json 2>/dev/null
  done
  python bench.py --no-traffic-pass --config k16 --steps 8 --warmup 2 --transcript blake2b --steady-seconds 0 > $OUT/bench_k16_2streams.json 2>/dev/null
  python bench.py --no-traffic-pass --steps 16 --warmup 2 --streams 1 --no-cpu-baseline --steady-seconds 0 > $OUT/bench_single_poseidon_announced.json 2>/dev/null
  ls -la $OUT
fi
# (g) host hashing modes, admission gate, the transcript cache off, the quotient by kind
if has g; then
  cd $REPO
  ZKFHE_HASH_MODE=shared python bench.py --no-traffic-pass --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_shared.json 2>/dev/null
  ZKFHE_HASH_MODE=shared python bench.py --no-traffic-pass --no-cpu-baseline > $OUT/bench_default_shared.json 2>/dev/null
  ZKFHE_GATE=4 python bench.py --no-traffic-pass --no-cpu-baseline > $OUT/bench_default_gate4.json 2>/dev/null
  ZKFHE_PREFIX_CACHE=0 python bench.py --no-traffic-pass --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_nocache.json 2>/dev/null
  /opt/rocm/lib/llvm/bin/clang++ -O3 -std=c++17 -I zk-fhe_amd/host tools/exp/poseidon_x8_check.cpp zk-fhe_amd/host/poseidon_x8.cpp zk-fhe_amd/host/poseidon_ifma.cpp -o /tmp/px8 -lpthread 2>/dev/null && /tmp/px8 > $OUT/poseidon_x8.txt 2>&1
  nproc > $OUT/host.txt; cat /sys/fs/cgroup/cpu.max >> $OUT/host.txt 2>&1; lscpu | grep -E "Model name" >> $OUT/host.txt
  ls -la $OUT
fi
