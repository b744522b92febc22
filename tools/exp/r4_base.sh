#!/bin/bash
# round-4 baseline on the day's box: driver command twice, host CPU per proof, single-proof host traces
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r4base
mkdir -p $OUT
cd $REPO
nproc > $OUT/host.txt; cat /sys/fs/cgroup/cpu.max >> $OUT/host.txt 2>&1; lscpu | grep -E "Model name|MHz" >> $OUT/host.txt
python bench.py --steps 20 --warmup 5 > $OUT/bench1.json 2>$OUT/bench1.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench2.json 2>/dev/null
python tools/exp/cpu_per_proof.py > $OUT/cpu_per_proof.txt 2>&1
ZKFHE_TRACE=1 python bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --steady-seconds 0 > $OUT/single_poseidon.json 2> $OUT/single_poseidon.trace
ZKFHE_TRACE=1 python bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --steady-seconds 0 --transcript blake2b > $OUT/single_blake2b.json 2> $OUT/single_blake2b.trace
g++ -O2 -std=c++17 -march=native -I zk-fhe_amd/host tools/exp/poseidon_ifma_check.cpp zk-fhe_amd/host/poseidon_ifma.cpp -o /tmp/pic 2>$OUT/pic.err && /tmp/pic > $OUT/poseidon_ifma_check.txt 2>&1
