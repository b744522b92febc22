#!/bin/bash
# round 6, GPU pass 18: soak of the shipped tree (products as inline assembly, upload without a copy command): every proof verified, repeated
# (input, seed) pairs compared byte for byte
set -u
OUT=gpurun_out/r6s; mkdir -p $OUT
timeout 600 python tools/soak.py --proofs 6000 --streams 16 --transcript poseidon --gate 4 > $OUT/soak_poseidon.txt 2>&1; tail -3 $OUT/soak_poseidon.txt
timeout 400 python tools/soak.py --proofs 4000 --streams 16 --transcript blake2b > $OUT/soak_blake2b.txt 2>&1; tail -3 $OUT/soak_blake2b.txt
timeout 400 python tools/soak.py --proofs 2000 --streams 20 --transcript poseidon --hash-mode shared --announce 0 > $OUT/soak_shared.txt 2>&1; tail -3 $OUT/soak_shared.txt
timeout 400 python tools/soak.py --proofs 200 --streams 3 --config k16 --transcript poseidon --announce 3 > $OUT/soak_k16.txt 2>&1; tail -3 $OUT/soak_k16.txt
