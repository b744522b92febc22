#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r3q4
mkdir -p $OUT
cd $REPO
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ntt or coset or poly_mul or witness" 2>&1 | tail -3 > $OUT/tests.log
python -m pytest tests/test_gpu_prover.py -m gpu -x -q -k "k14 or k16 or k19 or toy" 2>&1 | tail -3 >> $OUT/tests.log
python bench.py --config k16 --steps 4 --streams 1 --transcript blake2b --steady-seconds 0 > $OUT/bench_k16_blake2b.json 2>/dev/null
python bench.py --config k16 --steps 8 --warmup 2 --transcript blake2b --steady-seconds 0 > $OUT/bench_k16_2streams.json 2>/dev/null
python bench.py --config k19 --steps 3 --streams 1 --transcript blake2b --steady-seconds 0 > $OUT/bench_k19_blake2b.json 2>/dev/null
