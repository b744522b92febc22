#!/bin/bash
# Round-4 supplement to tools/profile_r4.sh: the wave's occupancy from a run WITHOUT the steady-state pass (so that the tail of the
# trace is the wave + the two profiled proofs), and the bench lines that changed when bench.py began to set the admission gate for
# runs with more steps than streams.  Results merge into gpurun_out/r4/.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r4
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_wave
rocprofv3 --kernel-trace -d /tmp/prof_wave -o r -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --steady-seconds 0 > $OUT/wave_bins_bench.json 2> $OUT/wave_bins_err.log
python $REPO/tools/busy_bins.py /tmp/prof_wave/r_results.db 260 2 > $OUT/a_driver_bins.txt
cd $REPO
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver_err.log
python bench.py --no-cpu-baseline > $OUT/bench_default.json 2>/dev/null
ZKFHE_GATE=0 python bench.py --no-cpu-baseline > $OUT/bench_default_gate0.json 2>/dev/null
python bench.py --transcript blake2b --no-cpu-baseline > $OUT/bench_blake2b.json 2>/dev/null
ZKFHE_HASH_MODE=shared python bench.py --no-cpu-baseline > $OUT/bench_default_shared.json 2>/dev/null
rm -f $OUT/bench_default_gate4.json
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -5 > $OUT/full_gpu_suite.log
