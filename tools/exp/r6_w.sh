#!/bin/bash
# round 6, GPU pass 21: where the waves of each kernel spend their cycles (one SQ counter pass, kernel trace only), one proof in flight and sixteen
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/r6w; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CTRS="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS"
rm -rf /tmp/pmc_wc1 /tmp/pmc_wc16
rocprofv3 --pmc $CTRS --kernel-trace -d /tmp/pmc_wc1 -o r -- python $REPO/bench.py --no-traffic-pass --steps 6 --warmup 1 --streams 1 --no-cpu-baseline --transcript blake2b --steady-seconds 0 > /dev/null 2> $OUT/wc1_err.log
python $REPO/tools/pmc_wave_cycles.py /tmp/pmc_wc1/r_results.db k_basis,k_g1_mul > $OUT/wave_cycles_1.md
rocprofv3 --pmc $CTRS --kernel-trace -d /tmp/pmc_wc16 -o r -- python $REPO/bench.py --no-traffic-pass --steps 48 --warmup 4 --no-cpu-baseline --steady-seconds 0 > /dev/null 2> $OUT/wc16_err.log
python $REPO/tools/pmc_wave_cycles.py /tmp/pmc_wc16/r_results.db k_basis,k_g1_mul > $OUT/wave_cycles_16.md
cat $OUT/wave_cycles_1.md; cat $OUT/wave_cycles_16.md; tail -2 $OUT/wc1_err.log
