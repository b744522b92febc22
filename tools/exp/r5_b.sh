#!/bin/bash
# round 5, second GPU call: counter calibration, table budget sweep, lone proofs, the 2^13 tile on the day's box
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r5b
mkdir -p $OUT
cd $REPO
nproc > $OUT/host.txt; cat /sys/fs/cgroup/cpu.max >> $OUT/host.txt 2>&1; lscpu | grep -E "Model name|MHz" >> $OUT/host.txt
bash tools/exp/pmc_calib.sh > $OUT/pmc_calib.log 2>&1
cp gpurun_out/pmc_calib/summary.txt $OUT/pmc_calib_summary.txt
cd $REPO
for gb in default 90 160; do
  if [ $gb = default ]; then
    # the library default (48 GB, a quarter of the free memory): bench.py would opt into 160, so name the default's value
    ZKFHE_TABLE_GB=48 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/table_${gb}_driver.json 2> $OUT/table_${gb}.err
    ZKFHE_TABLE_GB=48 python bench.py --no-cpu-baseline > $OUT/table_${gb}_96.json 2>> $OUT/table_${gb}.err
  else
    ZKFHE_TABLE_GB=$gb python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/table_${gb}_driver.json 2> $OUT/table_${gb}.err
    ZKFHE_TABLE_GB=$gb python bench.py --no-cpu-baseline > $OUT/table_${gb}_96.json 2>> $OUT/table_${gb}.err
  fi
done
ZKFHE_TRACE=1 python bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --steady-seconds 0 > $OUT/single_poseidon.json 2> $OUT/single_poseidon.trace
ZKFHE_TRACE=1 python bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --steady-seconds 0 --transcript blake2b > $OUT/single_blake2b.json 2> $OUT/single_blake2b.trace
ZKFHE_PREFIX_CACHE=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/nocache_driver.json 2> $OUT/nocache.err
python tools/exp/ntt13_bench.py > $OUT/ntt13_bench.txt 2>&1
python bench.py --config k16 --steps 3 --warmup 1 --streams 1 --transcript blake2b --steady-seconds 0 --no-cpu-baseline > $OUT/k16.json 2> $OUT/k16.err
python bench.py --config k19 --steps 2 --warmup 1 --streams 1 --transcript blake2b --steady-seconds 0 --no-cpu-baseline > $OUT/k19.json 2> $OUT/k19.err
