#!/bin/bash
# round 6, GPU pass 19: the 2^13 tile with its first round of workgroups started in four groups (ZKFHE_NTT13_SKEW = s_sleep(127) steps of 3.4 us
# between groups) -- is the tile's idle time the whole chip loading, computing and storing in step?
set -u
OUT=gpurun_out/r6u; mkdir -p $OUT
for sk in 0 1 2 3 4 6; do
  echo "== skew $sk" >> $OUT/ntt13.txt
  ZKFHE_NTT13_SKEW=$sk python tools/exp/ntt13_bench.py 2>/dev/null | grep -v amdgpu.ids >> $OUT/ntt13.txt
done
cat $OUT/ntt13.txt
