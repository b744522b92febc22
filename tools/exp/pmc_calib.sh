# Calibration of FETCH_SIZE / WRITE_SIZE on this library's access patterns (tools/exp/pmc_calib.hip): known bytes vs counter.
# Writes gpurun_out/pmc_calib/summary.txt; profiles/r5_pmc_calibration.md is made from it.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_calib
mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -std=c++17 $R/tools/exp/pmc_calib.hip -o /tmp/pmc_calib || exit 1
/tmp/pmc_calib > $O/known.txt 2>&1
for ctr in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  tag=$(echo $ctr | cut -d' ' -f1)
  rm -rf /tmp/pc_$tag
  rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pc_$tag -o r -- /tmp/pmc_calib > /tmp/pc_$tag.log 2>&1
  db=$(find /tmp/pc_$tag -name "*.db" | head -1)
  echo "== $ctr" >> $O/counters.txt
  for k in k_gather64_window "k_gather64(" "k_stream<1>" "k_stream<2>" "k_store<1>" "k_store<2>"; do
    python $R/tools/pmc_per_launch.py $db "$k" >> $O/counters.txt 2>&1
  done
done
cat $O/known.txt $O/counters.txt > $O/summary.txt
