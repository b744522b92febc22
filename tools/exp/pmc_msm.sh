# PMC comparison of the two summing kernels on the same 96 full-width columns (2^13 points):
# k_msm_accumulate (BITS=0: bucket pipeline, 10 MB table of window multiples) vs k_msm_table (BITS=13: 43 GB digit table)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for b in 0 13; do
  for set in "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_WAVES" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
    tag=$(echo $set | cut -d' ' -f1)
    rm -rf /tmp/pmc_m_${b}_$tag
    BITS=$b rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_m_${b}_$tag -o r -- python $R/tools/exp/msm_table_bench.py 13 96 full > /dev/null 2>&1
    db=$(find /tmp/pmc_m_${b}_$tag -name "*.db" | head -1)
    echo "== BITS=$b $tag"
    python $R/tools/pmc_per_launch.py $db k_msm_accumulate | grep -v "launches=0"
    python $R/tools/pmc_per_launch.py $db "k_msm_table<" | grep -v "launches=0"
  done
done
