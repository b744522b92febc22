cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $GRAFT_REPO_ROOT/gpurun_out/pmc_ic -o ic --output-format csv -- python $GRAFT_REPO_ROOT/tools/exp/ntt13_bench.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $GRAFT_REPO_ROOT/gpurun_out/pmc_ic -o ic2 --output-format csv -- python $GRAFT_REPO_ROOT/tools/exp/ntt13_bench.py > /dev/null 2>&1
rocprofv3 --list-avail 2>/dev/null | grep -i "icache\|ifetch\|inst_fetch\|SQ_WAIT\|SQ_IFETCH" | head -30 > $GRAFT_REPO_ROOT/gpurun_out/pmc_ic/avail.txt
