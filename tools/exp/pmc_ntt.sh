cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python tools/exp/ntt13_bench.py > gpurun_out/r3_ntt13_b.log 2>&1
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ntt or coset or ext" 2>&1 | tail -5 >> gpurun_out/r3_ntt13_b.log
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU -d gpurun_out/pmc_ntt13 -o pmc1 --output-format csv -- python tools/exp/ntt13_bench.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE -d gpurun_out/pmc_ntt13 -o pmc2 --output-format csv -- python tools/exp/ntt13_bench.py > /dev/null 2>&1
ls gpurun_out/pmc_ntt13 >> gpurun_out/r3_ntt13_b.log
