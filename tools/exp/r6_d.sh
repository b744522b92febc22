#!/bin/bash
# round 6, GPU passes 4 and 8: the nine-limb products with their multiply-adds tied to one accumulator (libzkfhe_hip_tied.so = the tree built
# with -DZK_MAD_TIED: pass 4 one asm statement per multiply-add, pass 8 one per column) against the C form (the default build): parity first, then micro-benchmarks and bench lines,
# alternating the two libraries on one box
set -u
OUT=gpurun_out/r6i; mkdir -p $OUT
L=zk-fhe_amd/libzkfhe_hip.so
cp zk-fhe_amd/libzkfhe_hip_tied.so /tmp/lib_asm.so; cp $L /tmp/lib_c.so
cp /tmp/lib_asm.so $L   # parity runs on the TIED build
python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $OUT/pytest_parity.log 2>&1; tail -3 $OUT/pytest_parity.log
python -m pytest tests/test_gpu_prover.py -m gpu -x -q -k "toy or bfv_in_k13 or k14 or config4" > $OUT/pytest_prover.log 2>&1; tail -3 $OUT/pytest_prover.log
for rep in 1 2; do
  for v in asm c; do
    cp /tmp/lib_$v.so $L
    python tools/exp/ntt13_bench.py > $OUT/ntt13_${v}_$rep.txt 2>&1
    BITS=13 python tools/exp/msm_table_bench.py 13 96 full > $OUT/msm96_${v}_$rep.txt 2>&1
    BITS=13 python tools/exp/msm_table_bench.py 13 240 small > $OUT/msm240_${v}_$rep.txt 2>&1
    python bench.py --steps 96 --warmup 4 --no-cpu-baseline --no-traffic-pass > $OUT/bench96_${v}_$rep.json 2>/dev/null
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic-pass --steady-seconds 0 > $OUT/bench20_${v}_$rep.json 2>/dev/null
    python bench.py --steps 8 --streams 1 --transcript blake2b --no-cpu-baseline --no-traffic-pass --steady-seconds 0 > $OUT/bench1_${v}_$rep.json 2>/dev/null
  done
done
for v in asm c; do
  cp /tmp/lib_$v.so $L
  python tools/microbench.py > $OUT/microbench_$v.json 2>/dev/null
  python bench.py --config k16 --steps 6 --streams 1 --transcript blake2b --no-traffic-pass --steady-seconds 0 > $OUT/bench_k16_$v.json 2>/dev/null
  python bench.py --config k19 --steps 4 --streams 1 --transcript blake2b --no-traffic-pass --steady-seconds 0 > $OUT/bench_k19_$v.json 2>/dev/null
done
cp /tmp/lib_c.so $L
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6i/bench*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); c=d['config']; r=d['roofline']
        print(f.split('/')[-1], round(d['value'],2), 'ms/step', round(d['ms_per_step'],3), 'steady', c.get('steady_state_proofs_per_s'), 'msm', round(r['avg_launch_ms'],4), round(r['int_alu']['frac'],3), 'ntt', round(r['ntt_tile']['avg_launch_ms'],4), r['ntt_tile']['int_alu_frac'])
    except Exception as e: print(f, 'ERR', e)
P
tail -n 6 $OUT/ntt13_*.txt $OUT/msm96_*.txt $OUT/msm240_*.txt
