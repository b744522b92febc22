#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r4p0b
mkdir -p $OUT
cd $REPO
python -m pytest tests/test_batch_gloo.py -m gpu -x -q -k "rccl or one_rank" 2>&1 | tail -6 > $OUT/tests_rccl.log
python -m pytest tests/test_bench_ranks.py -m gpu -x -q 2>&1 | tail -6 > $OUT/tests_bench_ranks.log
python -m pytest tests/test_gpu_prover.py -m gpu -x -q -k "config4 or config5 or k16 or k19 or 60" 2>&1 | tail -6 > $OUT/tests_big.log
for cfg in k13 k16 k19; do
  ZKFHE_TRACE=1 ZKFHE_TRACE0=1 python bench.py --config $cfg --steps 3 --warmup 1 --streams 1 --transcript blake2b --steady-seconds 0 --no-cpu-baseline > $OUT/${cfg}.json 2> $OUT/${cfg}.trace
done
python bench.py --steps 3 --warmup 1 --streams 1 --steady-seconds 0 --no-cpu-baseline > $OUT/k13_poseidon_single.json 2>/dev/null
for rep in 1 2 3; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/wave_$rep.json 2>/dev/null
done
/opt/rocm/lib/llvm/bin/clang++ -O3 -std=c++17 -I zk-fhe_amd/host tools/exp/poseidon_ifma_check.cpp zk-fhe_amd/host/poseidon_ifma.cpp -o /tmp/pic && /tmp/pic > $OUT/poseidon_single.txt
python - <<'PY' > $OUT/summary.txt
import json,glob,os
for f in sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r4p0b/*.json"))):
    try:
        d=json.load(open(f)); c=d['config']
        print("%-26s %6.1f ms/step %.2f steady %s cpu %.1f lat %s" % (os.path.basename(f), d['value'], d['ms_per_step'], c['steady_state_proofs_per_s'] and round(c['steady_state_proofs_per_s'],1), c['host_cpu_ms_per_proof'], {k:round(v,1) for k,v in c['per_proof_latency_ms'].items()}))
    except Exception as e:
        print(f, "unreadable", e)
PY
