#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r4p0
mkdir -p $OUT
cd $REPO
python -m pytest tests/test_gpu_prover.py -m gpu -x -q 2>&1 | tail -6 > $OUT/tests.log
for cfg in k13 k16 k19; do
  ZKFHE_TRACE=1 ZKFHE_TRACE0=1 python bench.py --config $cfg --steps 3 --warmup 1 --streams 1 --transcript blake2b --steady-seconds 0 --no-cpu-baseline > $OUT/${cfg}.json 2> $OUT/${cfg}.trace
done
for rep in 1 2; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/wave_$rep.json 2>/dev/null
  ZKFHE_PHASE0=generic python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/wave_generic_$rep.json 2>/dev/null
done
python -m pytest tests/test_batch_gloo.py -m gpu -x -q -k "8" 2>&1 | tail -6 > $OUT/tests_world8.log
python - <<'PY' > $OUT/summary.txt
import json,glob,os
for f in sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r4p0/*.json"))):
    try:
        d=json.load(open(f)); c=d['config']
        print("%-22s %6.1f ms/step %.2f steady %s cpu %.1f lat %s" % (os.path.basename(f), d['value'], d['ms_per_step'], c['steady_state_proofs_per_s'] and round(c['steady_state_proofs_per_s'],1), c['host_cpu_ms_per_proof'], {k:round(v,1) for k,v in c['per_proof_latency_ms'].items()}))
    except Exception as e:
        print(f, "unreadable", e)
PY
