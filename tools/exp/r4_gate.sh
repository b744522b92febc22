#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r4gate
mkdir -p $OUT
cd $REPO
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --steady-seconds 0"
for rep in 1 2 3; do
  for g in 0 2 3 4 6 8 12; do
    ZKFHE_POSEIDON_X8=0 ZKFHE_GATE=$g $B > $OUT/g${g}_$rep.json 2>/dev/null
  done
done
for g in 0 4 8; do
  ZKFHE_POSEIDON_X8=0 ZKFHE_GATE=$g python bench.py --no-cpu-baseline > $OUT/steady_g${g}.json 2>/dev/null
done
python - <<'PY' > $OUT/summary.txt
import json,glob,os
for f in sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r4gate/*.json"))):
    try:
        d=json.load(open(f)); c=d['config']
        print("%-22s %6.1f steady %s cpu %5.1f lat %s" % (os.path.basename(f), d['value'], c['steady_state_proofs_per_s'] and round(c['steady_state_proofs_per_s'],1), c['host_cpu_ms_per_proof'], {k:round(v,1) for k,v in c['per_proof_latency_ms'].items()}))
    except Exception as e:
        print(f, "unreadable", e)
PY
