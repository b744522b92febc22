#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r4streams
mkdir -p $OUT
cd $REPO
for rep in 1 2; do
  for st in 8 10 12 16 20; do
    for g in 0 3 4 6; do
      ZKFHE_GATE=$g python bench.py --steps 20 --warmup 5 --streams $st --no-cpu-baseline --steady-seconds 0 > $OUT/s${st}_g${g}_$rep.json 2>/dev/null
    done
  done
done
python - <<'PY' > $OUT/summary.txt
import json,glob,os
for f in sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r4streams/*.json"))):
    try:
        d=json.load(open(f)); c=d['config']
        print("%-18s %6.1f lat %s" % (os.path.basename(f), d['value'], {k:round(v,1) for k,v in c['per_proof_latency_ms'].items()}))
    except Exception as e:
        print(f, "unreadable", e)
PY
