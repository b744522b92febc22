#!/bin/bash
# fold fused into k_msm_table: parity, then call-level and proof-level timings for ZKFHE_MSM_FUSE = 0 / 1 / 2
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r4fuse
mkdir -p $OUT
cd $REPO
for f in 1 2; do
  ZKFHE_MSM_FUSE=$f python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "msm" 2>&1 | tail -3 > $OUT/tests_fuse$f.log
done
ZKFHE_MSM_FUSE=2 python -m pytest tests/test_gpu_prover.py -m gpu -x -q -k "k13 or toy or twelve" 2>&1 | tail -3 >> $OUT/tests_fuse2.log
for f in 0 1 2; do
  for wl in "96 full" "240 small" "136 full" "1 full" "3 full"; do
    echo "== fuse $f: $wl" >> $OUT/msm_calls.txt
    ZKFHE_MSM_FUSE=$f BITS=13 python tools/exp/msm_table_bench.py 13 $wl 2>/dev/null >> $OUT/msm_calls.txt
  done
done
for rep in 1 2; do
  for f in 0 1 2; do
    ZKFHE_MSM_FUSE=$f python bench.py --steps 8 --streams 1 --transcript blake2b --no-cpu-baseline --steady-seconds 0 > $OUT/single_f${f}_$rep.json 2>/dev/null
    ZKFHE_MSM_FUSE=$f python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/wave_f${f}_$rep.json 2>/dev/null
  done
done
python - <<'PY' > $OUT/summary.txt
import json,glob,os
for f in sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r4fuse/*.json"))):
    try:
        d=json.load(open(f)); c=d['config']
        print("%-22s %6.1f ms/step %.2f steady %s lat %s" % (os.path.basename(f), d['value'], d['ms_per_step'], c['steady_state_proofs_per_s'] and round(c['steady_state_proofs_per_s'],1), {k:round(v,1) for k,v in c['per_proof_latency_ms'].items()}))
    except Exception as e:
        print(f, "unreadable", e)
PY
