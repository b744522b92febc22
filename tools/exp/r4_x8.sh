#!/bin/bash
# eight-lane Poseidon service: parity, timing on the box's CPU, effect on the wave and on steady state
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r4x8
mkdir -p $OUT
cd $REPO
/opt/rocm/lib/llvm/bin/clang++ -O3 -std=c++17 -I zk-fhe_amd/host tools/exp/poseidon_x8_check.cpp zk-fhe_amd/host/poseidon_x8.cpp zk-fhe_amd/host/poseidon_ifma.cpp -o /tmp/px8 -lpthread 2>$OUT/px8.err && /tmp/px8 > $OUT/px8.txt 2>&1
ZKFHE_HASH_THREADS=1 /tmp/px8 > $OUT/px8_1thread.txt 2>&1
python -m pytest tests/test_poseidon.py -q 2>&1 | tail -3 > $OUT/tests.log
python -m pytest tests/test_gpu_prover.py -m gpu -x -q 2>&1 | tail -5 >> $OUT/tests.log
run() { # name, env...
  local name=$1; shift
  env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/$name.json 2>/dev/null
  env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --steady-seconds 0 > $OUT/${name}_b.json 2>/dev/null
}
run off ZKFHE_POSEIDON_X8=0
run on3 ZKFHE_HASH_THREADS=3
run on2 ZKFHE_HASH_THREADS=2
run on4 ZKFHE_HASH_THREADS=4
run on3_noearly ZKFHE_HASH_THREADS=3 ZKFHE_EARLY_P1=0
run on3_min1 ZKFHE_HASH_THREADS=3 ZKFHE_X8_MIN=1
env ZKFHE_HASH_THREADS=3 python tools/exp/cpu_per_proof.py > $OUT/cpu_per_proof_on.txt 2>&1
env ZKFHE_POSEIDON_X8=0 python tools/exp/cpu_per_proof.py > $OUT/cpu_per_proof_off.txt 2>&1
python - <<'PY' > $OUT/summary.txt
import json,glob,os
for f in sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r4x8/*.json"))):
    try:
        d=json.load(open(f)); c=d['config']
        print("%-22s %6.1f steady %s cpu %5.1f lat %s" % (os.path.basename(f), d['value'], c['steady_state_proofs_per_s'] and round(c['steady_state_proofs_per_s'],1), c['host_cpu_ms_per_proof'], {k:round(v,1) for k,v in c['per_proof_latency_ms'].items()}))
    except Exception as e:
        print(f, "unreadable", e)
PY
