#!/bin/bash
# round 6, GPU pass 7: where the HOST CPU of a proof goes (ZKFHE_TRACE with per-phase thread CPU time), latency and shared hashing modes
set -u
OUT=gpurun_out/r6g; mkdir -p $OUT
for tr in blake2b poseidon; do
  ZKFHE_TRACE=1 python bench.py --steps 3 --warmup 1 --streams 1 --transcript $tr --no-cpu-baseline --no-traffic-pass --steady-seconds 0 --announce off > $OUT/trace_$tr.json 2> $OUT/trace_$tr.err
  echo "== $tr"; grep "zkfhe trace" $OUT/trace_$tr.err | tail -17
done
python - <<'P'
import json
for tr in ("blake2b","poseidon"):
    d=json.loads([l for l in open("gpurun_out/r6g/trace_%s.json"%tr) if l.startswith("{")][-1])
    print(tr, "host_cpu_ms_per_proof", d["config"]["host_cpu_ms_per_proof"], "ms/step", d["ms_per_step"])
P
