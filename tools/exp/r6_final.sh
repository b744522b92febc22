#!/bin/bash
# round 6, final check on a fresh box: build state as committed -> smoke, the whole GPU suite, the driver's command
set -u
OUT=gpurun_out/r6z; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
python -m pytest tests -m gpu -x -q --durations=8 > $OUT/pytest.log 2>&1; tail -14 $OUT/pytest.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err ) 2>&1 | grep real
python - <<'P'
import json
d=json.loads([l for l in open('gpurun_out/r6z/bench_driver.json') if l.startswith('{')][-1]); c=d['config']; r=d['roofline']
print(round(d['value'],2), 'proofs/s', 'steady', c['steady_state_proofs_per_s'], 'cold', c['cold_key_proofs_per_s'], 'verified', c['verified'], 'hostcpu', round(c['host_cpu_ms_per_proof'],1))
print('roofline', r['frac'], r['traffic'], r['traffic_source'][:60], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
P
