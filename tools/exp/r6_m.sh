#!/bin/bash
# round 6, GPU pass 12: hardware queues for the driver's wave of 20 (40 streams with the auxiliary ones): GPU_MAX_HW_QUEUES 16 (default) / 20 / 24,
# alternating on one box (round 2 measured 4 / 8 / 32 against 16: -12 % / -5 % / -25 %)
set -u
OUT=gpurun_out/r6m; mkdir -p $OUT
for rep in 1 2 3 4; do
  for q in 16 20 24; do
    GPU_MAX_HW_QUEUES=$q python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic-pass --steady-seconds 0 > $OUT/wave_q${q}_$rep.json 2>/dev/null
  done
done
for q in 16 20 24; do
  GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu-baseline --no-traffic-pass > $OUT/steady_q${q}.json 2>/dev/null
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6m/*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); c=d['config']; r=d['roofline']
        print(f.split('/')[-1], round(d['value'],2), 'ms/step', round(d['ms_per_step'],3), 'steady', c.get('steady_state_proofs_per_s'), 'lat', {k:round(v,1) for k,v in c['per_proof_latency_ms'].items() if k in ('commit','quotient','open','total','phase0_commitment_back','first_challenge')}, 'hostcpu', round(c['host_cpu_ms_per_proof'],1))
    except Exception as e: print(f, 'ERR', e)
P
