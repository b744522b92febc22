#!/bin/bash
# round 6, GPU pass 16: two old switches re-measured on the round's kernels and host hash -- the early phase-1 commitment (ZKFHE_EARLY_P1=0 / default on
# with Poseidon: +9 % on the wave in round 3 with a 6.3 us permutation, +3 % with 4.7 us; the core is at 3.3 us now) and the elements per
# inversion of k_fr_batch_invert (ZKFHE_BI_CHUNK: 8 / 16 by array length today) -- wave and steady state, alternating
set -u
OUT=gpurun_out/r6q; mkdir -p $OUT
for rep in 1 2 3 4 5 6; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic-pass --steady-seconds 0 > $OUT/wave_base_$rep.json 2>/dev/null
  ZKFHE_EARLY_P1=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic-pass --steady-seconds 0 > $OUT/wave_noearly_$rep.json 2>/dev/null
  ZKFHE_BI_CHUNK=32 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic-pass --steady-seconds 0 > $OUT/wave_bi32_$rep.json 2>/dev/null
done
for rep in 1 2 3; do
  python bench.py --no-cpu-baseline --no-traffic-pass > $OUT/steady_base_$rep.json 2>/dev/null
  ZKFHE_EARLY_P1=0 python bench.py --no-cpu-baseline --no-traffic-pass > $OUT/steady_noearly_$rep.json 2>/dev/null
  ZKFHE_BI_CHUNK=32 python bench.py --no-cpu-baseline --no-traffic-pass > $OUT/steady_bi32_$rep.json 2>/dev/null
  ZKFHE_BI_CHUNK=64 python bench.py --no-cpu-baseline --no-traffic-pass > $OUT/steady_bi64_$rep.json 2>/dev/null
done
python - <<'P'
import json,glob,re,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r6q/*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); c=d['config']
        k=re.sub(r'_\d+\.json$','',f.split('/')[-1]); acc[k].append((d['value'], c.get('steady_state_proofs_per_s') or 0))
    except Exception as e: print(f, 'ERR', e)
for k,v in acc.items():
    print('%-16s'%k, 'value', [round(x[0],1) for x in v], 'mean %.1f'%(sum(x[0] for x in v)/len(v)), 'steady', [round(x[1],1) for x in v], 'mean %.1f'%(sum(x[1] for x in v)/len(v)))
P
