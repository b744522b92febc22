#!/bin/bash
# round 6, GPU pass 9: second A/B of the per-column tied products (libzkfhe_hip_tied.so) against the C form: the large configurations, one proof
# alone and the driver's wave, alternating on one box
set -u
OUT=gpurun_out/r6j; mkdir -p $OUT
L=zk-fhe_amd/libzkfhe_hip.so
cp zk-fhe_amd/libzkfhe_hip_tied.so /tmp/lib_asm.so; cp $L /tmp/lib_c.so
for rep in 1 2 3; do
  for v in asm c; do
    cp /tmp/lib_$v.so $L
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic-pass --steady-seconds 0 > $OUT/bench20_${v}_$rep.json 2>/dev/null
    python bench.py --steps 12 --streams 1 --transcript blake2b --no-cpu-baseline --no-traffic-pass --steady-seconds 0 > $OUT/bench1_${v}_$rep.json 2>/dev/null
    python bench.py --config k16 --steps 8 --streams 1 --transcript blake2b --no-traffic-pass --steady-seconds 0 > $OUT/bench_k16_${v}_$rep.json 2>/dev/null
    [ $rep -le 2 ] && python bench.py --config k19 --steps 5 --streams 1 --transcript blake2b --no-traffic-pass --steady-seconds 0 > $OUT/bench_k19_${v}_$rep.json 2>/dev/null
  done
done
cp /tmp/lib_c.so $L
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6j/bench*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); c=d['config']; r=d['roofline']
        print(f.split('/')[-1], round(d['value'],2), 'ms/step', round(d['ms_per_step'],3), 'msm', round(r['avg_launch_ms'],4), round(r['int_alu']['frac'],3), 'ntt', round(r['ntt_tile']['avg_launch_ms'],4), 'lat', {k:round(v,1) for k,v in c['per_proof_latency_ms'].items() if k in ('commit','quotient','open','total')})
    except Exception as e: print(f, 'ERR', e)
P
