#!/bin/bash
# round 6, GPU pass 11: A/B on ONE box, alternating -- (1) the shipped library (nine-limb products as per-column inline assembly) against the
# same tree built with -DZK_MAD_C (libzkfhe_hip_madc.so: the compiler's form of the products); (2) the driver's command with the
# per-public-key transcript cache on and off (profiles/r6_bench_lines.md had 193.5 against 212.5 from single runs)
set -u
OUT=gpurun_out/r6l; mkdir -p $OUT
L=zk-fhe_amd/libzkfhe_hip.so
cp $L /tmp/lib_asm.so; cp zk-fhe_amd/libzkfhe_hip_madc.so /tmp/lib_c.so
for rep in 1 2 3; do
  for v in asm c; do
    cp /tmp/lib_$v.so $L
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic-pass --steady-seconds 0 > $OUT/bench20_${v}_$rep.json 2>/dev/null
    python bench.py --no-cpu-baseline --no-traffic-pass > $OUT/bench96_${v}_$rep.json 2>/dev/null
    python bench.py --steps 12 --streams 1 --transcript blake2b --no-cpu-baseline --no-traffic-pass --steady-seconds 0 > $OUT/bench1_${v}_$rep.json 2>/dev/null
    [ $rep -le 2 ] && python bench.py --config k16 --steps 8 --streams 1 --transcript blake2b --no-traffic-pass --steady-seconds 0 > $OUT/bench_k16_${v}_$rep.json 2>/dev/null
    [ $rep -le 1 ] && python bench.py --config k19 --steps 5 --streams 1 --transcript blake2b --no-traffic-pass --steady-seconds 0 > $OUT/bench_k19_${v}_$rep.json 2>/dev/null
  done
done
cp /tmp/lib_asm.so $L
for rep in 1 2 3 4; do
  ZKFHE_PREFIX_CACHE=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic-pass --steady-seconds 0 > $OUT/wave_nocache_$rep.json 2>/dev/null
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic-pass --steady-seconds 0 > $OUT/wave_cache_$rep.json 2>/dev/null
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6l/*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); c=d['config']; r=d['roofline']
        print(f.split('/')[-1], round(d['value'],2), 'ms/step', round(d['ms_per_step'],3), 'steady', c.get('steady_state_proofs_per_s'), 'msm', round(r['avg_launch_ms'],4), round(r['int_alu']['frac'],3), 'ntt', round(r['ntt_tile']['avg_launch_ms'],4), 'lat', {k:round(v,1) for k,v in c['per_proof_latency_ms'].items() if k in ('commit','quotient','open','total','phase0_commitment_back','first_challenge')}, 'hostcpu', round(c['host_cpu_ms_per_proof'],1))
    except Exception as e: print(f, 'ERR', e)
P
