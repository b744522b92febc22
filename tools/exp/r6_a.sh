#!/bin/bash
# round 6, first GPU pass: the trimmed suite with durations, the driver's command, configs[3] / [4] under both transcripts with 1-3 proofs in flight
set -u
OUT=gpurun_out/r6b; mkdir -p $OUT
python -m pytest tests -m gpu -x -q --durations=25 > $OUT/pytest.log 2>&1; tail -40 $OUT/pytest.log
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; tail -c 600 $OUT/bench_driver.err
for cfg in k16 k19; do
  steps=8; [ $cfg = k19 ] && steps=6
  for st in 1 2 3; do
    python bench.py --config $cfg --steps $steps --warmup 1 --streams $st --steady-seconds 0 > $OUT/bench_${cfg}_poseidon_s$st.json 2> $OUT/bench_${cfg}_poseidon_s$st.err
  done
  python bench.py --config $cfg --steps $steps --warmup 1 --streams 1 --steady-seconds 0 --transcript blake2b > $OUT/bench_${cfg}_blake2b_s1.json 2> $OUT/bench_${cfg}_blake2b_s1.err
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6b/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); c=d['config']
        print(f.split('/')[-1], round(d['value'],2), 'ms/step', round(d['ms_per_step'],2), 'cold', c.get('cold_key_proofs_per_s'), 'lat', {k:round(v,1) for k,v in c['per_proof_latency_ms'].items()}, 'hostcpu', round(c['host_cpu_ms_per_proof'],1), 'hits', c['host'].get('prefix_cache_hits'), c['host'].get('prefix_cache_misses'))
    except Exception as e: print(f, 'ERR', e)
P
