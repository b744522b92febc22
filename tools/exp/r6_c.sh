#!/bin/bash
# round 6, GPU pass 3: announced proofs (tests + bench lines), timing laps of the large configurations, the quotient arithmetic probe,
# the driver's command with its in-run PMC traffic pass
set -u
OUT=gpurun_out/r6c; mkdir -p $OUT
python -m pytest tests/test_gpu_prover.py -m gpu -x -q -s -k "announced or config4 or config5 or prefix_cache" > $OUT/pytest_sel.log 2>&1; grep -E "config[45]:|passed|failed|Error" $OUT/pytest_sel.log | tail -40
hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=1000000 -I zk-fhe_amd/csrc tools/exp/quotient_gate_probe.hip -o /tmp/qprobe 2> $OUT/qprobe_build.log
for k in 13 16 19; do /tmp/qprobe $k; done > $OUT/quotient_probe.txt 2>&1; cat $OUT/quotient_probe.txt
( time python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err ) 2>&1 | grep real; tail -c 1500 $OUT/bench_driver.err
for cfg in k16 k19; do
  python bench.py --config $cfg --steps 8 --warmup 2 --streams 1 --steady-seconds 0 > $OUT/bench_${cfg}_poseidon_s1_announced.json 2> $OUT/bench_${cfg}_poseidon_s1_announced.err
  python bench.py --config $cfg --steps 8 --warmup 2 --streams 2 --steady-seconds 0 > $OUT/bench_${cfg}_poseidon_s2_announced.json 2> $OUT/bench_${cfg}_poseidon_s2_announced.err
  python bench.py --config $cfg --steps 8 --warmup 2 --streams 1 --steady-seconds 0 --announce off > $OUT/bench_${cfg}_poseidon_s1_plain.json 2>/dev/null
done
python bench.py --steps 16 --warmup 2 --streams 1 --steady-seconds 0 --no-cpu-baseline --no-traffic-pass > $OUT/bench_k13_poseidon_s1_announced.json 2>/dev/null
python bench.py --steps 16 --warmup 2 --streams 1 --steady-seconds 0 --no-cpu-baseline --no-traffic-pass --announce off > $OUT/bench_k13_poseidon_s1_plain.json 2>/dev/null
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6c/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); c=d['config']
        print(f.split('/')[-1], round(d['value'],2), 'ms/step', round(d['ms_per_step'],2), 'cold', c.get('cold_key_proofs_per_s'), 'lat', {k:round(v,1) for k,v in c['per_proof_latency_ms'].items()}, 'hostcpu', round(c['host_cpu_ms_per_proof'],1), 'announce', c.get('inputs_announced_one_proof_ahead'))
        if 'driver' in f: print(json.dumps(d['roofline'])[:1500])
    except Exception as e: print(f, 'ERR', e)
P
