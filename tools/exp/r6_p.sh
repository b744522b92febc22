#!/bin/bash
# round 6, GPU pass 15: the phase-0 / RLC upload as a kernel that reads the pinned table (default now) against the copy command it replaces
# (ZKFHE_UPLOAD=copy), the driver's wave, ten traced runs each, alternating; then parity of the new path (proof bytes against the oracle) and one
# proof alone / steady state / k19 both ways
set -u
OUT=gpurun_out/r6p; mkdir -p $OUT
python -m pytest tests/test_gpu_prover.py -m gpu -x -q -k "toy or bfv_in_k13 or k14 or config4 or announced" > $OUT/pytest_prover.log 2>&1; tail -2 $OUT/pytest_prover.log
for rep in 1 2 3 4 5 6 7 8 9 10; do
  ZKFHE_UPLOAD=copy ZKFHE_TRACE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic-pass --steady-seconds 0 > $OUT/wave_copy_$rep.json 2> $OUT/wave_copy_$rep.err
  ZKFHE_TRACE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic-pass --steady-seconds 0 > $OUT/wave_kernel_$rep.json 2> $OUT/wave_kernel_$rep.err
done
python tools/exp/wave_trace.py $OUT | cut -c1-250
for v in copy kernel; do
  [ $v = copy ] && export ZKFHE_UPLOAD=copy || unset ZKFHE_UPLOAD
  python bench.py --no-cpu-baseline --no-traffic-pass > $OUT/steady_$v.json 2>/dev/null
  python bench.py --steps 12 --streams 1 --transcript blake2b --no-cpu-baseline --no-traffic-pass --steady-seconds 0 > $OUT/single_$v.json 2>/dev/null
  python bench.py --config k19 --steps 4 --streams 1 --transcript blake2b --no-traffic-pass --steady-seconds 0 > $OUT/k19_$v.json 2>/dev/null
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6p/s*.json')+glob.glob('gpurun_out/r6p/k19*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); c=d['config']
        print(f.split('/')[-1], round(d['value'],2), 'ms/step', round(d['ms_per_step'],3), 'steady', c.get('steady_state_proofs_per_s'), 'hostcpu', round(c['host_cpu_ms_per_proof'],1))
    except Exception as e: print(f, 'ERR', e)
P
