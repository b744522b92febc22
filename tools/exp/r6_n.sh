#!/bin/bash
# round 6, GPU pass 13: why two runs of the driver's command on one box differ by 10 % (209 / 236): six traced runs (ZKFHE_TRACE=1: every
# proof's host phases with wall and thread-CPU time and the absolute clock), the timed region's bounds and the cgroup's throttle counters
set -u
OUT=gpurun_out/r6n; mkdir -p $OUT
cat /sys/fs/cgroup/cpu.max > $OUT/host.txt 2>&1; nproc >> $OUT/host.txt; cat /sys/fs/cgroup/cpu.stat >> $OUT/host.txt 2>&1
for rep in 1 2 3 4 5 6; do
  ZKFHE_TRACE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic-pass --steady-seconds 0 > $OUT/wave_$rep.json 2> $OUT/wave_$rep.err
done
cat /sys/fs/cgroup/cpu.stat >> $OUT/host.txt 2>&1
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6n/*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); c=d['config']
        print(f.split('/')[-1], round(d['value'],2), 'lat', {k:round(v,1) for k,v in c['per_proof_latency_ms'].items()}, 'hostcpu', round(c['host_cpu_ms_per_proof'],1))
    except Exception as e: print(f, 'ERR', e)
P
grep "bench trace" $OUT/*.err
