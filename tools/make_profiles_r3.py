#!/usr/bin/env python3
"""Turn the outputs of tools/profile_r3.sh (gpurun_out/r3/) into the tracked summaries profiles/r3_*.  Run from the repo root."""
import json
import os

R = 'gpurun_out/r3/'
P = 'profiles/'


def rd(f):
    try:
        return open(R + f).read()
    except OSError:
        return '(missing: %s)\n' % f


def jl(f):
    try:
        return json.loads(rd(f).strip().splitlines()[-1])
    except Exception:  # noqa: BLE001
        return None


def val(txt, k):
    for l in txt.splitlines():
        if l.startswith(k):
            return float(l.split('avg=')[1]), int(l.split('launches=')[1].split()[0])
    return None, None


def head(txt, n):
    return '\n'.join(txt.splitlines()[:n])


a, au = jl('a_driver_bench.json'), jl('bench_driver.json')
lp = rd('b_single_last_proof.txt')
single_avg = None
for l in lp.splitlines():
    if l.startswith('k_msm_table<false>'):
        single_avg = float(l.split()[3]) / int(l.split()[2])
open(P + 'r3_a_driver_kernel_stats.md', 'w').write("""# r3 (a) -- kernel stats of the driver's command (k = 13, Poseidon transcript, one wave of 20 concurrent proofs)

Command (MI355X box): `cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_a_driver -o r -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline`,
summarised with `tools/rocpd_stats.py` (`tools/profile_r3.sh` runs all of these profiles, `tools/make_profiles_r3.py` writes
these files).  The run printed %.1f proofs/s under the profiler; un-profiled `python bench.py --steps 20 --warmup 5` right after:
%.1f proofs/s, steady-state pass %.1f.  With 20 proofs sharing the GPU a launch's duration includes the time its workgroups
wait for CUs, so the averages below are NOT per-kernel costs: those are in `r3_b_single_proof.md`.

Agreement check required by the bench contract: `bench.py` times the `k_msm_table` launches of two extra proofs with HIP
events (nothing else in flight): `avg_launch_ms` = %.3f; the same launches in `r3_b_single_proof.md` (rocprof, one proof in
flight, the two calls of 266 and 136 columns) average %.3f ms.

## All kernels of the run

%s
""" % (a['value'] if a else 0, au['value'] if au else 0, au['config']['steady_state_proofs_per_s'] if au else 0,
       au['roofline']['avg_launch_ms'] if au else 0, single_avg or 0, head(rd('a_driver_kernel_stats.md'), 40)))

b, bs, bp = jl('b_single_bench.json'), jl('bench_single_blake2b.json'), jl('bench_single_poseidon.json')
open(P + 'r3_b_single_proof.md', 'w').write("""# r3 (b) -- one proof in flight (k = 13, Blake2b transcript so that the host hash does not pace the GPU)

`rocprofv3 --kernel-trace --stats -- python bench.py --steps 4 --warmup 1 --streams 1 --no-cpu-baseline --transcript blake2b --steady-seconds 0`
(%.2f ms per proof under the profiler; %.2f ms un-profiled, %.1f ms with the Poseidon transcript: `r3_bench_lines.md`).
Round 2 (`r2b_b_single_proof.md`): 9.19 ms of kernels in an 11.64 ms span, 47 `copyBuffer` + 7 fills, `k_ntt_tile<13>` 1.96 ms.

## Kernels of the last proof (`tools/last_proof_stats.py`)

```
%s```

## Timeline of the same proof (`tools/last_proof_timeline.py`, launches >= 10 us, consecutive launches of a kernel merged)

```
%s```
""" % (b['ms_per_step'] if b else 0, bs['ms_per_step'] if bs else 0, bp['ms_per_step'] if bp else 0, lp, rd('b_single_timeline.txt')))

for tag, name, cfgn in (('c_k16', 'k16', 'BASELINE configs[3]: N = 4096, Q = 2^60 - 93'), ('d_k19', 'k19', 'BASELINE configs[4]: N = 16384, Q = 2^60 - 93')):
    d = jl(tag + '_bench.json')
    note = ("Calls of many columns take the bucket pipeline here (a 48 GB table allows 9-bit digits at n = 2^16: 29 windows against the "
            "pipeline's 19); calls of <= 8 columns take `k_msm_table`.") if name == 'k16' else \
           "No digit-multiple table at n = 2^19 (8-bit digits would need 137 GB per SRS half): every call takes the bucket pipeline."
    open(P + 'r3_%s_kernel_stats.md' % name, 'w').write("""# r3 -- %s (%s), one proof in flight, Blake2b transcript

`rocprofv3 --kernel-trace --stats -- python bench.py --config %s --steps %d --warmup 1 --streams 1 --transcript blake2b --steady-seconds 0`
%.1f ms per proof under the profiler.  Un-profiled: `r3_bench_lines.md`.  %s

## Kernels of the last proof

```
%s```

## All kernels of the run

%s
""" % (name, cfgn, name, d['steps'] if d else 0, d['ms_per_step'] if d else 0, note, rd(tag + '_last_proof.txt'), head(rd(tag + '_kernel_stats.md'), 30)))

fs, ws = rd('pmc_FETCH_SIZE.txt'), rd('pmc_WRITE_SIZE.txt')
f, nl = val(fs, 'k_msm_table<false>')
w, _ = val(ws, 'k_msm_table<false>')
nf, nnl = val(fs, 'k_ntt13')
nw, _ = val(ws, 'k_ntt13')
traffic = {}
if f and w:
    bpl = int(2 * f * 1024 + w * 1024)
    alg = int((266 + 136) / 2 * 8192 * 96)
    traffic = {"kernel": "k_msm_table", "fetch_size_kb_avg": f, "write_size_kb_avg": w, "bytes_per_launch": bpl, "launches": nl,
               "algorithmic_bytes_per_launch": alg,
               "note": "2 x FETCH_SIZE (gfx950: wide loads are tallied at half their bytes, MI355X_MICROARCH.md HBM section) + WRITE_SIZE (uncalibrated); KB units; "
                       "separate --pmc passes; k_msm_table<false> = the two calls of 266 and 136 columns of a k = 13 proof"}
if nf and nw:
    traffic["ntt13"] = {"fetch_size_kb_avg": nf, "write_size_kb_avg": nw, "bytes_per_launch": int(2 * nf * 1024 + nw * 1024), "launches": nnl,
                        "note": "k_ntt13 launches of a k = 13 proof (inverse transform of 408 columns, three coset rows of each, five single-column calls): average"}
json.dump(traffic, open(P + 'r3_pmc_traffic.json', 'w'), indent=1)
open(P + 'r3_pmc.md', 'w').write("""# r3 -- PMC counters (rocprofv3, one or two counters per pass, kernel trace only)

## HBM traffic, k = 13, one proof in flight, Poseidon transcript

`rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --steady-seconds 0`,
the same with `--pmc WRITE_SIZE`; `tools/pmc_stats.py`.  Units: KB per launch, averaged over the launches of the run.  gfx950
correction from MI355X_MICROARCH.md (HBM section): FETCH_SIZE tallies wide (16 B / lane) loads at half their bytes, so the
bytes of a launch are 2 x FETCH_SIZE + WRITE_SIZE (WRITE_SIZE uncalibrated).

```
%s
%s```

## The 2^13 tile alone (`tools/exp/ntt13_bench.py`: 256 columns forward, inverse, and 4 coset rows; `tools/pmc_per_launch.py`)

Algorithmic bytes of a launch of 256 columns (grid 262144 = 512 workgroups of 512 threads): 64 MB read + 64 MB written = 134 MB
(KB units below: 65 536 each way); the coset launch (grid 1048576) reads 64 MB -- every row reads the same coefficients -- and
writes 256 MB.  Each column is read by both of its workgroups: FETCH_SIZE shows how much of the second read came from L2.

```
%s
%s
%s```

## VALU instructions per proof (who uses the ALUs), k = 13, one proof in flight, Blake2b

`rocprofv3 --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES --kernel-trace -- python bench.py --steps 6 --warmup 1 --streams 1 --no-cpu-baseline --transcript blake2b --steady-seconds 0`,
`tools/pmc_per_proof.py <db> 10` (sums over the run / 10 proofs incl. warm-up and the two profiled ones; `k_basis_multiples`,
`k_g1_mul`, `k_basis_table` are the one-time SRS set-up and dominate the totals -- read the other rows against each other).

%s
""" % (fs, ws, rd('ntt13_bench.txt'), rd('ntt13_pmc.txt'), rd('ntt13_resources.txt'), rd('pmc_valu_per_proof.md')))

open(P + 'r3_msm_table.md', 'w').write("""# r3 -- one MSM call: bucket pipeline (bits 0: explicit 13-bit windows, no table) against the digit-multiple table (13-bit digits)

`BITS=0,13 python tools/exp/msm_table_bench.py 13 <columns> <kind>`: n = 2^13 points; `full` = random 248-bit scalars,
`small` = 8-bit, `mixed` = a quarter each of 248-bit / 8-bit / 29-bit / 0-1 columns.  `call` = the whole `zkfhe_msm_batch`
(HIP events around it, best of 5), `summing kernel` = `k_msm_accumulate` resp. `k_msm_table` alone, `adds` = mixed additions.
New in round 3: the fold (`call` - `summing kernel` on the table path) runs 512 threads over the even / odd visits of a column.

```
%s```
""" % rd('msm_calls.txt'))

lines = ["# r3 -- bench lines (un-profiled, MI355X box, `tools/profile_r3.sh` section (f))", "",
         "| command | proofs/s | ms per proof | proofs in flight | steady-state pass | host CPU ms / proof | dominant kernel: avg launch ms, int_alu frac |",
         "|---|---|---|---|---|---|---|"]
for fn, cmd in (('bench_driver', "`python bench.py --steps 20 --warmup 5` (the driver's command)"), ('bench_default', '`python bench.py --no-cpu-baseline`'),
                ('bench_blake2b', '`--transcript blake2b`'), ('bench_blake2b_20', '`--transcript blake2b --steps 20 --warmup 5`'),
                ('bench_single_blake2b', '`--steps 8 --streams 1 --transcript blake2b`'), ('bench_single_poseidon', '`--steps 8 --streams 1`'),
                ('bench_k16_blake2b', '`--config k16 --steps 4 --streams 1 --transcript blake2b`'),
                ('bench_k16_2streams', '`--config k16 --steps 8 --warmup 2 --transcript blake2b` (2 in flight)'),
                ('bench_k16_poseidon', '`--config k16 --steps 4 --streams 1`'), ('bench_k19_blake2b', '`--config k19 --steps 4 --streams 1 --transcript blake2b`'),
                ('bench_k19_poseidon', '`--config k19 --steps 4 --streams 1`')):
    d = jl(fn + '.json')
    if not d:
        lines.append("| %s | (missing) | | | | | |" % cmd)
        continue
    c = d['config']
    r = d['roofline']
    lines.append("| %s | %.2f | %.2f | %s | %s | %.1f | %s: %.3f, %.2f |" % (
        cmd, d['value'], d['ms_per_step'], c['concurrent_proofs_per_gpu'], ('%.1f' % c['steady_state_proofs_per_s']) if c['steady_state_proofs_per_s'] else '-',
        c['host_cpu_ms_per_proof'], r['kernel'], r['avg_launch_ms'], r['int_alu']['frac']))
dd = jl('bench_driver.json')
if dd and dd.get('cpu_baseline'):
    cb = dd['cpu_baseline']
    lines += ["", "`cpu_baseline` of the driver's command: %.3f %s on %s threads (`kind: %s`) -- %s.  Seconds per proof by thread count: %s; "
              "phases of the last proof (ms): %s." % (cb['value'], cb['unit'], cb['cores'], cb['kind'], cb['sample'], cb.get('seconds_per_proof_by_threads'),
                                                      cb.get('phase_ms_last_proof'))]
lines += ["", "The full JSON line of the driver's command:", "", "```", rd('bench_driver.json').strip().splitlines()[-1] if os.path.exists(R + 'bench_driver.json') else '', "```", ""]
open(P + 'r3_bench_lines.md', 'w').write('\n'.join(lines))
open(P + 'r3_microbench.md', 'w').write("# r3 -- micro-benchmarks (`python tools/microbench.py`, MI355X box; the NTT sweep runs out of place: `zkfhe_ntt_batch_to`)\n\n```\n" + rd('microbench.json') + "```\n")
print(open(P + 'r3_bench_lines.md').read()[:3000])
