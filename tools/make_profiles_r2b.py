#!/usr/bin/env python3
"""Turn the outputs of tools/profile_r2b.sh (gpurun_out/r2b/) into the tracked summaries profiles/r2b_*.  Run from the repo root."""
import json

R = 'gpurun_out/r2b/'
P = 'profiles/'


def rd(f):
    return open(R + f).read()


def jl(f):
    return json.loads(rd(f).strip().splitlines()[-1])


def val(txt, k):
    for l in txt.splitlines():
        if l.startswith(k):
            return float(l.split('avg=')[1]), int(l.split('launches=')[1].split()[0])
    return None, None


a = jl('a_driver_bench.json')
au = jl('bench_driver.json')
lp = rd('b_single_last_proof.txt')
single_avg = None
for l in lp.splitlines():
    if l.startswith('k_msm_table<false>'):
        single_avg = float(l.split()[3]) / int(l.split()[2])
open(P + 'r2b_a_driver_kernel_stats.md', 'w').write("""# r2b (a) -- kernel stats of the driver's command (k = 13, Poseidon transcript, one wave of 20 concurrent proofs)

Command (MI355X box): `cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_a_driver -o r -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline`,
summarised with `tools/rocpd_stats.py` (`tools/profile_r2b.sh` runs all of these profiles, `tools/make_profiles_r2b.py` writes
these files).  The run printed %.1f proofs/s under the profiler; un-profiled `python bench.py --steps 20 --warmup 5` right after:
%.1f proofs/s, steady-state pass %.1f.  With 20 proofs sharing the GPU a launch's duration includes the time its workgroups
wait for CUs, so the averages below are NOT per-kernel costs: those are in `r2b_b_single_proof.md`.

Agreement check required by the bench contract: `bench.py` times the `k_msm_table` launches of two extra proofs with HIP
events (nothing else in flight): `avg_launch_ms` = %.3f; the same launches in `r2b_b_single_proof.md` (rocprof, one proof in
flight, the two calls of 266 and 136 columns) average %.3f ms.

## All kernels of the run

%s
""" % (a['value'], au['value'], au['config']['steady_state_proofs_per_s'], au['roofline']['avg_launch_ms'], single_avg,
       '\n'.join(rd('a_driver_kernel_stats.md').splitlines()[:34])))

a2 = jl('a2_steady_bench.json')
open(P + 'r2b_a2_steady_state.md', 'w').write("""# r2b (a2) -- 192 proofs, 16 in flight, Blake2b transcript: who holds the GPU in steady state

`rocprofv3 --kernel-trace --stats -- python bench.py --steps 192 --warmup 4 --no-cpu-baseline --transcript blake2b --steady-seconds 0`
(%.1f proofs/s under the profiler), `tools/timeline_share.py <db> 200 700`: the last 700 ms of the trace, elapsed time split
equally among the kernels executing at each moment (a kernel that fills 9 CUs counts like one that fills 256 -- this is a
view of the timeline, not of the ALU; the instruction view is `r2b_pmc.md`).

%s

## All kernels of the run

%s
""" % (a2['value'], rd('a2_steady_timeline_share.md'), '\n'.join(rd('a2_steady_kernel_stats.md').splitlines()[:30])))

b = jl('b_single_bench.json')
bs = jl('bench_single_blake2b.json')
bp = jl('bench_single_poseidon.json')
open(P + 'r2b_b_single_proof.md', 'w').write("""# r2b (b) -- one proof in flight (k = 13, Blake2b transcript so that the host hash does not pace the GPU)

`rocprofv3 --kernel-trace --stats -- python bench.py --steps 4 --warmup 1 --streams 1 --no-cpu-baseline --transcript blake2b --steady-seconds 0`
(%.2f ms per proof under the profiler; %.2f ms un-profiled, %.1f ms with the Poseidon transcript: `r2b_bench_lines.md`).

## Kernels of the last proof (`tools/last_proof_stats.py`)

```
%s```

Before the table path (`r2_b_single_proof.md`): 11.06 ms of kernels, 13.96 ms span.  What moved: the thirteen launches of the
bucket pipeline (accumulate 1.89 + scatter / hist / task lists / merge / marginals / weighted 2.1 ms for the two wide calls) are
two `k_msm_table` launches + two folds; the five calls of 1-3 columns (`k_msm_direct`, 64 four-bit windows: 1.57 ms) are
`k_msm_table<true>` with 20 thirteen-bit windows; the two phase-0 polynomial products left the GPU (`host/poly_ntt64.hpp`).

## Timeline of the same proof (`tools/last_proof_timeline.py`, launches >= 10 us, consecutive launches of a kernel merged)

```
%s```
""" % (b['ms_per_step'], bs['ms_per_step'], bp['ms_per_step'], lp, rd('b_single_timeline.txt')))

for tag, name, cfgn in (('c_k16', 'k16', 'BASELINE configs[3]): N = 4096, Q = 2^60 - 93'), ('d_k19', 'k19', 'BASELINE configs[4]): N = 16384, Q = 2^60 - 93')):
    d = jl(tag + '_bench.json')
    note = ("Calls of many columns take the bucket pipeline here (a 48 GB table allows 9-bit digits at n = 2^16: 29 windows against the "
            "pipeline's 19); calls of <= 8 columns take `k_msm_table`.") if name == 'k16' else \
           "No digit-multiple table at n = 2^19 (8-bit digits would need 137 GB per SRS half): every call takes the bucket pipeline."
    open(P + 'r2b_%s_kernel_stats.md' % name, 'w').write("""# r2b -- %s (%s, one proof in flight, Blake2b transcript

`rocprofv3 --kernel-trace --stats -- python bench.py --config %s --steps %d --warmup 1 --streams 1 --transcript blake2b --steady-seconds 0`
%.1f ms per proof under the profiler.  Un-profiled: `r2b_bench_lines.md`.  %s

## Kernels of the last proof

```
%s```

## All kernels of the run

%s
""" % (name, cfgn, name, d['steps'], d['ms_per_step'], note, rd(tag + '_last_proof.txt'), '\n'.join(rd(tag + '_kernel_stats.md').splitlines()[:28])))

fs = rd('pmc_FETCH_SIZE.txt')
ws = rd('pmc_WRITE_SIZE.txt')
f, nl = val(fs, 'k_msm_table<false>')
w, _ = val(ws, 'k_msm_table<false>')
bpl = int(2 * f * 1024 + w * 1024)
alg = int((266 + 136) / 2 * 8192 * 96)
json.dump({"kernel": "k_msm_table", "fetch_size_kb_avg": f, "write_size_kb_avg": w, "bytes_per_launch": bpl, "launches": nl,
           "algorithmic_bytes_per_launch": alg,
           "note": "2 x FETCH_SIZE (gfx950: wide loads are tallied at half their bytes, MI355X_MICROARCH.md HBM section) + WRITE_SIZE (uncalibrated); KB units; "
                   "separate --pmc passes; k_msm_table<false> = the two calls of 266 and 136 columns of a k = 13 proof"},
          open(P + 'r2b_pmc_traffic.json', 'w'), indent=1)
open(P + 'r2b_pmc.md', 'w').write("""# r2b -- PMC counters (rocprofv3, one or two counters per pass, kernel trace only)

## HBM traffic, k = 13, one proof in flight, Poseidon transcript

`rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --steady-seconds 0`,
the same with `--pmc WRITE_SIZE`; `tools/pmc_stats.py`.  Units: KB per launch, averaged over the launches of the run.  gfx950
correction from MI355X_MICROARCH.md (HBM section): FETCH_SIZE tallies wide (16 B / lane) loads at half their bytes, so the
bytes of a launch are 2 x FETCH_SIZE + WRITE_SIZE (WRITE_SIZE uncalibrated).

```
%s
%s```

`k_msm_table<false>` (the two wide calls of a proof): 2 x %.0f MB + %.0f MB = **%.2f GB per launch**, %.1f x the algorithmic
%.0f MB (96 B per scalar-point pair).  Expected: every mixed addition gathers its own 64-byte table point from a 43 GB table
(12.6 M additions per launch = 0.8 GB), nothing of the table is reused within a launch.  At 1.15 ms per launch that is
1.6 TB/s -- 20 %% of the HBM peak; the kernel is bound by multiply-add issue (`int_alu` in the bench line), the table
traffic is what the absence of sort, buckets and bucket reduction costs.

## VALU instructions per proof (who uses the ALUs), k = 13, one proof in flight, Blake2b

`rocprofv3 --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES --kernel-trace -- python bench.py --steps 6 --warmup 1 --streams 1 --no-cpu-baseline --transcript blake2b --steady-seconds 0`,
`tools/pmc_per_proof.py <db> 10` (sums over the run / 10 proofs incl. warm-up and the two profiled ones; `k_basis_multiples`,
`k_g1_mul`, `k_basis_table` are the one-time SRS set-up and dominate the totals -- read the other rows against each other).

%s
""" % (fs, ws, f / 1024, w / 1024, bpl / 1e9, bpl / alg, alg / 1e6, rd('pmc_valu_per_proof.md')))

open(P + 'r2b_msm_table.md', 'w').write("""# r2b -- one MSM call: bucket pipeline (bits 0: explicit 13-bit windows, no table) against the digit-multiple table (12- and 13-bit digits)

`BITS=0,12,13 python tools/exp/msm_table_bench.py 13 <columns> <kind>`: n = 2^13 points; `full` = random 248-bit scalars,
`small` = 8-bit, `mixed` = a quarter each of 248-bit / 8-bit / 29-bit / 0-1 columns.  `call` = the whole `zkfhe_msm_batch`
(HIP events around it, best of 5), `summing kernel` = `k_msm_accumulate` resp. `k_msm_table` alone, `adds` = mixed additions.

```
%s```

VALU instructions per call (`rocprofv3 --pmc SQ_INSTS_VALU`, sums over 8 calls / 8; bucket-pipeline kernels and table-path
kernels of the same run):

```
%s```

96 full-width columns: table path (`k_msm_table<false>` + fold) 0.89 x the instructions of the bucket pipeline's kernels; 240
eight-bit columns: 0.75 x; a lone column: 0.84 x and 0.21 ms instead of 0.53 ms.  What the 20-proofs-in-flight throughput
showed while this path was built (same binary, `bench.py --transcript blake2b --steps 192`): bucket pipeline for every call
187 proofs/s; table path with the entry list in LDS 176 (the NTT tile kernel needs 147 of a CU's 160 KB: they could not share a
CU); list in global memory 182; lone columns in 64 workgroups instead of 256 and butterflies across waves through LDS first
(9 wave-wide additions instead of 26) 210.
""" % (rd('msm_calls.txt'), rd('msm_calls_valu.txt')))

lines = ["# r2b -- bench lines (un-profiled, MI355X box, `tools/profile_r2b.sh` section (f))", "",
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
    c = d['config']
    r = d['roofline']
    lines.append("| %s | %.2f | %.2f | %s | %s | %.1f | %s: %.3f, %.2f |" % (
        cmd, d['value'], d['ms_per_step'], c['concurrent_proofs_per_gpu'], ('%.1f' % c['steady_state_proofs_per_s']) if c['steady_state_proofs_per_s'] else '-',
        c['host_cpu_ms_per_proof'], r['kernel'], r['avg_launch_ms'], r['int_alu']['frac']))
lines += ["", "The full JSON line of the driver's command:", "", "```", rd('bench_driver.json').strip().splitlines()[-1], "```", ""]
open(P + 'r2b_bench_lines.md', 'w').write('\n'.join(lines))
open(P + 'r2b_microbench.md', 'w').write("# r2b -- micro-benchmarks (`python tools/microbench.py`, MI355X box)\n\n```\n" + rd('microbench.json') + "```\n")
print(open(P + 'r2b_bench_lines.md').read()[:2600])
