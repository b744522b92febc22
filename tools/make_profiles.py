#!/usr/bin/env python3
"""Turn the outputs of `bash tools/profile.sh <round>` (gpurun_out/r<round>/) into the tracked summaries profiles/r<round>_*.
One generator for every round (rounds 2-5 each had a copy of this file that differed by file names):

    python tools/make_profiles.py --round 6        # from the repo root
"""
import argparse
import json
import os

_ap = argparse.ArgumentParser()
_ap.add_argument("--round", required=True, help="round tag: 6 -> gpurun_out/r6/ -> profiles/r6_*")
_ap.add_argument("--calibration", default="r5_pmc_calibration.md", help="the PMC calibration file the read factors come from")
_args = _ap.parse_args()
RT = 'r%s' % _args.round          # the tag of this round's files
CAL = _args.calibration
R = 'gpurun_out/%s/' % RT
P = 'profiles/'


def W(name, text):
    """profiles/<round tag><name>; @RT@ / @CAL@ in the text = this round's tag / the calibration file"""
    open(P + RT + name, 'w').write(text.replace('@RT@', RT).replace('@CAL@', CAL))


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
W('_a_driver_kernel_stats.md', """# @RT@ (a) -- kernel stats of the driver's command (k = 13, Poseidon transcript, one wave of 20 concurrent proofs)

Command (MI355X box): `cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_a_driver -o r -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline`,
summarised with `tools/rocpd_stats.py` (`tools/profile.sh` runs all of these profiles, `tools/make_profiles.py` writes
these files).  The run printed %.1f proofs/s under the profiler; un-profiled `python bench.py --steps 20 --warmup 5` right after:
%.1f proofs/s, steady-state pass %.1f.  With 20 proofs sharing the GPU a launch's duration includes the time its workgroups
wait for CUs, so the averages below are NOT per-kernel costs: those are in `@RT@_b_single_proof.md`.

Agreement check required by the bench contract: `bench.py` times the `k_msm_table` launches of two extra proofs with HIP
events (nothing else in flight): `avg_launch_ms` = %.3f; the same launches in `@RT@_b_single_proof.md` (rocprof, one proof in
flight, the two calls of 266 and 136 columns) average %.3f ms.

## All kernels of the run

%s
""" % (a['value'] if a else 0, au['value'] if au else 0, au['config']['steady_state_proofs_per_s'] if au else 0,
       au['roofline']['avg_launch_ms'] if au else 0, single_avg or 0, head(rd('a_driver_kernel_stats.md'), 40)))

b, bs, bp = jl('b_single_bench.json'), jl('bench_single_blake2b.json'), jl('bench_single_poseidon.json')
W('_b_single_proof.md', """# @RT@ (b) -- one proof in flight (k = 13, Blake2b transcript so that the host hash does not pace the GPU)

`rocprofv3 --kernel-trace --stats -- python bench.py --steps 4 --warmup 1 --streams 1 --no-cpu-baseline --transcript blake2b --steady-seconds 0`
(%.2f ms per proof under the profiler; %.2f ms un-profiled, %.1f ms with the Poseidon transcript: `@RT@_bench_lines.md`).

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
    W('_%s_kernel_stats.md' % name, """# @RT@ -- %s (%s), one proof in flight, Blake2b transcript

`rocprofv3 --kernel-trace --stats -- python bench.py --config %s --steps %d --warmup 1 --streams 1 --transcript blake2b --steady-seconds 0`
%.1f ms per proof under the profiler.  Un-profiled: `@RT@_bench_lines.md`.  %s

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
    # calibrated factors (profiles/@CAL@): scattered 64-byte gathers are counted at their size, coalesced 16 / 32-byte
    # reads at half of it, writes at their size.  A launch reads its scalars coalesced (32 B x 8192 x columns: counted at half, the
    # other half is added back) and gathers table points (x 1).
    scalars = (266 + 136) / 2 * 8192 * 32
    bpl = int(f * 1024 + scalars / 2 + w * 1024)
    alg = int((266 + 136) / 2 * 8192 * 96)
    traffic = {"kernel": "k_msm_table", "fetch_size_kb_avg": f, "write_size_kb_avg": w, "bytes_per_launch": bpl, "launches": nl,
               "algorithmic_bytes_per_launch": alg,
               "note": "FETCH_SIZE x 1 (64-byte table gathers are counted at their size: profiles/@CAL@) + half the streamed scalar bytes "
                       "(coalesced 32-byte reads are counted at half) + WRITE_SIZE x 1; KB units; separate --pmc passes; k_msm_table<false> = the two calls of 266 and 136 "
                       "columns of a k = 13 proof"}
if nf and nw:
    traffic["ntt13"] = {"fetch_size_kb_avg": nf, "write_size_kb_avg": nw, "bytes_per_launch": int(2 * nf * 1024 + nw * 1024), "launches": nnl,
                        "note": "k_ntt13 launches of a k = 13 proof (inverse transform of 408 columns, three coset rows of each, single-column calls): average; "
                                "2 x FETCH_SIZE (coalesced 32-byte reads) + WRITE_SIZE"}
open(P + RT + '_pmc_traffic.json', 'w').write(json.dumps(traffic, indent=1).replace('@CAL@', CAL))
W('_pmc.md', """# @RT@ -- PMC counters (rocprofv3, one or two counters per pass, kernel trace only)

## HBM traffic, k = 13, one proof in flight, Poseidon transcript

`rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --steady-seconds 0`,
the same with `--pmc WRITE_SIZE`; `tools/pmc_stats.py`.  Units: KB per launch, averaged over the launches of the run.  gfx950
factors measured on this library's own access patterns (`@CAL@`): coalesced 16 / 32-byte reads are tallied at half
their bytes (x 2), scattered 64-byte gathers at their size (x 1), writes at their size (x 1).

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

W('_msm_table.md', """# @RT@ -- one MSM call: bucket pipeline (bits 0: explicit 13-bit windows, no table) against the digit-multiple table (13-bit digits)

`BITS=0,13 python tools/exp/msm_table_bench.py 13 <columns> <kind>`: n = 2^13 points; `full` = random 248-bit scalars,
`small` = 8-bit, `mixed` = a quarter each of 248-bit / 8-bit / 29-bit / 0-1 columns.  `call` = the whole `zkfhe_msm_batch`
(HIP events around it, best of 5), `summing kernel` = `k_msm_accumulate` resp. `k_msm_table` alone, `adds` = mixed additions.
New in round 3: the fold (`call` - `summing kernel` on the table path) runs 512 threads over the even / odd visits of a column.

```
%s```
""" % rd('msm_calls.txt'))

lines = ["# @RT@ -- bench lines (un-profiled, MI355X box, `tools/profile.sh` section (f))", "",
         "`cold` = the same K proofs again with the per-public-key transcript cache off (every proof under a key never seen: BASELINE configs[2]'s 64 keys); `first challenge` = ms from the",
         "start of a proof to its first Fiat-Shamir challenge, in brackets the part of it the proof waited for the HOST after the phase-0 commitment was back from the GPU (the sequential",
         "sponge over the 5 N + 1 public inputs); `announced` = every input announced one proof ahead (`zkfhe_bfv_pk_prehash`: that sponge runs on a host thread while the previous proof is on the GPU).", "",
         "| command | proofs/s | ms per proof | in flight | steady-state pass | cold keys | first challenge ms (host wait) | announced | prefix cache hits / misses | host CPU ms / proof | dominant kernel: avg launch ms, int_alu frac |",
         "|---|---|---|---|---|---|---|---|---|---|---|"]
for fn, cmd in (('bench_driver', "`python bench.py --steps 20 --warmup 5` (the driver's command)"), ('bench_default', '`python bench.py --no-cpu-baseline`'),
                ('bench_blake2b', '`--transcript blake2b`'), ('bench_blake2b_20', '`--transcript blake2b --steps 20 --warmup 5`'),
                ('bench_single_blake2b', '`--steps 8 --streams 1 --transcript blake2b`'), ('bench_single_poseidon', '`--steps 8 --streams 1 --announce off`'),
                ('bench_single_poseidon_announced', '`--steps 16 --streams 1` (announced)'),
                ('bench_k16_blake2b', '`--config k16 --steps 4 --streams 1 --transcript blake2b`'),
                ('bench_k16_2streams', '`--config k16 --steps 8 --warmup 2 --transcript blake2b` (2 in flight)'),
                ('bench_k16_poseidon', '`--config k16 --steps 6 --streams 1 --announce off` (a lone proof, host-bound)'),
                ('bench_k16_poseidon_announced', '`--config k16 --steps 9 --streams 1` (announced)'),
                ('bench_k16_poseidon_s2', '`--config k16 --steps 9 --streams 2` (announced)'), ('bench_k16_poseidon_s3', '`--config k16 --steps 9 --streams 3` (announced)'),
                ('bench_k16_poseidon_s3_plain', '`--config k16 --steps 9 --streams 3 --announce off`'),
                ('bench_k19_blake2b', '`--config k19 --steps 4 --streams 1 --transcript blake2b`'),
                ('bench_k19_poseidon', '`--config k19 --steps 6 --streams 1 --announce off` (a lone proof, host-bound)'),
                ('bench_k19_poseidon_announced', '`--config k19 --steps 9 --streams 1` (announced)'),
                ('bench_k19_poseidon_s2', '`--config k19 --steps 9 --streams 2` (announced)'), ('bench_k19_poseidon_s3', '`--config k19 --steps 9 --streams 3` (announced)'),
                ('bench_k19_poseidon_s3_plain', '`--config k19 --steps 9 --streams 3 --announce off`'),
                ('bench_driver_shared', "`ZKFHE_HASH_MODE=shared` + the driver's command (eight-lane Poseidon service)"),
                ('bench_default_shared', '`ZKFHE_HASH_MODE=shared python bench.py --no-cpu-baseline`'),
                ('bench_driver_nocache', "`ZKFHE_PREFIX_CACHE=0` + the driver's command (no per-public-key transcript cache)")):
    d = jl(fn + '.json')
    if not d:
        lines.append("| %s | (missing) | | | | | | | | | |" % cmd)
        continue
    c = d['config']
    r = d['roofline']
    lat = c.get('per_proof_latency_ms', {})
    lines.append("| %s | %.2f | %.2f | %s | %s | %s | %s | %s | %s / %s | %.1f | %s: %.3f, %.2f |" % (
        cmd, d['value'], d['ms_per_step'], c['concurrent_proofs_per_gpu'], ('%.1f' % c['steady_state_proofs_per_s']) if c['steady_state_proofs_per_s'] else '-',
        ('%.2f' % c['cold_key_proofs_per_s']) if c.get('cold_key_proofs_per_s') else '-',
        ('%.1f (%.1f)' % (lat['first_challenge'], lat['host_wait_for_first_challenge'])) if 'first_challenge' in lat else '-',
        'yes' if c.get('inputs_announced_one_proof_ahead') else 'no', c['host'].get('prefix_cache_hits', '-'), c['host'].get('prefix_cache_misses', '-'),
        c['host_cpu_ms_per_proof'], r['kernel'], r['avg_launch_ms'], r['int_alu']['frac']))
dd = jl('bench_driver.json')
if dd and dd.get('cpu_baseline'):
    cb = dd['cpu_baseline']
    lines += ["", "`cpu_baseline` of the driver's command: %.3f %s on %s threads (`kind: %s`) -- %s.  Seconds per proof by thread count: %s; "
              "phases of the last proof (ms): %s." % (cb['value'], cb['unit'], cb['cores'], cb['kind'], cb['sample'], cb.get('seconds_per_proof_by_threads'),
                                                      cb.get('phase_ms_last_proof'))]
lines += ["", "The full JSON line of the driver's command:", "", "```", rd('bench_driver.json').strip().splitlines()[-1] if os.path.exists(R + 'bench_driver.json') else '', "```", ""]
W('_bench_lines.md', '\n'.join(lines))
W('_microbench.md', "# @RT@ -- micro-benchmarks (`python tools/microbench.py`, MI355X box; the NTT sweep runs out of place: `zkfhe_ntt_batch_to`)\n\n```\n" + rd('microbench.json') + "```\n")

# ---- wave occupancy --------------------------------------------------------------------------------------------------------
W('_wave_occupancy.md', """# @RT@ -- GPU occupancy over the driver's wave of 20 concurrent proofs (2 ms bins)

`rocprofv3 --kernel-trace -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --steady-seconds 0` (`tools/profile.sh`), `tools/busy_bins.py <db> 400 2`:
per bin the fraction of time with at least one kernel running, the average number of kernels in flight and the kernel with the largest
share.  The trace ends with what follows the timed region -- the cold-key pass (the same 20 proofs with the per-key transcript cache off:
a second wave of ~100 ms) and the two profiled proofs (one in flight, ~60 ms); the timed wave is the ~100 ms block before those: a head of 14-18 ms in which the proofs' early work (phase-0 commitment, gadgets, early phase-1 commitment) shares the
chip while every proof hashes its 5 121 public inputs, 50-75 ms with 12-16 kernels in flight -- the proofs move through their rounds
together, so this block is one phase after the other (the 2^13 tiles of all twenty proofs alone fill ~35 ms of it) -- and 14-25 ms in
which the kernels in flight fall from ten to one: the proofs' serial ends (evaluations, 758 Poseidon permutations on the host, two
one-column commitments).

```
%s```
""" % rd('a_driver_bins.txt'))

# ---- roofline table: one recomputable line per kernel >= 5 % of a configuration ----------------------------------------------
GATHER = ('k_msm_table<false>', 'k_msm_table<true>', 'k_msm_accumulate')   # reads dominated by scattered 64-byte table points


def last_proof(f):
    out = {}
    for l in rd(f).splitlines():
        t = l.split()
        if len(t) >= 6 and t[1] == 'calls':
            out[t[0]] = (int(t[2]), float(t[3]), float(t[5].rstrip('%')))
    return out


def pmc(f):
    out = {}
    for l in rd(f).splitlines():
        if 'avg=' in l:
            out[l.split()[0]] = (float(l.split('avg=')[1]), int(l.split('launches=')[1].split()[0]))
    return out


def cols_of(d):
    c = d['config'].get('columns') if d else None
    if not c:
        return None
    adv = c['gate0'] + c['gate1'] + c['lookup'] + c['rlc']
    perm = adv + 2                      # advice + constants + instance
    chunks = -(-perm // 2)              # degree 4: two columns per grand product
    return dict(c, advice=adv, perm=perm, chunks=chunks, all=adv + 3 * c['lookup'] + chunks + 1)


rows = ["# @RT@ -- roofline table: every kernel that takes >= 5 % of a lone proof's kernel time in some configuration", "",
        "Columns: calls and total duration in ONE proof (rocprofv3 kernel trace, `tools/last_proof_stats.py`: the files `@RT@_b_single_proof.md`, `@RT@_k16_kernel_stats.md`,",
        "`@RT@_k19_kernel_stats.md`), algorithmic bytes of those calls (formula in the last column, SURVEY.md 8(d)), algorithmic GB/s = bytes / duration, fraction of the 8 TB/s",
        "HBM peak, and the bytes the memory-side counters saw per proof: (factor x FETCH_SIZE + WRITE_SIZE) x 1 KiB summed over the kernel's launches (separate `--pmc` passes,",
        "`tools/pmc_stats.py`), factor = 2 for the streaming kernels and 1 for the MSM kernels whose reads are scattered 64-byte table points -- measured, `@CAL@` --",
        "and their ratio to the algorithmic bytes (for k_msm_accumulate the counter includes Infinity-Cache hits on its window table: L2-miss traffic, not HBM traffic).",
        "All of these kernels do 256-bit modular integer arithmetic; none is bound by HBM (DESIGN.md section 3): the integer rate against the bare-product-loop rate measured by",
        "`tools/exp/mad_rate.hip` / `tools/microbench.py` (168-182 G products/s, NOT a hardware bound) is in `@RT@_bench_lines.md` (`int_alu`).", "",
        "| config | kernel | calls | ms | % of kernel time | algorithmic MB | GB/s | frac of 8 TB/s | counter MB | counter / algorithmic | algorithmic bytes |",
        "|---|---|---|---|---|---|---|---|---|---|---|"]
for cfg, stats_f, bench_f, pre, n in (('k13', 'b_single_last_proof.txt', 'bench_single_blake2b.json', 'pmc_', 8192), ('k16', 'c_k16_last_proof.txt', 'bench_k16_blake2b.json', 'pmc_k16_', 65536),
                                      ('k19', 'd_k19_last_proof.txt', 'bench_k19_blake2b.json', 'pmc_k19_', 524288)):
    lp_, cc = last_proof(stats_f), cols_of(jl(bench_f))
    fe, wr = pmc(pre + 'FETCH_SIZE.txt'), pmc(pre + 'WRITE_SIZE.txt')
    if not lp_ or not cc:
        rows.append("| %s | (missing: %s / %s) | | | | | | | | | |" % (cfg, stats_f, bench_f))
        continue
    ext_cols = cc['all']
    wide = (cc['advice'] - cc['gate0'] + 2 * cc['lookup']) + (cc['chunks'] + cc['lookup'])   # the two wide commitment calls
    small = cc['gate0'] + 1 + 3 + 1 + 1                                                         # phase 0, random, h pieces, two openings
    alg = {
        'k_msm_table<false>': (96.0 * n * wide, "96 n x %d columns (the two wide commitment calls)" % wide),
        'k_msm_accumulate': (96.0 * n * (wide + small), "96 n x %d columns (every commitment of the proof)" % (wide + small)),
        'k_msm_table<true>': (96.0 * n * small, "96 n x %d columns (the calls of 1-3 columns)" % small),
        'k_ntt13': (64.0 * n * (ext_cols * 4 + 3 + 2), "64 n per transform: %d columns x (1 inverse + 3 coset rows) + 5 single columns" % ext_cols),
        'k_dif_fused<3>': (64.0 * n * (ext_cols * 4 + 5) * max(0, (n.bit_length() - 1 - 13 + 2) // 3), "64 n per pass and transform, %d passes of three stages" % max(0, (n.bit_length() - 1 - 13 + 2) // 3)),
        'zkp::k_quotient_partials': (32.0 * 3 * n * (cc['advice'] + (cc['gate0'] + cc['gate1'] + cc['rlc'] + 2) + cc['perm'] + cc['chunks'] + 3 * cc['lookup'] + 4 + 1),
                                     "32 B x 3 n points x (advice + fixed + sigma + products + lookup polynomials + l_0 / l_last / l_active / X + 1 output)"),
        'k_fr_batch_invert': (64.0 * n * (cc['chunks'] + cc['lookup'] + 6 + 8 + 1), "64 B per element: grand-product denominators, barycentric weights, SHPLONK denominators"),
        'k_msm_scatter': ((32.0 + 4.0 * 16) * n * (wide + small), "(32 B scalar + 4 B x ~16 entries) per scalar"),
        'k_msm_cscatter': ((32.0 + 4.0 * 16) * n * (wide + small), "(32 B scalar + 4 B x ~16 staged entries) per scalar"),
        'k_msm_fine': ((3 * 4.0 * 16) * n * (wide + small), "4 B x ~16 entries per scalar: two reads of the staged segment, one write of the sorted entries"),
        'k_dif_lds<3>': (64.0 * n * (ext_cols * 4 + 5), "64 n per transform, six stages in one pass (+ 32 n of coset factors on the forward rows)"),
        'k_dif8_two': (64.0 * n * (ext_cols * 4 + 5), "64 n per transform: the six stages above the tile as one four-step pass (the 32 n table of a pass is shared by all columns: not counted)"),
        'k_dif8_one': (64.0 * n * (ext_cols * 4 + 5), "64 n per transform: the three stages above the tile as one four-step pass"),
        'zkp::k_quotient_blocks': (32.0 * 3 * n * (cc['advice'] + (cc['gate0'] + cc['gate1'] + cc['rlc'] + 2) + cc['perm'] + cc['chunks'] + 3 * cc['lookup'] + 4 + 1), "as k_quotient_partials"),
        'k_dif_lds<2>': (64.0 * n * (ext_cols * 4 + 5), "64 n per transform, five stages in one pass"),
        'k_dif_lds<1>': (64.0 * n * (ext_cols * 4 + 5), "64 n per transform, four stages in one pass"),
        'k_msm_table_fold': (None, "latency-bound: sums the partial lists (128 B per partial), one workgroup per column"),
        'zkp::k_prefix_product': (64.0 * n * (cc['chunks'] + cc['lookup']), "64 B per element of the grand products"),
    }
    for name, (calls, ms, pct) in sorted(lp_.items(), key=lambda kv: -kv[1][1]):
        if pct < 5.0:
            continue
        a_bytes, formula = alg.get(name, (None, "--"))
        key = name[-40:]
        cnt = None
        if key in fe and key in wr:
            # counters are per-launch averages over the run (warm-up + timed proofs): launches per proof = this table's calls
            rf = 1.0 if name in GATHER else 2.0     # read factor of the kernel's dominant pattern (@CAL@)
            cnt = (rf * fe[key][0] + wr[key][0]) * 1024 * calls
        rows.append("| %s | %s | %d | %.3f | %.1f | %s | %s | %s | %s | %s | %s |" % (
            cfg, name, calls, ms, pct, ('%.1f' % (a_bytes / 1e6)) if a_bytes else '-', ('%.0f' % (a_bytes / ms / 1e6)) if a_bytes else '-',
            ('%.3f' % (a_bytes / ms / 1e6 / 8000)) if a_bytes else '-', ('%.1f' % (cnt / 1e6)) if cnt else '-', ('%.1f' % (cnt / a_bytes)) if (cnt and a_bytes) else '-', formula))
rows += ["", "Micro-benchmarks at the sizes SURVEY.md 8(d) lists (NTT 256 columns at 2^13 ... 2^19, 64 at 2^21; MSM at 2^16 / 2^19, uniform and witness-like scalars): `@RT@_microbench.md`.", ""]
W('_roofline.md', '\n'.join(rows))
print(open(P + RT + '_bench_lines.md').read()[:3000])
