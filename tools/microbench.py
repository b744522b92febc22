#!/usr/bin/env python3
"""Micro-benchmarks behind the roofline numbers in DESIGN.md: Fr modmul/s (ALU bound of every kernel),
element-wise Fr mul GB/s (HBM-bound reference point), NTT and MSM sweeps.  Run on the GPU box."""
import json
import sys
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch  # noqa: F401
    import zk_fhe_amd as zk
    ctx = zk.Context(0)
    rng = np.random.default_rng(1)
    res = {}
    # modmul throughput: n independent chains of `iters` squarings
    n, iters = 1 << 20, 512
    raw = np.frombuffer(rng.bytes(32 * n), dtype=np.uint64).reshape(n, 4).copy()
    raw[:, 3] &= np.uint64(0x0FFFFFFFFFFFFFFF)
    d = ctx.to_device(raw)
    o = ctx.alloc(n * 32)
    for _ in range(2):
        ctx._check(ctx.lib.zkfhe_fr_sqr_chain(ctx.h, d.at(0), o.at(0), n, iters))
    ts = []
    for _ in range(5):
        ctx.timer_start()
        ctx._check(ctx.lib.zkfhe_fr_sqr_chain(ctx.h, d.at(0), o.at(0), n, iters))
        ts.append(ctx.timer_stop_ms())
    ms = float(np.median(ts))
    res["modmul_per_s"] = n * iters / (ms * 1e-3)
    # the radix-2^29 product of the MSM kernels (fq29.hip.hpp), same loop
    for _ in range(2):
        ctx._check(ctx.lib.zkfhe_fq29_sqr_chain(ctx.h, d.at(0), o.at(0), n, iters))
    ts = []
    for _ in range(5):
        ctx.timer_start()
        ctx._check(ctx.lib.zkfhe_fq29_sqr_chain(ctx.h, d.at(0), o.at(0), n, iters))
        ts.append(ctx.timer_stop_ms())
    res["modmul29_per_s"] = n * iters / (float(np.median(ts)) * 1e-3)
    ctx._check(ctx.lib.zkfhe_fq29_sqr_chain(ctx.h, d.at(0), o.at(0), 64, 20000))
    ctx.timer_start()
    ctx._check(ctx.lib.zkfhe_fq29_sqr_chain(ctx.h, d.at(0), o.at(0), 64, 20000))
    res["modmul29_latency_us_single_wave"] = ctx.timer_stop_ms() * 1e3 / 20000
    # single-wave dependent-chain latency of one modmul (64 threads, long chain)
    lat_iters = 20000
    ctx._check(ctx.lib.zkfhe_fr_sqr_chain(ctx.h, d.at(0), o.at(0), 64, lat_iters))
    ctx.timer_start()
    ctx._check(ctx.lib.zkfhe_fr_sqr_chain(ctx.h, d.at(0), o.at(0), 64, lat_iters))
    res["modmul_latency_us_single_wave"] = ctx.timer_stop_ms() * 1e3 / lat_iters
    # element-wise mul: 96 B per element
    for _ in range(2):
        ctx.fr_binop_dev("mul", d, d, o, n)
    ts = []
    for _ in range(10):
        ctx.timer_start()
        ctx.fr_binop_dev("mul", d, d, o, n)
        ts.append(ctx.timer_stop_ms())
    ms = float(np.median(ts))
    res["fr_mul_elementwise_GBs"] = 64.0 * n / (ms * 1e-3) / 1e9  # a==b: 32 B read + 32 B write
    ts = []
    for _ in range(10):
        ctx.timer_start()
        ctx.fr_binop_dev("add", d, d, o, n)
        ts.append(ctx.timer_stop_ms())
    res["fr_add_elementwise_GBs"] = 64.0 * n / (float(np.median(ts)) * 1e-3) / 1e9
    # NTT sweep, 256 columns
    res["ntt"] = {}
    # 256 columns at 2^13 / 2^15 / 2^16 / 2^19 (4 GiB in + 4 GiB out), 2^21 as 64 columns (the same 4 GiB)
    big = "--big" in sys.argv
    for log_n in (10, 12, 13, 15, 16) + ((19, 21) if big else ()):
        cols = 256 if log_n <= 19 else 64
        buf = ctx.alloc(cols * (32 << log_n))
        out = ctx.alloc(cols * (32 << log_n))   # out of place (zkfhe_ntt_batch_to): how the prover runs every column transform
        ctx._check(ctx.lib.zkfhe_memset_dev(ctx.h, buf.at(0), 1, buf.nbytes))
        for _ in range(2):
            ctx.ntt_to_dev(buf, out, cols, log_n, False)
        ts = []
        for _ in range(10):
            ctx.timer_start()
            ctx.ntt_to_dev(buf, out, cols, log_n, False)
            ts.append(ctx.timer_stop_ms())
        ms = float(np.median(ts))
        res["ntt"]["2^%d x %d" % (log_n, cols)] = {"ms": ms, "GBs_algorithmic": 64.0 * cols * (1 << log_n) / (ms * 1e-3) / 1e9,
                                                  "modmul_per_s": cols * (1 << log_n) / 2 * log_n / (ms * 1e-3)}
        buf.free()
        out.free()
    # MSM sweep: uniform scalars, n = 8192
    nn = 8192
    # bases: k_i * G for random k_i, made on the device (zkfhe_g1_mul); G = (1, 2), Montgomery limbs
    Q_MOD = 21888242871839275222246405745257275088696311157297823662689037894645226208583
    R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617

    def limbs(v):
        return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]
    gen = np.array([limbs((1 << 256) % Q_MOD) + limbs((2 << 256) % Q_MOD)] * nn, dtype=np.uint64)
    ks = np.array([limbs(((int(x) % R_MOD) << 256) % R_MOD) for x in rng.integers(1, 1 << 62, nn)], dtype=np.uint64)
    bases = ctx.g1_mul(gen, ks)
    res["msm"] = {}
    for c in (10, 11, 12, 13, 14):
        B = zk.Basis(ctx, bases, c)
        for cols in (1, 16, 197):
            raw = np.frombuffer(rng.bytes(32 * nn * cols), dtype=np.uint64).reshape(-1, 4).copy()
            raw[:, 3] &= np.uint64(0x0FFFFFFFFFFFFFFF)
            ds = ctx.to_device(raw)
            do = ctx.alloc(cols * 64)
            for _ in range(2):
                ctx.msm_dev(B, ds, cols, do)
            ts = []
            for _ in range(5):
                ctx.timer_start()
                ctx.msm_dev(B, ds, cols, do)
                ts.append(ctx.timer_stop_ms())
            ms = float(np.median(ts))
            res["msm"]["c=%d cols=%d" % (c, cols)] = {"ms": ms, "us_per_msm": ms * 1e3 / cols}
            ds.free(), do.free()
        B.destroy()
    if big:
        # MSM at n = 2^16 and 2^19 (the bucket pipeline; 2^13 is above), two scalar mixes per size:
        #   uniform = 254-bit scalars (worst case);  witness = the mix of a gate column of the BFV circuit -- half the cells 8-bit
        #   limbs of the range checks, a quarter 30-bit coefficients, 15 % 70-bit intermediate sums, 10 % full-width (inverses)
        res["msm_big"] = {}
        for log_n, cols in ((16, 32), (19, 8)):
            nb = 1 << log_n
            gen_b = np.array([limbs((1 << 256) % Q_MOD) + limbs((2 << 256) % Q_MOD)] * nb, dtype=np.uint64)
            kk = rng.integers(1, 1 << 62, nb)
            ks_b = np.array([limbs(((int(x) % R_MOD) << 256) % R_MOD) for x in kk], dtype=np.uint64)
            bases_b = ctx.g1_mul(gen_b, ks_b)
            Bb = zk.Basis(ctx, bases_b)
            for mix in ("uniform", "witness"):
                raw = np.frombuffer(rng.bytes(32 * nb * cols), dtype=np.uint64).reshape(-1, 4).copy()
                raw[:, 3] &= np.uint64(0x0FFFFFFFFFFFFFFF)
                if mix == "witness":
                    # canonical small values, then to Montgomery form on the device (the prover's columns are Montgomery values)
                    u = rng.random(nb * cols)
                    raw[u < 0.9, 1:] = 0
                    raw[u < 0.9, 1] = raw[u < 0.9, 0] & np.uint64(0x3F)          # 70 bits
                    raw[u < 0.75, 1] = 0
                    raw[u < 0.75, 0] &= np.uint64((1 << 30) - 1)
                    raw[u < 0.5, 0] &= np.uint64(0xFF)
                ds = ctx.to_device(raw)
                if mix == "witness":
                    ctx._check(ctx.lib.zkfhe_fr_to_mont(ctx.h, ds.at(0), ds.at(0), nb * cols))
                do = ctx.alloc(cols * 64)
                for _ in range(2):
                    ctx.msm_dev(Bb, ds, cols, do)
                ts = []
                for _ in range(5):
                    ctx.timer_start()
                    ctx.msm_dev(Bb, ds, cols, do)
                    ts.append(ctx.timer_stop_ms())
                ms = float(np.median(ts))
                res["msm_big"]["2^%d x %d %s" % (log_n, cols, mix)] = {"ms": ms, "ms_per_msm": ms / cols, "GBs_algorithmic": 96.0 * nb * cols / (ms * 1e-3) / 1e9}
                ds.free(), do.free()
            Bb.destroy()
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
