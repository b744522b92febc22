#!/usr/bin/env python3
"""Per-workgroup phase clocks of the 2^13 tile (GPU box; needs the instrumented build libzkfhe_hip_clk.so, made by tools/exp/r6_v.sh's
recipe: wall_clock64 stamps at entry / after the load + radix-2 stage / after the first radix-8 pass and its workgroup exchange / after the
three in-wave passes / after the output exchange / after the stores are issued / after they drained, and the HW_ID / XCC_ID registers)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch  # noqa: F401
    import zk_fhe_amd as zk
    from oracle import binding as orc
    ctx = zk.Context(0)
    lib = ctx.lib
    rng = np.random.default_rng(1)
    n_cols, log_n = 256, 13
    n = 1 << log_n
    raw = np.frombuffer(rng.bytes(32 * n * n_cols), dtype=np.uint64).reshape(n_cols, n, 4).copy()
    raw[..., 3] &= np.uint64(0x0FFFFFFFFFFFFFFF)
    d = ctx.to_device(raw)
    g = orc.ints_to_mont([7])[0]
    o = ctx.alloc(n_cols * n * 4 * 32)
    for which in ("plain", "coset"):
        for _ in range(3):
            if which == "plain":
                ctx.ntt_to_dev(d, o, n_cols, log_n, inverse=False)
            else:
                ctx.coset_ntt_dev(d, o, n_cols, log_n, 2, g)
        ctx.sync()
        nwg = 512 if which == "plain" else 2048
        buf = np.zeros((16384, 8), dtype=np.uint64)
        rc = lib.zkfhe_debug_ntt13_clk(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
        assert rc == 0, rc
        b = buf[:nwg].astype(np.int64)
        t0 = b[:, 0].min()
        rel = (b[:, :7] - t0) * 0.01   # us (100 MHz)
        names = ["load + radix 2", "radix-8 pass + workgroup exchange", "three in-wave passes", "output exchange", "stores issued", "stores drained"]
        print("== %s: %d workgroups, launch span %.1f us" % (which, nwg, rel[:, 6].max()))
        dur = np.diff(rel, axis=1)
        for i, nm in enumerate(names):
            print("  %-36s mean %6.2f us  p10 %6.2f  p90 %6.2f" % (nm, dur[:, i].mean(), np.percentile(dur[:, i], 10), np.percentile(dur[:, i], 90)))
        life = rel[:, 6] - rel[:, 0]
        print("  workgroup lifetime                   mean %6.2f us  p10 %6.2f  p90 %6.2f" % (life.mean(), np.percentile(life, 10), np.percentile(life, 90)))
        # per CU: gaps between consecutive workgroups
        hw = b[:, 7]
        cu_key = ((hw >> 32) & 0xF) * 4096 + (hw & 0xFFFF & ~0x3F)   # xcc id | se / sh / cu bits (wave and simd bits masked)
        gaps, per_cu = [], []
        for k in np.unique(cu_key):
            idx = np.where(cu_key == k)[0]
            idx = idx[np.argsort(rel[idx, 0])]
            per_cu.append(len(idx))
            for a, c in zip(idx, idx[1:]):
                gaps.append(rel[c, 0] - rel[a, 6])
        gaps = np.array(gaps) if gaps else np.zeros(1)
        print("  distinct CU keys %d, workgroups per CU %s..%s; gap between a workgroup's end and the next one's entry on its CU: mean %.2f us, p10 %.2f, p90 %.2f"
              % (len(per_cu), min(per_cu), max(per_cu), gaps.mean(), np.percentile(gaps, 10), np.percentile(gaps, 90)))
        first = np.sort(rel[:, 0])[:256]
        print("  entry of the first 256 workgroups: %.2f .. %.2f us after the first" % (first.min(), first.max()))


if __name__ == "__main__":
    main()
