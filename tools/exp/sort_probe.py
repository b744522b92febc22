#!/usr/bin/env python3
"""Timing probe of the bucket pipeline's sort on a long basis: n = 2^LOG points, COLS full-width columns, three calls.
usage (GPU box, under rocprofv3 --kernel-trace --stats): tools/exp/sort_probe.py [log_n] [cols]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import zk_fhe_amd as zk

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 19
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 32
Q_MOD = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def limbs(v):
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


ctx = zk.Context(0)
rng = np.random.default_rng(5)
nb = 1 << log_n
gen = np.array([limbs((1 << 256) % Q_MOD) + limbs((2 << 256) % Q_MOD)] * nb, dtype=np.uint64)
ks = np.array([limbs(((int(x) % R_MOD) << 256) % R_MOD) for x in rng.integers(1, 1 << 62, nb)], dtype=np.uint64)
bases = ctx.g1_mul(gen, ks)
os.environ.setdefault("ZKFHE_TABLE_GB", "1")
B = zk.Basis(ctx, bases, 16 if log_n >= 18 else 14)
raw = np.frombuffer(rng.bytes(32 * nb * cols), dtype=np.uint64).reshape(-1, 4).copy()
raw[:, 3] &= np.uint64(0x0FFFFFFFFFFFFFFF)
ds = ctx.to_device(raw)
do = ctx.alloc(cols * 64)
ts = []
for _ in range(3):
    ctx.timer_start()
    ctx.msm_dev(B, ds, cols, do)
    ts.append(ctx.timer_stop_ms())
print("msm 2^%d x %d: %s ms" % (log_n, cols, ["%.2f" % t for t in ts]))
