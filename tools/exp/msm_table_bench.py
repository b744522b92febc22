"""Timing of one MSM call against a basis with a digit-multiple table (k_msm_table) and against the bucket pipeline.
usage: python tools/exp/msm_table_bench.py [log_n] [n_cols] [kind]   (kind: full | small | mixed)"""
import importlib.util, os, sys, time
import numpy as np
spec = importlib.util.spec_from_file_location("zk_fhe_amd", os.path.join(os.path.dirname(__file__), "../../zk-fhe_amd/__init__.py"))
zk = importlib.util.module_from_spec(spec); sys.modules["zk_fhe_amd"] = zk; spec.loader.exec_module(zk)
Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
def limbs(v): return [(v >> (64 * i)) & (2**64 - 1) for i in range(4)]
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 13
n_cols = int(sys.argv[2]) if len(sys.argv) > 2 else 64
kind = sys.argv[3] if len(sys.argv) > 3 else "full"
n = 1 << log_n
ctx = zk.Context(0)
rng = np.random.default_rng(1)
gen = np.array([limbs((1 << 256) % Q) + limbs((2 << 256) % Q)] * n, dtype=np.uint64)
ks = np.array([limbs(int.from_bytes(rng.bytes(31), "little") * (1 << 256) % R) for _ in range(n)], dtype=np.uint64)
pts = ctx.g1_mul(gen, ks)
def col(j):
    if kind == "full" or (kind == "mixed" and j % 4 == 0):
        return [int.from_bytes(rng.bytes(31), "little") for _ in range(n)]
    if kind == "small" or j % 4 == 1:
        return [int(x) for x in rng.integers(0, 256, n)]
    if j % 4 == 2:
        return [int(x) for x in rng.integers(0, 1 << 29, n)]
    return [int(x) for x in rng.integers(0, 2, n)]
S = np.array([[limbs(v * (1 << 256) % R) for v in col(j)] for j in range(n_cols)], dtype=np.uint64)
Sd = ctx.to_device(S)
out = ctx.alloc(64 * n_cols)
for bits in os.environ.get("BITS", "0,8,10,12").split(","):
    bits = int(bits)
    if bits:
        os.environ["ZKFHE_TABLE_BITS"] = str(bits)
        t0 = time.time(); B = zk.Basis(ctx, pts); tb = time.time() - t0
    else:
        t0 = time.time(); B = zk.Basis(ctx, pts, 13 if log_n >= 13 else 10); tb = time.time() - t0
    ctx.prof_enable(True)
    for _ in range(3): ctx.msm_dev(B, Sd, n_cols, out)
    ctx.sync()
    p0, p2 = ctx.prof_read(0), ctx.prof_read(2)
    ctx.prof_enable(False)
    ts = []
    for _ in range(5):
        ctx.timer_start(); ctx.msm_dev(B, Sd, n_cols, out); ts.append(ctx.timer_stop_ms())
    adds = (p0.get("ops", 0) + p2.get("ops", 0)) / 3
    kms = (p0["total_ms"] + p2["total_ms"]) / 3
    print("bits %2d  basis %.2fs  call %.3f ms  summing kernel %.3f ms  adds %.2fM" % (bits, tb, min(ts), kms, adds / 1e6))
    B.destroy()
