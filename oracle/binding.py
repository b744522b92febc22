"""ctypes binding of the C oracle (oracle/liboracle.so). TEST INFRASTRUCTURE ONLY -- see bn254_ref.h.

Arrays are numpy uint64 with trailing dimension 4 (Fr/Fq, Montgomery form) or 8 (G1 affine x||y).
"""
import ctypes
import os
import subprocess

import numpy as np

from . import pyref

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("oracle.c", "bn254_ref.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = ctypes.CDLL(so)
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def ints_to_arr(xs):
    """canonical python ints -> (n,4) uint64 array (canonical, NOT Montgomery)"""
    buf = b"".join(int(x).to_bytes(32, "little") for x in xs)
    return np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).copy()


def arr_to_ints(a):
    b = np.ascontiguousarray(a, dtype=np.uint64).tobytes()
    return [int.from_bytes(b[i:i + 32], "little") for i in range(0, len(b), 32)]


def to_mont(a, which=0):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    out = np.empty_like(a)
    lib().orc_fe_to_mont(which, _p(a), _p(out), ctypes.c_size_t(a.size // 4))
    return out


def from_mont(a, which=0):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    out = np.empty_like(a)
    lib().orc_fe_from_mont(which, _p(a), _p(out), ctypes.c_size_t(a.size // 4))
    return out


def ints_to_mont(xs, which=0):
    return to_mont(ints_to_arr(xs), which)


def mont_to_ints(a, which=0):
    return arr_to_ints(from_mont(a, which))


def fe_binop(name, a, b, which=0):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    out = np.empty_like(a)
    getattr(lib(), "orc_fe_" + name)(which, _p(a), _p(b), _p(out), ctypes.c_size_t(a.size // 4))
    return out


def fe_inv(a, which=0):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    out = np.empty_like(a)
    lib().orc_fe_inv(which, _p(a), _p(out), ctypes.c_size_t(a.size // 4))
    return out


def fr_batch_inv(a):
    a = np.ascontiguousarray(a, dtype=np.uint64).copy()
    lib().orc_fr_batch_inv(_p(a), ctypes.c_size_t(a.size // 4))
    return a


def fr_horner(poly, x):
    poly = np.ascontiguousarray(poly, dtype=np.uint64)
    x = np.ascontiguousarray(x, dtype=np.uint64)
    out = np.empty(4, dtype=np.uint64)
    lib().orc_fr_horner(_p(poly), ctypes.c_size_t(poly.size // 4), _p(x), _p(out))
    return out


def ntt(cols, log_n, inverse=False):
    """cols: (n_cols, 2^log_n, 4) Montgomery Fr. Returns a transformed copy."""
    a = np.ascontiguousarray(cols, dtype=np.uint64).copy()
    n = 1 << log_n
    lib().orc_ntt(_p(a), ctypes.c_size_t(a.size // 4 // n), log_n, int(bool(inverse)))
    return a


def fft(vec, log_n, omega):
    a = np.ascontiguousarray(vec, dtype=np.uint64).copy()
    omega = np.ascontiguousarray(omega, dtype=np.uint64)
    lib().orc_fft(_p(a), log_n, _p(omega))
    return a


def coset_ntt(vec, log_ext, g, inverse=False):
    a = np.ascontiguousarray(vec, dtype=np.uint64)
    g = np.ascontiguousarray(g, dtype=np.uint64)
    out = np.empty(((1 << log_ext), 4), dtype=np.uint64)
    lib().orc_coset_ntt(_p(a), ctypes.c_size_t(a.size // 4), _p(out), log_ext, _p(g), int(bool(inverse)))
    return out


def root_of_unity(log_n):
    out = np.empty(4, dtype=np.uint64)
    lib().orc_root_of_unity(log_n, _p(out))
    return out


def points_to_arr(pts):
    """list of affine big-int points (None = identity) -> (n,8) uint64 Montgomery Fq"""
    xs = []
    for p in pts:
        x, y = pyref.g1_affine_to_xy(p)
        xs += [x, y]
    return ints_to_mont(xs, 1).reshape(-1, 8)


def arr_to_points(a):
    v = mont_to_ints(np.ascontiguousarray(a).reshape(-1, 4), 1)
    out = []
    for i in range(0, len(v), 2):
        out.append(None if (v[i] == 0 and v[i + 1] == 0) else (v[i], v[i + 1]))
    return out


def g1_add(a, b):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    out = np.empty_like(a)
    lib().orc_g1_add(_p(a), _p(b), _p(out), ctypes.c_size_t(a.size // 8))
    return out


def g1_mul(p, k):
    p = np.ascontiguousarray(p, dtype=np.uint64)
    k = np.ascontiguousarray(k, dtype=np.uint64)
    out = np.empty_like(p)
    lib().orc_g1_mul(_p(p), _p(k), _p(out), ctypes.c_size_t(p.size // 8))
    return out


def g1_on_curve(p):
    p = np.ascontiguousarray(p, dtype=np.uint64)
    return bool(lib().orc_g1_on_curve(_p(p)))


def msm(scalars, bases):
    """scalars (n_cols, n, 4) Montgomery Fr, bases (n, 8) -> (n_cols, 8)"""
    s = np.ascontiguousarray(scalars, dtype=np.uint64)
    b = np.ascontiguousarray(bases, dtype=np.uint64)
    n = b.size // 8
    n_cols = s.size // 4 // n
    out = np.empty((n_cols, 8), dtype=np.uint64)
    lib().orc_msm(_p(s), ctypes.c_size_t(n_cols), _p(b), ctypes.c_size_t(n), _p(out))
    return out


def msm_naive(scalars, bases):
    s = np.ascontiguousarray(scalars, dtype=np.uint64)
    b = np.ascontiguousarray(bases, dtype=np.uint64)
    out = np.empty(8, dtype=np.uint64)
    lib().orc_msm_naive(_p(s), _p(b), ctypes.c_size_t(b.size // 8), _p(out))
    return out


def g1_powers(start, step, n):
    """n deterministic affine points (start*step^i)*G -- synthetic basis for tests/bench."""
    out = np.empty((n, 8), dtype=np.uint64)
    lib().orc_g1_powers(_p(np.ascontiguousarray(start)), _p(np.ascontiguousarray(step)), _p(out), ctypes.c_size_t(n))
    return out


def num_threads():
    return int(lib().orc_num_threads())


def fr_scale(a, s):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    out = np.empty_like(a)
    lib().orc_fr_scale(_p(a), _p(np.ascontiguousarray(s, dtype=np.uint64)), _p(out), ctypes.c_size_t(a.size // 4))
    return out


def fr_add_scalar(a, s):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    out = np.empty_like(a)
    lib().orc_fr_add_scalar(_p(a), _p(np.ascontiguousarray(s, dtype=np.uint64)), _p(out), ctypes.c_size_t(a.size // 4))
    return out


def fr_axpy(acc, x, s):
    """acc += s * x (in place on acc)"""
    assert acc.flags["C_CONTIGUOUS"] and acc.dtype == np.uint64
    x = np.ascontiguousarray(x, dtype=np.uint64)
    lib().orc_fr_axpy(_p(acc), _p(x), _p(np.ascontiguousarray(s, dtype=np.uint64)), ctypes.c_size_t(x.size // 4))
    return acc


def fr_mul(a, b):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    out = np.empty_like(a)
    lib().orc_fr_mul_vec(_p(a), _p(b), _p(out), ctypes.c_size_t(a.size // 4))
    return out


def fr_add(a, b):
    return fe_binop("add", a, b)


def fr_sub(a, b):
    return fe_binop("sub", a, b)


def fr_prefix_prod(a, init):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    out = np.empty((a.shape[0] + 1, 4), dtype=np.uint64)
    lib().orc_fr_prefix_prod(_p(a), _p(np.ascontiguousarray(init, dtype=np.uint64)), _p(out), ctypes.c_size_t(a.shape[0]))
    return out


def fr_powers(start, base, n):
    out = np.empty((n, 4), dtype=np.uint64)
    lib().orc_fr_powers(_p(np.ascontiguousarray(start, dtype=np.uint64)), _p(np.ascontiguousarray(base, dtype=np.uint64)), _p(out), ctypes.c_size_t(n))
    return out


def fr_div_linear(p, root):
    p = np.ascontiguousarray(p, dtype=np.uint64).reshape(-1, 4)
    out = np.empty_like(p)
    lib().orc_fr_div_linear(_p(p), ctypes.c_size_t(p.shape[0]), _p(np.ascontiguousarray(root, dtype=np.uint64)), _p(out))
    return out


def fr_lincomb(cols, scalars):
    cols = np.ascontiguousarray(cols, dtype=np.uint64)
    n_cols, n = cols.shape[0], cols.shape[1]
    s = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(n_cols, 4)
    out = np.empty((n, 4), dtype=np.uint64)
    lib().orc_fr_lincomb(_p(cols), ctypes.c_size_t(n_cols), ctypes.c_size_t(n), _p(s), _p(out))
    return out


def fr_horner_batch(cols, xs):
    cols = np.ascontiguousarray(cols, dtype=np.uint64)
    n_cols, n = cols.shape[0], cols.shape[1]
    xs = np.ascontiguousarray(xs, dtype=np.uint64).reshape(n_cols, 4)
    out = np.empty((n_cols, 4), dtype=np.uint64)
    lib().orc_fr_horner_batch(_p(cols), ctypes.c_size_t(n_cols), ctypes.c_size_t(n), _p(xs), _p(out))
    return out


def coset_ntt_cols(cols, log_ext, g):
    cols = np.ascontiguousarray(cols, dtype=np.uint64)
    n_cols, n_in = cols.shape[0], cols.shape[1]
    out = np.empty((n_cols, 1 << log_ext, 4), dtype=np.uint64)
    lib().orc_coset_ntt_cols(_p(cols), ctypes.c_size_t(n_cols), ctypes.c_size_t(n_in), _p(out), log_ext, _p(np.ascontiguousarray(g, dtype=np.uint64)))
    return out
