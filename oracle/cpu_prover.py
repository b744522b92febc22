"""ctypes binding of oracle/libcpuprover.so (oracle/cpu_prover.cpp). TEST INFRASTRUCTURE ONLY.

The native multithreaded CPU prover behind bench.py's `cpu_baseline` leg: one C call per proof, no Python in the
timed path.  The proving key and the SRS come from oracle/halo2_ref.py (keygen is not timed); the proof must equal
`halo2_ref.prove`'s bytes for the same seed (tests/test_cpu_prover.py).
"""
import ctypes
import hashlib
import os
import subprocess

import numpy as np

from . import pyref

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
PHASES = ("phase0_witness_commit", "phase1_witness", "advice_commit", "lookup_permute_commit", "grand_products_commit",
          "ntt", "quotient", "evaluations", "multiopen")


class _Cfg(ctypes.Structure):
    _fields_ = [("k", ctypes.c_uint32), ("n_gate0", ctypes.c_uint32), ("n_gate1", ctypes.c_uint32), ("n_lookup", ctypes.c_uint32),
                ("n_rlc", ctypes.c_uint32), ("unusable_rows", ctypes.c_uint32), ("lookup_bits", ctypes.c_uint32), ("transcript", ctypes.c_uint32),
                ("bp_gate0", ctypes.POINTER(ctypes.c_uint32)), ("n_bp_gate0", ctypes.c_uint32),
                ("bp_gate1", ctypes.POINTER(ctypes.c_uint32)), ("n_bp_gate1", ctypes.c_uint32),
                ("bp_rlc", ctypes.POINTER(ctypes.c_uint32)), ("n_bp_rlc", ctypes.c_uint32),
                ("bfv_n", ctypes.c_uint64), ("bfv_q", ctypes.c_uint64), ("bfv_t", ctypes.c_uint64), ("bfv_b", ctypes.c_uint64)]


def build(force=False):
    so = os.path.join(_HERE, "libcpuprover.so")
    host = os.path.join(_HERE, "..", "zk-fhe_amd", "host")
    srcs = [os.path.join(_HERE, f) for f in ("cpu_prover.cpp", "bn254_ref.h")] + [os.path.join(host, f) for f in os.listdir(host) if f.endswith((".hpp", "_ifma.cpp"))]
    srcs.append(os.path.join(host, "..", "csrc", "bn254.hip.hpp"))   # fe.hpp includes it
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libcpuprover.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libcpuprover.so")
        if not os.path.exists(so):
            build()
        # idle OpenMP workers sleep instead of spinning through the serial stretches (witness generation, transcript):
        # on a box whose container has fewer CPUs than it shows, spinning threads starve the working ones
        os.environ.setdefault("KMP_BLOCKTIME", "0")
        L = ctypes.CDLL(so)
        L.cpu_pk_create.restype = ctypes.c_void_p
        L.cpu_pk_create.argtypes = [ctypes.POINTER(_Cfg), ctypes.c_char_p, ctypes.c_char_p] + [ctypes.c_void_p] * 7 + [ctypes.c_char_p, ctypes.c_size_t]
        L.cpu_pk_destroy.argtypes = [ctypes.c_void_p]
        L.cpu_prove.restype = ctypes.c_int
        L.cpu_prove.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_size_t,
                                ctypes.POINTER(ctypes.c_size_t), ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
        _LIB = L
    return _LIB


def threads():
    return int(lib().cpu_prover_threads())


def set_threads(n):
    lib().cpu_prover_set_threads(int(n))


def usable_cpus():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota when there is one."""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def seed32(seed):
    """oracle/halo2_ref.py Rng's rule: pad up to 32 bytes, hash anything longer"""
    seed = bytes(seed)
    return seed.ljust(32, b"\0") if len(seed) <= 32 else hashlib.blake2b(seed, digest_size=32, person=b"zkfhe-seed").digest()


class CpuProver:
    """cfg: halo2_ref.Config, pk: halo2_ref.ProvingKey (its keygen), srs: halo2_ref.make_srs, prm: circuit_ref.BfvParams"""

    def __init__(self, cfg, pk, srs, prm):
        from .halo2_ref import TRANSCRIPT_ID
        L = lib()
        bps = [np.ascontiguousarray(pk.break_points[name], dtype=np.uint32) for name in ("gate0", "gate1", "rlc")]
        c = _Cfg(cfg.k, cfg.n_gate0, cfg.n_gate1, cfg.n_lookup, cfg.n_rlc, cfg.unusable_rows, cfg.lookup_bits, TRANSCRIPT_ID[cfg.transcript],
                 bps[0].ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), len(bps[0]),
                 bps[1].ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), len(bps[1]),
                 bps[2].ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), len(bps[2]),
                 prm.N, prm.Q, prm.T, prm.B)
        arrs = [np.ascontiguousarray(a, dtype=np.uint64) for a in (pk.fixed_lagrange, pk.sigma_lagrange, pk.fixed_coeff, pk.sigma_coeff, pk.l_coeff,
                                                                    srs["g_lagrange"], srs["g"])]
        assert arrs[0].shape == (cfg.n_fixed, cfg.n, 4) and arrs[1].shape == (cfg.n_perm, cfg.n, 4) and arrs[4].shape == (3, cfg.n, 4)
        assert arrs[5].shape == (cfg.n, 8) and arrs[6].shape == (cfg.n, 8)
        err = ctypes.create_string_buffer(256)
        self._h = L.cpu_pk_create(ctypes.byref(c), int(pk.vk_digest).to_bytes(32, "little"), int(pyref.FR_DELTA).to_bytes(32, "little"),
                                  *[a.ctypes.data_as(ctypes.c_void_p) for a in arrs], err, 256)
        if not self._h:
            raise RuntimeError("cpu_pk_create: " + err.value.decode())
        self.phase_ms = None

    def prove(self, input_json_text, seed):
        L = lib()
        out = ctypes.create_string_buffer(1 << 20)
        n = ctypes.c_size_t(0)
        ms = (ctypes.c_double * len(PHASES))()
        err = ctypes.create_string_buffer(256)
        rc = L.cpu_prove(self._h, input_json_text.encode() if isinstance(input_json_text, str) else input_json_text, seed32(seed), out, len(out),
                         ctypes.byref(n), ms, err, 256)
        if rc != 0:
            raise RuntimeError("cpu_prove failed (%d): %s" % (rc, err.value.decode()))
        self.phase_ms = dict(zip(PHASES, [float(v) for v in ms]))
        return out.raw[: n.value]

    def close(self):
        if self._h:
            lib().cpu_pk_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
