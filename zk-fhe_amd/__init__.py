"""zk-fhe_amd: host-side binding of the MI355X (gfx950) backend for zk-fhe's BFV-proof hot path.

This package is a thin ctypes mirror of include/zkfhe.h -- the C ABI is the product boundary, the
HIP kernels behind it are the product.  There is NO CPU fallback: if libzkfhe_hip.so is missing or
no gfx950 device is present, creating a Context raises.

The directory name contains a hyphen (it mirrors the reference's repository name), so import it
through the root shim:  `import zk_fhe_amd as zk`.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libzkfhe_hip.so")
_lib = None

EXPORTS = [
    "zkfhe_ctx_create", "zkfhe_ctx_destroy", "zkfhe_last_error", "zkfhe_sync", "zkfhe_stream", "zkfhe_device_info",
    "zkfhe_dev_alloc", "zkfhe_dev_free", "zkfhe_upload", "zkfhe_download", "zkfhe_copy_dev", "zkfhe_memset_dev",
    "zkfhe_timer_start", "zkfhe_timer_stop_ms", "zkfhe_prof_enable", "zkfhe_prof_reset", "zkfhe_prof_read", "zkfhe_prof_read_ops", "zkfhe_ctx_last_proof_marks",
    "zkfhe_fr_add", "zkfhe_fr_sub", "zkfhe_fr_mul", "zkfhe_fr_scale", "zkfhe_fr_to_mont", "zkfhe_fr_from_mont",
    "zkfhe_fr_batch_invert", "zkfhe_fr_sqr_chain", "zkfhe_fq29_sqr_chain",
    "zkfhe_ntt_batch", "zkfhe_ntt_batch_to", "zkfhe_coset_ntt_batch",
    "zkfhe_basis_create", "zkfhe_basis_destroy", "zkfhe_basis_len", "zkfhe_msm_batch",
    "zkfhe_g1_add", "zkfhe_g1_mul", "zkfhe_msm_sparse", "zkfhe_msm_batch_xyzz", "zkfhe_msm_sparse_xyzz", "zkfhe_g1_xyzz_to_affine", "zkfhe_basis_has_multiples", "zkfhe_basis_table_bits", "zkfhe_basis_table_bytes", "zkfhe_srs_table_bits", "zkfhe_srs_table_info",
    "zkfhe_comm_unique_id", "zkfhe_comm_create", "zkfhe_comm_create_with_transport", "zkfhe_comm_destroy", "zkfhe_comm_rank", "zkfhe_comm_world", "zkfhe_comm_active",
    "zkfhe_comm_point_range", "zkfhe_comm_all_gather", "zkfhe_comm_all_gather_async", "zkfhe_msm_batch_sharded", "zkfhe_msm_batch_sharded_async", "zkfhe_comm_join", "zkfhe_comm_record_event", "zkfhe_srs_create_sharded",
    "zkfhe_witness_poly_mul_u64", "zkfhe_host_poly_mul_u32", "zkfhe_witness_div_mod",
    "zkfhe_bfv_build_tables", "zkfhe_bfv_auto_config", "zkfhe_bfv_tables_free", "zkfhe_bfv_tables_count", "zkfhe_bfv_tables_copy_advice",
    "zkfhe_bfv_tables_copy_fixed", "zkfhe_bfv_tables_copy_instance", "zkfhe_bfv_tables_copy_copies",
    "zkfhe_bfv_tables_copy_break_points", "zkfhe_bfv_mock_check", "zkfhe_bfv_tables_poke_advice",
    "zkfhe_srs_create", "zkfhe_srs_from_points", "zkfhe_srs_destroy", "zkfhe_srs_save", "zkfhe_srs_drop_host_copy", "zkfhe_srs_load", "zkfhe_srs_g2", "zkfhe_srs_set_g2", "zkfhe_srs_file_g2",
    "zkfhe_chacha20_block", "zkfhe_snark_encode", "zkfhe_snark_decode", "zkfhe_bfv_keygen", "zkfhe_bfv_pk_destroy", "zkfhe_bfv_pk_release_ctx", "zkfhe_bfv_pk_info", "zkfhe_bfv_pk_prefix_cache",
    "zkfhe_bfv_pk_commitments", "zkfhe_bfv_pk_break_points", "zkfhe_bfv_pk_prehash", "zkfhe_bfv_prove", "zkfhe_bfv_pk_export_vk", "zkfhe_bfv_pk_save", "zkfhe_bfv_pk_load", "zkfhe_bfv_witness_stream", "zkfhe_lookup_permute", "zkfhe_bfv_verify", "zkfhe_bfv_verify_g2",
    "zkfhe_transcript_create", "zkfhe_transcript_destroy", "zkfhe_transcript_common_scalar", "zkfhe_transcript_write_scalar",
    "zkfhe_transcript_common_point", "zkfhe_transcript_write_point", "zkfhe_transcript_squeeze", "zkfhe_transcript_bytes",
    "zkfhe_poseidon_permute", "zkfhe_poseidon_constants", "zkfhe_poseidon_hash_many", "zkfhe_host_hash_mode", "zkfhe_prover_gate",
    "zkfhe_version",
]


# concurrent proofs run on one HIP stream each; give every stream its own hardware queue (ROCm's default is 4 per process,
# streams sharing a queue serialise).  Only effective if the HIP runtime has not been initialised yet in this process.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")


class ZkfheError(RuntimeError):
    pass


def load_library():
    """dlopen the in-tree HIP extension. Raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ZkfheError("HIP extension %s is missing: run `python zk-fhe_amd/build.py` (or __graft_entry__.build())" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    lib.zkfhe_last_error.restype = ctypes.c_char_p
    lib.zkfhe_last_error.argtypes = [vp]
    lib.zkfhe_version.restype = ctypes.c_char_p
    lib.zkfhe_stream.restype = vp
    lib.zkfhe_stream.argtypes = [vp]
    lib.zkfhe_basis_len.restype = sz
    lib.zkfhe_basis_len.argtypes = [vp]
    lib.zkfhe_ctx_create.argtypes = [ci, vp, ctypes.POINTER(vp)]
    lib.zkfhe_ctx_destroy.argtypes = [vp]
    lib.zkfhe_sync.argtypes = [vp]
    lib.zkfhe_device_info.argtypes = [vp, ctypes.c_char_p, sz, ctypes.POINTER(ci), ctypes.POINTER(sz)]
    lib.zkfhe_dev_alloc.argtypes = [vp, sz, ctypes.POINTER(vp)]
    lib.zkfhe_dev_free.argtypes = [vp, vp]
    lib.zkfhe_upload.argtypes = [vp, vp, vp, sz]
    lib.zkfhe_download.argtypes = [vp, vp, vp, sz]
    lib.zkfhe_copy_dev.argtypes = [vp, vp, vp, sz]
    lib.zkfhe_memset_dev.argtypes = [vp, vp, ci, sz]
    lib.zkfhe_timer_start.argtypes = [vp]
    lib.zkfhe_timer_stop_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
    for f in ("zkfhe_fr_add", "zkfhe_fr_sub", "zkfhe_fr_mul"):
        getattr(lib, f).argtypes = [vp, vp, vp, vp, sz]
    lib.zkfhe_fr_scale.argtypes = [vp, vp, vp, vp, sz]
    lib.zkfhe_fr_to_mont.argtypes = [vp, vp, vp, sz]
    lib.zkfhe_fr_from_mont.argtypes = [vp, vp, vp, sz]
    lib.zkfhe_fr_batch_invert.argtypes = [vp, vp, sz]
    lib.zkfhe_fr_sqr_chain.argtypes = [vp, vp, vp, sz, ci]
    lib.zkfhe_ntt_batch.argtypes = [vp, vp, sz, ci, ci]
    lib.zkfhe_coset_ntt_batch.argtypes = [vp, vp, vp, sz, ci, ci, vp, ci]
    lib.zkfhe_basis_create.argtypes = [vp, vp, sz, ci, ctypes.POINTER(vp)]
    lib.zkfhe_basis_destroy.argtypes = [vp, vp]
    lib.zkfhe_msm_batch.argtypes = [vp, vp, vp, sz, vp]
    lib.zkfhe_g1_add.argtypes = [vp, vp, vp, vp, sz]
    lib.zkfhe_g1_mul.argtypes = [vp, vp, vp, vp, sz]
    lib.zkfhe_witness_poly_mul_u64.argtypes = [vp, vp, vp, sz, vp]
    lib.zkfhe_witness_div_mod.argtypes = [vp, vp, ctypes.c_uint64, vp, vp, sz]
    _lib = lib
    return lib


class DeviceBuffer:
    """A device (HBM) allocation owned by a Context."""

    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        p = ctypes.c_void_p()
        ctx._check(ctx.lib.zkfhe_dev_alloc(ctx.h, self.nbytes, ctypes.byref(p)))
        self.ptr = p.value

    def free(self):
        if self.ptr:
            self.ctx.lib.zkfhe_dev_free(self.ctx.h, self.ptr)
            self.ptr = None

    def at(self, byte_offset):
        return ctypes.c_void_p(self.ptr + int(byte_offset))

    def upload(self, arr, byte_offset=0):
        a = np.ascontiguousarray(arr)
        assert byte_offset + a.nbytes <= self.nbytes
        self.ctx._check(self.ctx.lib.zkfhe_upload(self.ctx.h, self.at(byte_offset), a.ctypes.data_as(ctypes.c_void_p), a.nbytes))
        return self

    def download(self, dtype=np.uint64, shape=None, byte_offset=0, nbytes=None):
        nbytes = self.nbytes - byte_offset if nbytes is None else nbytes
        out = np.empty(nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        self.ctx._check(self.ctx.lib.zkfhe_download(self.ctx.h, out.ctypes.data_as(ctypes.c_void_p), self.at(byte_offset), nbytes))
        return out.reshape(shape) if shape is not None else out

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Basis:
    """Device-resident MSM basis (SRS half) with its per-window tables."""

    def __init__(self, ctx, bases, window_bits=0):
        b = np.ascontiguousarray(bases, dtype=np.uint64).reshape(-1, 8)
        self.ctx = ctx
        h = ctypes.c_void_p()
        ctx._check(ctx.lib.zkfhe_basis_create(ctx.h, b.ctypes.data_as(ctypes.c_void_p), b.shape[0], int(window_bits), ctypes.byref(h)))
        self.h = h
        self.n = b.shape[0]

    @property
    def has_table(self):
        """True when the basis holds a digit-multiple table (every MSM against it is a plain sum of table points)."""
        return bool(self.ctx.lib.zkfhe_basis_has_multiples(self.h))

    def destroy(self):
        if self.h:
            self.ctx.lib.zkfhe_basis_destroy(self.ctx.h, self.h)
            self.h = None


class Context:
    """One per GPU. Mirrors zkfhe_ctx; raises ZkfheError on any non-zero status."""

    def __init__(self, device_id=0, stream=None):
        self.lib = load_library()
        h = ctypes.c_void_p()
        rc = self.lib.zkfhe_ctx_create(int(device_id), ctypes.c_void_p(stream) if stream else None, ctypes.byref(h))
        if rc != 0:
            raise ZkfheError("zkfhe_ctx_create failed (%d): %s" % (rc, self.lib.zkfhe_last_error(None).decode()))
        self.h = h

    def _check(self, rc):
        if rc != 0:
            raise ZkfheError("zkfhe call failed (%d): %s" % (rc, self.lib.zkfhe_last_error(self.h).decode()))

    def close(self):
        if self.h:
            self.lib.zkfhe_ctx_destroy(self.h)
            self.h = None

    def sync(self):
        self._check(self.lib.zkfhe_sync(self.h))

    def device_info(self):
        name = ctypes.create_string_buffer(64)
        cu = ctypes.c_int()
        mem = ctypes.c_size_t()
        self._check(self.lib.zkfhe_device_info(self.h, name, 64, ctypes.byref(cu), ctypes.byref(mem)))
        return {"arch": name.value.decode(), "num_cu": cu.value, "hbm_bytes": mem.value}

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def to_device(self, arr):
        a = np.ascontiguousarray(arr)
        return self.alloc(a.nbytes).upload(a)

    def prof_enable(self, on=True):
        self.lib.zkfhe_prof_enable.argtypes = [ctypes.c_void_p, ctypes.c_int]
        self.lib.zkfhe_prof_reset.argtypes = [ctypes.c_void_p]
        self._check(self.lib.zkfhe_prof_reset(self.h))
        self._check(self.lib.zkfhe_prof_enable(self.h, int(bool(on))))

    def last_proof_marks(self):
        """zkfhe_ctx_last_proof_marks: ms from the start of the last proof on this context to (phase-0 commitment back from the GPU,
        first challenge squeezed, proof complete)"""
        m = (ctypes.c_float * 3)()
        self.lib.zkfhe_ctx_last_proof_marks.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
        self._check(self.lib.zkfhe_ctx_last_proof_marks(self.h, m))
        return list(m)

    def prof_read(self, which):
        self.lib.zkfhe_prof_read.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64),
                                             ctypes.POINTER(ctypes.c_double)]
        ms, cnt, by = ctypes.c_double(), ctypes.c_uint64(), ctypes.c_double()
        self._check(self.lib.zkfhe_prof_read(self.h, which, ctypes.byref(ms), ctypes.byref(cnt), ctypes.byref(by)))
        self.lib.zkfhe_prof_read_ops.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
        ops = ctypes.c_double()
        self._check(self.lib.zkfhe_prof_read_ops(self.h, which, ctypes.byref(ops)))
        return {"total_ms": ms.value, "launches": cnt.value, "algorithmic_bytes": by.value, "ops": ops.value}

    def timer_start(self):
        self._check(self.lib.zkfhe_timer_start(self.h))

    def timer_stop_ms(self):
        ms = ctypes.c_float()
        self._check(self.lib.zkfhe_timer_stop_ms(self.h, ctypes.byref(ms)))
        return ms.value

    # ---- device-resident calls (pointers are DeviceBuffer or c_void_p) ----
    @staticmethod
    def _p(x):
        return ctypes.c_void_p(x.ptr) if isinstance(x, DeviceBuffer) else x

    def fr_binop_dev(self, op, a, b, out, n):
        self._check(getattr(self.lib, "zkfhe_fr_" + op)(self.h, self._p(a), self._p(b), self._p(out), n))

    def ntt_dev(self, cols, n_cols, log_n, inverse=False):
        self._check(self.lib.zkfhe_ntt_batch(self.h, self._p(cols), n_cols, log_n, int(bool(inverse))))

    def ntt_to_dev(self, src, dst, n_cols, log_n, inverse=False):
        self.lib.zkfhe_ntt_batch_to.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int]
        self._check(self.lib.zkfhe_ntt_batch_to(self.h, self._p(src), self._p(dst), n_cols, log_n, int(bool(inverse))))

    def coset_ntt_dev(self, src, dst, n_cols, log_n, log_ext_factor, g, inverse=False):
        g = np.ascontiguousarray(g, dtype=np.uint64)
        self._check(self.lib.zkfhe_coset_ntt_batch(self.h, self._p(src), self._p(dst), n_cols, log_n, log_ext_factor,
                                                  g.ctypes.data_as(ctypes.c_void_p), int(bool(inverse))))

    def msm_dev(self, basis, scalars, n_cols, out):
        self._check(self.lib.zkfhe_msm_batch(self.h, basis.h, self._p(scalars), n_cols, self._p(out)))

    # ---- numpy convenience (host in, host out; used by tests and smoke) ----
    def _fr(self, a):
        return np.ascontiguousarray(a, dtype=np.uint64)

    def fr_binop(self, op, a, b):
        a, b = self._fr(a), self._fr(b)
        n = a.size // 4
        da, db = self.to_device(a), self.to_device(b)
        self.fr_binop_dev(op, da, db, da, n)
        out = da.download(shape=a.shape)
        da.free(), db.free()
        return out

    def fr_unop(self, name, a, *extra):
        a = self._fr(a)
        n = a.size // 4
        da = self.to_device(a)
        if name == "batch_invert":
            self._check(self.lib.zkfhe_fr_batch_invert(self.h, self._p(da), n))
        elif name == "scale":
            s = self._fr(extra[0])
            self._check(self.lib.zkfhe_fr_scale(self.h, self._p(da), s.ctypes.data_as(ctypes.c_void_p), self._p(da), n))
        elif name == "sqr_chain":
            self._check(self.lib.zkfhe_fr_sqr_chain(self.h, self._p(da), self._p(da), n, int(extra[0])))
        else:
            self._check(getattr(self.lib, "zkfhe_fr_" + name)(self.h, self._p(da), self._p(da), n))
        out = da.download(shape=a.shape)
        da.free()
        return out

    def ntt(self, cols, log_n, inverse=False):
        a = self._fr(cols)
        n = 1 << log_n
        n_cols = a.size // 4 // n
        d = self.to_device(a)
        self.ntt_dev(d, n_cols, log_n, inverse)
        out = d.download(shape=a.shape)
        d.free()
        return out

    def coset_ntt(self, cols, log_n, log_ext_factor, g, inverse=False):
        a = self._fr(cols)
        n, ne = 1 << log_n, 1 << (log_n + log_ext_factor)
        if not inverse:
            n_cols = a.size // 4 // n
            src, dst = self.to_device(a), self.alloc(n_cols * ne * 32)
        else:
            n_cols = a.size // 4 // ne
            src, dst = self.to_device(a), self.alloc(n_cols * ne * 32)
        self.coset_ntt_dev(src, dst, n_cols, log_n, log_ext_factor, g, inverse)
        out = dst.download(shape=(n_cols, ne, 4))
        src.free(), dst.free()
        return out

    def msm(self, basis, scalars):
        s = self._fr(scalars)
        n_cols = s.size // 4 // basis.n
        ds, do = self.to_device(s), self.alloc(n_cols * 64)
        self.msm_dev(basis, ds, n_cols, do)
        out = do.download(shape=(n_cols, 8))
        ds.free(), do.free()
        return out

    def msm_xyzz(self, basis, scalars):
        """zkfhe_msm_batch_xyzz + zkfhe_g1_xyzz_to_affine: the sums in accumulator form from the GPU, ONE inversion for all of them
        on the host.  Returns (affine (n_cols, 8) uint64, raw xyzz (n_cols, 16) uint64)."""
        scalars = self._fr(scalars)
        n_cols = scalars.size // 4 // basis.n
        vp = ctypes.c_void_p
        self.lib.zkfhe_msm_batch_xyzz.argtypes = [vp, vp, vp, ctypes.c_size_t, vp]
        self.lib.zkfhe_g1_xyzz_to_affine.argtypes = [vp, ctypes.c_size_t, vp]
        d = self.to_device(scalars)
        out = self.alloc(n_cols * 128)
        self._check(self.lib.zkfhe_msm_batch_xyzz(self.h, basis.h, d.ptr, n_cols, out.ptr))
        raw = np.ascontiguousarray(out.download(dtype=np.uint64, shape=(n_cols, 16)))
        aff = np.empty((n_cols, 8), dtype=np.uint64)
        self._check(self.lib.zkfhe_g1_xyzz_to_affine(raw.ctypes.data_as(vp), n_cols, aff.ctypes.data_as(vp)))
        d.free(), out.free()
        return aff, raw

    def msm_sharded(self, comm, basis_slice, scalars, lo):
        """zkfhe_msm_batch_sharded: `scalars` = the FULL columns (n_cols, n_full, 4); this rank's basis slice covers rows
        lo .. lo + len(basis_slice).  Every rank gets the full commitments."""
        s = self._fr(scalars)
        n_cols, n_full = s.shape[0], s.shape[1]
        ds, do = self.to_device(s), self.alloc(n_cols * 64)
        self.lib.zkfhe_msm_batch_sharded.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p]
        self._check(self.lib.zkfhe_msm_batch_sharded(self.h, comm.h, basis_slice.h, ds.at(lo * 32), n_full, n_cols, do.ptr))
        out = do.download(shape=(n_cols, 8))
        ds.free(), do.free()
        return out

    def g1_add(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 8)
        b = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, 8)
        da, db = self.to_device(a), self.to_device(b)
        self._check(self.lib.zkfhe_g1_add(self.h, self._p(da), self._p(db), self._p(da), a.shape[0]))
        out = da.download(shape=a.shape)
        da.free(), db.free()
        return out

    def g1_mul(self, p, k):
        p = np.ascontiguousarray(p, dtype=np.uint64).reshape(-1, 8)
        k = np.ascontiguousarray(k, dtype=np.uint64).reshape(-1, 4)
        dp, dk = self.to_device(p), self.to_device(k)
        self._check(self.lib.zkfhe_g1_mul(self.h, self._p(dp), self._p(dk), self._p(dp), p.shape[0]))
        out = dp.download(shape=p.shape)
        dp.free(), dk.free()
        return out

    def witness_poly_mul_u64(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64)
        n = a.size
        da, db, do = self.to_device(a), self.to_device(b), self.alloc((2 * n - 1) * 32)
        self._check(self.lib.zkfhe_witness_poly_mul_u64(self.h, self._p(da), self._p(db), n, self._p(do)))
        out = do.download(shape=(2 * n - 1, 4))
        da.free(), db.free(), do.free()
        return out

    def witness_div_mod(self, a, q):
        a = self._fr(a).reshape(-1, 4)
        n = a.shape[0]
        da, dd, dr = self.to_device(a), self.alloc(n * 32), self.alloc(n * 32)
        self._check(self.lib.zkfhe_witness_div_mod(self.h, self._p(da), ctypes.c_uint64(int(q)), self._p(dd), self._p(dr), n))
        d, r = dd.download(shape=(n, 4)), dr.download(shape=(n, 4))
        da.free(), dd.free(), dr.free()
        return d, r


def version():
    return load_library().zkfhe_version().decode()


def seed32(seed):
    """The 32-byte blinding seed of zkfhe_bfv_prove from arbitrary bytes: up to 32 bytes are zero-padded, anything longer
    is hashed (BLAKE2b-256, person "zkfhe-seed") so that long seeds sharing a prefix do not collide."""
    import hashlib
    seed = bytes(seed)
    if len(seed) <= 32:
        return seed.ljust(32, b"\x00")
    return hashlib.blake2b(seed, digest_size=32, person=b"zkfhe-seed").digest()


# ----------------------------------------------------------------------------- transcript (host only)
class HostTranscript:
    """zkfhe_transcript_*: the prover's Fiat-Shamir transcript (Poseidon as in snark-verifier, or halo2's Blake2b)."""

    def __init__(self, kind="poseidon"):
        self.lib = load_library()
        self.lib.zkfhe_transcript_destroy.restype = None
        self.lib.zkfhe_transcript_destroy.argtypes = [ctypes.c_void_p]
        for f in ("common_scalar", "write_scalar", "common_point", "write_point", "squeeze"):
            getattr(self.lib, "zkfhe_transcript_" + f).argtypes = [ctypes.c_void_p, ctypes.c_char_p]
        self.lib.zkfhe_transcript_bytes.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
        self.h = ctypes.c_void_p()
        rc = self.lib.zkfhe_transcript_create(ctypes.c_uint32(TRANSCRIPT_ID[kind]), ctypes.byref(self.h))
        if rc != 0:
            raise ZkfheError("zkfhe_transcript_create failed (%d)" % rc)

    def _call(self, name, payload):
        rc = getattr(self.lib, "zkfhe_transcript_" + name)(self.h, payload)
        if rc != 0:
            raise ZkfheError("zkfhe_transcript_%s refused its argument (%d)" % (name, rc))

    def common_scalar(self, s):
        self._call("common_scalar", int(s).to_bytes(32, "little"))

    def write_scalar(self, s):
        self._call("write_scalar", int(s).to_bytes(32, "little"))

    def common_point(self, P):
        x, y = (0, 0) if P is None else P
        self._call("common_point", x.to_bytes(32, "little") + y.to_bytes(32, "little"))

    def write_point(self, P):
        x, y = (0, 0) if P is None else P
        self._call("write_point", x.to_bytes(32, "little") + y.to_bytes(32, "little"))

    def squeeze(self):
        buf = ctypes.create_string_buffer(32)
        self._call("squeeze", buf)
        return int.from_bytes(buf.raw, "little")

    def stream(self):
        n = ctypes.c_size_t(0)
        self.lib.zkfhe_transcript_bytes(self.h, None, 0, ctypes.byref(n))
        buf = ctypes.create_string_buffer(max(1, n.value))
        self.lib.zkfhe_transcript_bytes(self.h, buf, n.value, ctypes.byref(n))
        return buf.raw[:n.value]

    def close(self):
        if self.h:
            self.lib.zkfhe_transcript_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def poseidon_permute(state):
    """one Poseidon permutation (BN254 Fr, t = 3, R_F = 8, R_P = 57) of three python ints, by the host library"""
    lib = load_library()
    buf = ctypes.create_string_buffer(b"".join(int(v).to_bytes(32, "little") for v in state), 96)
    lib.zkfhe_poseidon_permute.argtypes = [ctypes.c_char_p]
    rc = lib.zkfhe_poseidon_permute(buf)
    if rc != 0:
        raise ZkfheError("zkfhe_poseidon_permute refused its argument (%d)" % rc)
    return [int.from_bytes(buf.raw[32 * i:32 * i + 32], "little") for i in range(3)]


def host_hash_mode(mode=None):
    """'latency' | 'shared' | None (query): how concurrent provers' Poseidon transcripts hash (zkfhe_host_hash_mode)"""
    lib = load_library()
    lib.zkfhe_host_hash_mode.argtypes = [ctypes.c_int]
    code = {None: -1, "latency": 0, "shared": 1}[mode]
    rc = lib.zkfhe_host_hash_mode(code)
    if rc < 0:
        raise ZkfheError("zkfhe_host_hash_mode(%r) failed" % (mode,))
    return ["latency", "shared"][rc]


def prover_gate(n=-1):
    """zkfhe_prover_gate: proofs admitted at once to the GPU-heavy middle of a proof (0 = no gate, negative = query); returns the previous setting"""
    lib = load_library()
    lib.zkfhe_prover_gate.argtypes = [ctypes.c_int]
    return int(lib.zkfhe_prover_gate(int(n)))


def poseidon_hash_many(sequences, mode=1):
    """sponge digests of independent scalar sequences (python ints); mode 0 = single-sponge path, 1 = eight AVX-512 lanes on this
    thread, 2 = through the hash service threads (what concurrent provers use).  None when the CPU lacks AVX-512 IFMA (modes 1, 2)."""
    lib = load_library()
    flat = b"".join(int(v).to_bytes(32, "little") for s in sequences for v in s)
    counts = (ctypes.c_size_t * max(1, len(sequences)))(*[len(s) for s in sequences])
    out = ctypes.create_string_buffer(32 * max(1, len(sequences)))
    lib.zkfhe_poseidon_hash_many.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_size_t), ctypes.c_size_t, ctypes.c_int, ctypes.c_char_p]
    rc = lib.zkfhe_poseidon_hash_many(flat, counts, len(sequences), mode, out)
    if rc == -4:
        return None
    if rc != 0:
        raise ZkfheError("zkfhe_poseidon_hash_many failed (%d)" % rc)
    return [int.from_bytes(out.raw[32 * i:32 * i + 32], "little") for i in range(len(sequences))]


def poseidon_constants():
    lib = load_library()
    rc = ctypes.create_string_buffer(65 * 3 * 32)
    mds = ctypes.create_string_buffer(9 * 32)
    lib.zkfhe_poseidon_constants.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    lib.zkfhe_poseidon_constants(rc, mds)
    rcv = [int.from_bytes(rc.raw[32 * i:32 * i + 32], "little") for i in range(195)]
    mdv = [int.from_bytes(mds.raw[32 * i:32 * i + 32], "little") for i in range(9)]
    return [rcv[3 * r:3 * r + 3] for r in range(65)], [mdv[3 * i:3 * i + 3] for i in range(3)]


# ----------------------------------------------------------------------------- BFV circuit (host layer)
class BfvParamsC(ctypes.Structure):
    _fields_ = [("n", ctypes.c_uint64), ("q", ctypes.c_uint64), ("t", ctypes.c_uint64), ("b", ctypes.c_uint64)]


class BfvConfigC(ctypes.Structure):
    _fields_ = [("k", ctypes.c_uint32), ("n_gate0", ctypes.c_uint32), ("n_gate1", ctypes.c_uint32), ("n_lookup", ctypes.c_uint32),
                ("n_rlc", ctypes.c_uint32), ("unusable_rows", ctypes.c_uint32), ("lookup_bits", ctypes.c_uint32),
                ("bp_gate0", ctypes.POINTER(ctypes.c_uint32)), ("n_bp_gate0", ctypes.c_uint32),
                ("bp_gate1", ctypes.POINTER(ctypes.c_uint32)), ("n_bp_gate1", ctypes.c_uint32),
                ("bp_rlc", ctypes.POINTER(ctypes.c_uint32)), ("n_bp_rlc", ctypes.c_uint32),
                ("replay", ctypes.c_int), ("transcript", ctypes.c_uint32)]


TRANSCRIPT_ID = {"poseidon": 0, "blake2b": 1}


class BfvConfig:
    """configs/<name>.json: column counts + break points (the reference's pinning, README.md:38)."""

    def __init__(self, k, n_gate0, n_gate1, n_lookup, n_rlc, unusable_rows, lookup_bits=8, break_points=None, transcript="poseidon"):
        self.k, self.n_gate0, self.n_gate1, self.n_lookup, self.n_rlc = k, n_gate0, n_gate1, n_lookup, n_rlc
        self.unusable_rows, self.lookup_bits = unusable_rows, lookup_bits
        self.break_points = break_points  # dict gate0/gate1/rlc or None
        if transcript not in TRANSCRIPT_ID:
            raise ValueError("transcript must be 'poseidon' (the reference's) or 'blake2b'")
        self.transcript = transcript

    @staticmethod
    def from_pinning(cfg_json, transcript="poseidon"):
        p = cfg_json["params"]
        bp = cfg_json.get("break_points")
        bpd = None
        if bp:
            bpd = {"gate0": bp["gate"][0], "gate1": bp["gate"][1], "rlc": bp["rlc"]}
        return BfvConfig(p["degree"], p["num_range_advice"][0], p["num_range_advice"][1], p["num_lookup_advice"][1],
                         p["num_rlc_columns"], p["unusable_rows"], p["lookup_bits"], bpd, transcript)

    def to_c(self, replay):
        c = BfvConfigC(self.k, self.n_gate0, self.n_gate1, self.n_lookup, self.n_rlc, self.unusable_rows, self.lookup_bits)
        self._keep = []
        if self.break_points:
            for name in ("gate0", "gate1", "rlc"):
                arr = (ctypes.c_uint32 * len(self.break_points[name]))(*self.break_points[name])
                self._keep.append(arr)
                setattr(c, "bp_" + name, ctypes.cast(arr, ctypes.POINTER(ctypes.c_uint32)))
                setattr(c, "n_bp_" + name, len(self.break_points[name]))
        c.replay = 1 if (replay and self.break_points) else 0
        c.transcript = TRANSCRIPT_ID[self.transcript]
        return c


def _bfv_sigs(lib):
    vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    lib.zkfhe_bfv_build_tables.argtypes = [ctypes.c_char_p, ctypes.POINTER(BfvParamsC), ctypes.POINTER(BfvConfigC), ctypes.c_char_p, ci,
                                           ctypes.POINTER(vp), ctypes.c_char_p, sz]
    lib.zkfhe_bfv_tables_free.argtypes = [vp]
    lib.zkfhe_bfv_tables_free.restype = None
    lib.zkfhe_bfv_tables_count.argtypes = [vp, ci]
    lib.zkfhe_bfv_tables_count.restype = sz
    for f in ("advice", "fixed", "instance", "copies"):
        getattr(lib, "zkfhe_bfv_tables_copy_" + f).argtypes = [vp, vp]
    lib.zkfhe_bfv_tables_copy_break_points.argtypes = [vp, ci, vp]


def bfv_auto_config(input_json_text, params, k, unusable_rows=109, lookup_bits=8, transcript="poseidon"):
    """zkfhe_bfv_auto_config: the BfvConfig (column counts) the circuit needs at 2^k rows; host only."""
    lib = load_library()
    lib.zkfhe_bfv_auto_config.argtypes = [ctypes.c_char_p, ctypes.POINTER(BfvParamsC), ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                          ctypes.POINTER(ctypes.c_uint32), ctypes.c_char_p, ctypes.c_size_t]
    prm = BfvParamsC(*params)
    counts = (ctypes.c_uint32 * 4)()
    err = ctypes.create_string_buffer(256)
    text = input_json_text if isinstance(input_json_text, bytes) else input_json_text.encode()
    rc = lib.zkfhe_bfv_auto_config(text, ctypes.byref(prm), k, unusable_rows, lookup_bits, counts, err, 256)
    if rc != 0:
        raise ZkfheError("zkfhe_bfv_auto_config failed (%d): %s" % (rc, err.value.decode()))
    return BfvConfig(k, counts[0], counts[1], counts[2], counts[3], unusable_rows, lookup_bits, transcript=transcript)


def bfv_build_tables(input_json_text, params, config, gamma, keygen_mode, replay=False):
    """Host-only witness tables (no GPU). params = (N, Q, T, B); gamma = python int. Returns a dict of numpy arrays."""
    lib = load_library()
    _bfv_sigs(lib)
    prm = BfvParamsC(*params)
    cfg = config.to_c(replay)
    h = ctypes.c_void_p()
    err = ctypes.create_string_buffer(512)
    rc = lib.zkfhe_bfv_build_tables(input_json_text.encode(), ctypes.byref(prm), ctypes.byref(cfg), int(gamma).to_bytes(32, "little"),
                                    int(bool(keygen_mode)), ctypes.byref(h), err, 512)
    if rc != 0:
        raise ZkfheError("zkfhe_bfv_build_tables failed (%d): %s" % (rc, err.value.decode()))
    cnt = lambda w: int(lib.zkfhe_bfv_tables_count(h, w))  # noqa: E731
    n_adv, n_fix, n, n_inst, n_cp = cnt(0), cnt(1), cnt(2), cnt(3), cnt(4)
    out = {"n": n, "cells": (cnt(8), cnt(9), cnt(10)), "lookups": cnt(11)}
    adv = np.empty((n_adv, n, 4), dtype=np.uint64)
    lib.zkfhe_bfv_tables_copy_advice(h, adv.ctypes.data_as(ctypes.c_void_p))
    out["advice"] = adv
    if n_fix:
        fx = np.empty((n_fix, n, 4), dtype=np.uint64)
        lib.zkfhe_bfv_tables_copy_fixed(h, fx.ctypes.data_as(ctypes.c_void_p))
        out["fixed"] = fx
    inst = np.empty((n_inst, 4), dtype=np.uint64)
    lib.zkfhe_bfv_tables_copy_instance(h, inst.ctypes.data_as(ctypes.c_void_p))
    out["instance"] = inst
    cp = np.empty((n_cp, 2), dtype=np.uint64)
    if n_cp:
        lib.zkfhe_bfv_tables_copy_copies(h, cp.ctypes.data_as(ctypes.c_void_p))
    out["copies"] = cp
    bps = {}
    for i, name in enumerate(("gate0", "gate1", "rlc")):
        a = np.empty(cnt(5 + i), dtype=np.uint32)
        if a.size:
            lib.zkfhe_bfv_tables_copy_break_points(h, i, a.ctypes.data_as(ctypes.c_void_p))
        bps[name] = a.tolist()
    out["break_points"] = bps
    lib.zkfhe_bfv_tables_free(h)
    return out


def bfv_mock(input_json_text, params, config, gamma=7, pokes=()):
    """`mock` (README.md:18-22): build the table (keygen mode: values, fixed columns, copy constraints) and check every
    constraint on every row.  pokes: [(advice column, row, value)] overwritten before the check (negative tests).
    Returns (number of violated rows / constraints, description of the first)."""
    lib = load_library()
    _bfv_sigs(lib)
    lib.zkfhe_bfv_mock_check.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_char_p, ctypes.c_size_t]
    lib.zkfhe_bfv_tables_poke_advice.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_char_p]
    prm = BfvParamsC(*params)
    cfg = config.to_c(False)
    h = ctypes.c_void_p()
    err = ctypes.create_string_buffer(512)
    g = int(gamma).to_bytes(32, "little")
    text = input_json_text if isinstance(input_json_text, bytes) else input_json_text.encode()
    rc = lib.zkfhe_bfv_build_tables(text, ctypes.byref(prm), ctypes.byref(cfg), g, 1, ctypes.byref(h), err, 512)
    if rc != 0:
        raise ZkfheError("circuit is not satisfied (%d): %s" % (rc, err.value.decode()))
    try:
        for col, row, val in pokes:
            if lib.zkfhe_bfv_tables_poke_advice(h, col, row, int(val).to_bytes(32, "little")) != 0:
                raise ZkfheError("zkfhe_bfv_tables_poke_advice refused (%d, %d)" % (col, row))
        nf = ctypes.c_uint64(0)
        rc = lib.zkfhe_bfv_mock_check(h, g, ctypes.byref(nf), err, 512)
        if rc != 0:
            raise ZkfheError("zkfhe_bfv_mock_check failed (%d): %s" % (rc, err.value.decode()))
        return int(nf.value), err.value.decode()
    finally:
        lib.zkfhe_bfv_tables_free(h)


# ----------------------------------------------------------------------------- SRS / keygen / prove (GPU)
ALLGATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p)


class Comm:
    """zkfhe_comm: the ranks that shard the commitments of one proof (one process per GPU).
    unique_id (128 bytes from Comm.unique_id() on rank 0, shared out of band): RCCL over xGMI, ncclAllGather on the context's
    stream.  all_gather (callable: bytes -> list of `world` byte strings, e.g. zk_fhe_amd.batch.all_gather_bytes over gloo):
    a host transport instead, for tests and for hosts with their own."""

    def __init__(self, ctx, rank, world, unique_id=None, all_gather=None):
        lib = ctx.lib
        self.ctx, self.rank, self.world = ctx, rank, world
        self.h = ctypes.c_void_p()
        lib.zkfhe_comm_destroy.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        if world > 1 and unique_id is None and all_gather is None:
            raise ValueError("Comm(world > 1) needs a transport: unique_id= (RCCL) or all_gather= (a host all-gather callable)")
        if unique_id is None:     # host transport (or, with world == 1 and no callback, no transport at all)
            def cb(_user, send, nbytes, recv):
                try:
                    parts = all_gather(ctypes.string_at(send, nbytes))
                    data = b"".join(parts)
                    if len(data) != nbytes * world:
                        return 1
                    ctypes.memmove(recv, data, len(data))
                    return 0
                except Exception:  # noqa: BLE001  (must not propagate into C)
                    return 1
            self._cb = ALLGATHER_FN(cb)   # kept alive for the lifetime of the communicator
            lib.zkfhe_comm_create_with_transport.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ALLGATHER_FN, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
            ctx._check(lib.zkfhe_comm_create_with_transport(ctx.h, rank, world, self._cb, None, ctypes.byref(self.h)))
        else:
            lib.zkfhe_comm_create.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p)]
            ctx._check(lib.zkfhe_comm_create(ctx.h, rank, world, bytes(unique_id), ctypes.byref(self.h)))

    @staticmethod
    def unique_id():
        lib = load_library()
        buf = ctypes.create_string_buffer(128)
        rc = lib.zkfhe_comm_unique_id(buf)
        if rc != 0:
            raise ZkfheError("zkfhe_comm_unique_id failed (%d): is librccl.so available?" % rc)
        return buf.raw

    def all_gather(self, send, recv, nbytes):
        """zkfhe_comm_all_gather on the context's stream: DeviceBuffer send (nbytes) -> DeviceBuffer recv (world * nbytes)"""
        lib = self.ctx.lib
        lib.zkfhe_comm_all_gather.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        self.ctx._check(lib.zkfhe_comm_all_gather(self.ctx.h, self.h, send.ptr, recv.ptr, nbytes))

    def all_gather_async(self, send, recv, nbytes, byte_offset_send=0, byte_offset_recv=0):
        """zkfhe_comm_all_gather_async: the same collective on the communicator's own stream (behind what is queued on the context's
        stream); join() orders the context's stream or the caller behind it."""
        lib = self.ctx.lib
        lib.zkfhe_comm_all_gather_async.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        self.ctx._check(lib.zkfhe_comm_all_gather_async(self.ctx.h, self.h, send.at(byte_offset_send), recv.at(byte_offset_recv), nbytes))

    def join(self, block_host=False):
        """zkfhe_comm_join: the context's stream (or, with block_host, the calling thread) waits for the collectives queued so far"""
        lib = self.ctx.lib
        lib.zkfhe_comm_join.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        self.ctx._check(lib.zkfhe_comm_join(self.ctx.h, self.h, 1 if block_host else 0))

    def point_range(self, n):
        lo, hi = ctypes.c_size_t(), ctypes.c_size_t()
        self.ctx.lib.zkfhe_comm_point_range.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
        self.ctx.lib.zkfhe_comm_point_range.restype = None
        self.ctx.lib.zkfhe_comm_point_range(self.h, n, ctypes.byref(lo), ctypes.byref(hi))
        return lo.value, hi.value

    def destroy(self):
        if self.h:
            self.ctx.lib.zkfhe_comm_destroy(self.ctx.h, self.h)
            self.h = ctypes.c_void_p()


# zkfhe.h ZKFHE_SRS_HALO2_UNSAFE: as a seed it selects the reference's own derivation (ChaCha20Rng::from_seed([0; 32]))
SRS_HALO2_UNSAFE = b"halo2:ParamsKZG::setup(k, ChaCha20Rng::from_seed([0u8; 32]))"


def chacha20_block(key, counter_nonce):
    """one ChaCha20 block (RFC 7539 2.3): key 32 bytes, counter_nonce = the four state words 12..15"""
    lib = load_library()
    out = ctypes.create_string_buffer(64)
    lib.zkfhe_chacha20_block.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint32), ctypes.c_char_p]
    if lib.zkfhe_chacha20_block(bytes(key), (ctypes.c_uint32 * 4)(*counter_nonce), out) != 0:
        raise ZkfheError("zkfhe_chacha20_block refused its arguments")
    return out.raw


def _g2_tuple(b):
    v = [int.from_bytes(b[32 * i:32 * i + 32], "little") for i in range(4)]
    return ((v[0], v[1]), (v[2], v[3]))


def srs_file_g2(path):
    """(k, G2, s G2) from the tail of a params/kzg_bn254_<k>.srs file; points as ((x.c0, x.c1), (y.c0, y.c1)).  Host only."""
    lib = load_library()
    a, b, k = ctypes.create_string_buffer(128), ctypes.create_string_buffer(128), ctypes.c_uint32()
    lib.zkfhe_srs_file_g2.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint32), ctypes.c_char_p, ctypes.c_char_p]
    if lib.zkfhe_srs_file_g2(os.fsencode(path), ctypes.byref(k), a, b) != 0:
        raise ZkfheError("%s is not a params file" % path)
    return k.value, _g2_tuple(a.raw), _g2_tuple(b.raw)


def snark_encode(instances, proof):
    """data/<name>.snark container (zkfhe.h zkfhe_snark_encode): instances = ints or an Instances object"""
    lib = load_library()
    inst = instances.raw if isinstance(instances, Instances) else b"".join(int(v).to_bytes(32, "little") for v in instances)
    n = len(inst) // 32
    lib.zkfhe_snark_encode.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    ln = ctypes.c_size_t()
    lib.zkfhe_snark_encode(inst, n, bytes(proof), len(proof), None, 0, ctypes.byref(ln))
    out = ctypes.create_string_buffer(ln.value)
    if lib.zkfhe_snark_encode(inst, n, bytes(proof), len(proof), out, ln.value, ctypes.byref(ln)) != 0:
        raise ZkfheError("zkfhe_snark_encode: an instance is not a reduced scalar")
    return out.raw


def snark_decode(snark):
    """-> (list of instance ints, proof bytes); raises on a malformed container"""
    lib = load_library()
    lib.zkfhe_snark_decode.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.POINTER(ctypes.c_size_t), ctypes.c_char_p, ctypes.POINTER(ctypes.c_size_t)]
    n, pl = ctypes.c_size_t(), ctypes.c_size_t()
    if lib.zkfhe_snark_decode(bytes(snark), len(snark), None, ctypes.byref(n), None, ctypes.byref(pl)) != 0:
        raise ZkfheError("not a snark container")
    inst, proof = ctypes.create_string_buffer(32 * n.value + 1), ctypes.create_string_buffer(pl.value + 1)
    if lib.zkfhe_snark_decode(bytes(snark), len(snark), inst, ctypes.byref(n), proof, ctypes.byref(pl)) != 0:
        raise ZkfheError("snark container: an instance is not a reduced scalar")
    return [int.from_bytes(inst.raw[32 * i:32 * i + 32], "little") for i in range(n.value)], proof.raw[:pl.value]


class Srs:
    def __init__(self, ctx, k, seed=b"zkfhe-unsafe-srs", comm=None):
        """comm: a Comm -> only this rank's point range of both halves is built and every commitment of keygen / prove is
        sharded over the communicator (zkfhe_srs_create_sharded)."""
        lib = ctx.lib
        lib.zkfhe_srs_create.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]
        lib.zkfhe_srs_create_sharded.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]
        lib.zkfhe_srs_destroy.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self.ctx, self.k, self.comm = ctx, k, comm
        h = ctypes.c_void_p()
        if comm is None:
            ctx._check(lib.zkfhe_srs_create(ctx.h, k, bytes(seed), len(seed), ctypes.byref(h)))
        else:
            ctx._check(lib.zkfhe_srs_create_sharded(ctx.h, comm.h, k, bytes(seed), len(seed), ctypes.byref(h)))
        self.h = h

    @classmethod
    def from_points(cls, ctx, k, g, g_lagrange):
        """zkfhe_srs_from_points: an SRS computed elsewhere.  g, g_lagrange: (2^k, 8) uint64 arrays, affine Montgomery limbs."""
        lib = ctx.lib
        lib.zkfhe_srs_from_points.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
        lib.zkfhe_srs_destroy.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        a = np.ascontiguousarray(g, dtype=np.uint64).reshape(-1, 8)
        b = np.ascontiguousarray(g_lagrange, dtype=np.uint64).reshape(-1, 8)
        if a.shape[0] != 1 << k or b.shape[0] != 1 << k:
            raise ValueError("an SRS for k = %d needs 2^k points in both bases" % k)
        self = cls.__new__(cls)
        self.ctx, self.k = ctx, k
        h = ctypes.c_void_p()
        ctx._check(lib.zkfhe_srs_from_points(ctx.h, k, a.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p), ctypes.byref(h)))
        self.h = h
        return self

    @classmethod
    def load(cls, ctx, path):
        """zkfhe_srs_load: a params/kzg_bn254_<k>.srs file (halo2 ParamsKZG RawBytes layout)"""
        lib = ctx.lib
        lib.zkfhe_srs_load.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p)]
        lib.zkfhe_srs_destroy.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self = cls.__new__(cls)
        h = ctypes.c_void_p()
        ctx._check(lib.zkfhe_srs_load(ctx.h, os.fsencode(path), ctypes.byref(h)))
        self.ctx, self.h, self.comm = ctx, h, None
        self.k = int.from_bytes(open(path, "rb").read(4), "little")
        return self

    def save(self, path):
        lib = self.ctx.lib
        lib.zkfhe_srs_save.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p]
        self.ctx._check(lib.zkfhe_srs_save(self.ctx.h, self.h, os.fsencode(path)))

    def drop_host_copy(self):
        """zkfhe_srs_drop_host_copy: give back the host copies kept for save() (a later save() raises)."""
        self.ctx.lib.zkfhe_srs_drop_host_copy.argtypes = [ctypes.c_void_p]
        self.ctx._check(self.ctx.lib.zkfhe_srs_drop_host_copy(self.h))

    def g2(self):
        """(G2, s G2) as ((x.c0, x.c1), (y.c0, y.c1)) ints: what bfv_verify(g2=, s_g2=) takes"""
        lib = self.ctx.lib
        a, b = ctypes.create_string_buffer(128), ctypes.create_string_buffer(128)
        lib.zkfhe_srs_g2.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p]
        if lib.zkfhe_srs_g2(self.h, a, b) != 0:
            raise ZkfheError("this SRS has no G2 half")
        return _g2_tuple(a.raw), _g2_tuple(b.raw)

    def set_g2(self, g2, s_g2):
        lib = self.ctx.lib
        enc = lambda p: b"".join(int(v).to_bytes(32, "little") for v in (p[0][0], p[0][1], p[1][0], p[1][1]))  # noqa: E731
        lib.zkfhe_srs_set_g2.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p]
        if lib.zkfhe_srs_set_g2(self.h, enc(g2), enc(s_g2)) != 0:
            raise ZkfheError("zkfhe_srs_set_g2: a point is not on the twist")

    def table_bits(self):
        """(digit width of the Lagrange half's digit-multiple table or 0, whether calls of many columns take the table path)."""
        lib = self.ctx.lib
        lib.zkfhe_srs_table_bits.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
        wide = ctypes.c_int()
        bits = lib.zkfhe_srs_table_bits(self.h, ctypes.byref(wide))
        return int(bits), bool(wide.value)

    def table_info(self):
        """zkfhe_srs_table_info: {"bits": (monomial half, Lagrange half), "gb": resident GB of both tables, "narrowed": whether a half
        got a narrower table than its budget allowed because the device did not have the room}."""
        lib = self.ctx.lib
        lib.zkfhe_srs_table_info.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int * 2), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_int)]
        bits, nbytes, narrowed = (ctypes.c_int * 2)(), ctypes.c_uint64(), ctypes.c_int()
        self.ctx._check(lib.zkfhe_srs_table_info(self.h, ctypes.byref(bits), ctypes.byref(nbytes), ctypes.byref(narrowed)))
        return {"bits": (int(bits[0]), int(bits[1])), "gb": nbytes.value / 2.0 ** 30, "narrowed": bool(narrowed.value)}

    def destroy(self):
        if self.h:
            self.ctx.lib.zkfhe_srs_destroy(self.ctx.h, self.h)
            self.h = None


class Instances:
    """The public inputs of a proof: a read-only sequence of ints over the 32-byte little-endian words the C ABI returns.
    Decoded on demand -- turning 5121 words into Python ints costs more host time than the C side of a k = 13 proof
    spends outside the GPU, and holds the GIL while other proving threads wait for it."""

    __slots__ = ("raw", "_ints")

    def __init__(self, raw):
        self.raw = bytes(raw)
        self._ints = None

    def _decode(self):
        if self._ints is None:
            r = self.raw
            self._ints = [int.from_bytes(r[i:i + 32], "little") for i in range(0, len(r), 32)]
        return self._ints

    def __len__(self):
        return len(self.raw) // 32

    def __iter__(self):
        return iter(self._decode())

    def __getitem__(self, i):
        return self._decode()[i]

    def __eq__(self, other):
        if isinstance(other, Instances):
            return self.raw == other.raw
        try:
            return self._decode() == list(other)
        except TypeError:
            return NotImplemented

    def __repr__(self):
        return "Instances(%d words)" % len(self)


class BfvProvingKey:
    """zkfhe_bfv_keygen: fixed + sigma polynomials, commitments and extended-domain tables resident in HBM."""

    @staticmethod
    def _sigs(lib):
        vp = ctypes.c_void_p
        lib.zkfhe_bfv_pk_save.argtypes = [vp, vp, ctypes.c_char_p]
        lib.zkfhe_bfv_pk_load.argtypes = [vp, vp, ctypes.c_char_p, ctypes.POINTER(vp)]

    @classmethod
    def load(cls, ctx, srs, path, n_poly):
        """zkfhe_bfv_pk_load: a key written by save() (or by `bfv ... keygen`).  n_poly = N of the BFV parameters (sizes the
        instance buffer of prove())."""
        self = cls.__new__(cls)
        cls._sigs(ctx.lib)
        h = ctypes.c_void_p()
        ctx._check(ctx.lib.zkfhe_bfv_pk_load(ctx.h, srs.h, os.fsencode(path), ctypes.byref(h)))
        self.ctx, self.srs, self.params, self.config, self.h = ctx, srs, (int(n_poly), 0, 0, 0), None, h
        self._prove_sigs(ctx.lib)
        return self

    def save(self, path):
        self._sigs(self.ctx.lib)
        self.ctx._check(self.ctx.lib.zkfhe_bfv_pk_save(self.ctx.h, self.h, os.fsencode(path)))

    @staticmethod
    def _prove_sigs(lib):
        vp = ctypes.c_void_p
        lib.zkfhe_bfv_pk_destroy.argtypes = [vp, vp]
        lib.zkfhe_bfv_pk_info.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
        lib.zkfhe_bfv_pk_commitments.argtypes = [vp, ctypes.c_char_p, ctypes.c_char_p]
        lib.zkfhe_bfv_pk_break_points.argtypes = [vp, ctypes.c_int, vp, ctypes.POINTER(ctypes.c_uint32)]
        lib.zkfhe_bfv_prove.argtypes = [vp, vp, vp, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t),
                                        ctypes.c_char_p, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_float)]

    def __init__(self, ctx, srs, input_json_text, params, config, replay=False):
        lib = ctx.lib
        vp = ctypes.c_void_p
        lib.zkfhe_bfv_keygen.argtypes = [vp, vp, ctypes.c_char_p, ctypes.POINTER(BfvParamsC), ctypes.POINTER(BfvConfigC), ctypes.POINTER(vp)]
        lib.zkfhe_bfv_pk_destroy.argtypes = [vp, vp]
        lib.zkfhe_bfv_pk_info.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
        lib.zkfhe_bfv_pk_commitments.argtypes = [vp, ctypes.c_char_p, ctypes.c_char_p]
        lib.zkfhe_bfv_pk_break_points.argtypes = [vp, ctypes.c_int, vp, ctypes.POINTER(ctypes.c_uint32)]
        lib.zkfhe_bfv_prove.argtypes = [vp, vp, vp, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t),
                                        ctypes.c_char_p, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_float)]
        self.ctx, self.srs, self.params, self.config = ctx, srs, params, config
        prm = BfvParamsC(*params)
        cfg = config.to_c(replay)
        h = vp()
        ctx._check(lib.zkfhe_bfv_keygen(ctx.h, srs.h, input_json_text.encode(), ctypes.byref(prm), ctypes.byref(cfg), ctypes.byref(h)))
        self.h = h

    def info(self):
        d = ctypes.create_string_buffer(32)
        nf, ns = ctypes.c_uint32(), ctypes.c_uint32()
        self.ctx.lib.zkfhe_bfv_pk_info(self.h, d, ctypes.byref(nf), ctypes.byref(ns))
        fx = ctypes.create_string_buffer(64 * nf.value)
        sg = ctypes.create_string_buffer(64 * ns.value)
        self.ctx.lib.zkfhe_bfv_pk_commitments(self.h, fx, sg)

        def pts(buf, cnt):
            out = []
            for i in range(cnt):
                x = int.from_bytes(buf.raw[64 * i:64 * i + 32], "little")
                y = int.from_bytes(buf.raw[64 * i + 32:64 * i + 64], "little")
                out.append(None if (x == 0 and y == 0) else (x, y))
            return out
        bps = {}
        for i, name in enumerate(("gate0", "gate1", "rlc")):
            cnt = ctypes.c_uint32(0)
            self.ctx.lib.zkfhe_bfv_pk_break_points(self.h, i, None, ctypes.byref(cnt))
            arr = (ctypes.c_uint32 * max(1, cnt.value))()
            self.ctx.lib.zkfhe_bfv_pk_break_points(self.h, i, arr, ctypes.byref(cnt))
            bps[name] = list(arr)[: cnt.value]
        return {"vk_digest": int.from_bytes(d.raw, "little"), "fixed_commit": pts(fx, nf.value), "sigma_commit": pts(sg, ns.value),
                "break_points": bps}

    def prove(self, input_json_text, seed=None, ctx=None):
        """One proof. `ctx`: the context (stream + workspace) to run on -- several contexts of the same GPU may
        prove concurrently against this key from different threads (ctypes releases the GIL).
        seed: the 32-byte blinding seed.  Zero knowledge rests on it being fresh and secret: None draws os.urandom(32);
        a fixed seed is for reproducible tests.  Shorter seeds are zero-padded, longer ones hashed (seed32())."""
        ctx = ctx or self.ctx
        seed = os.urandom(32) if seed is None else seed32(seed)
        cap = 1 << 18
        buf = ctypes.create_string_buffer(cap)
        plen = ctypes.c_size_t()
        ninst = ctypes.c_size_t(5 * int(self.params[0]) + 8)   # 4 polynomials of N coefficients + cyclo (N + 1) are public
        tm = (ctypes.c_float * 5)()
        text = input_json_text if isinstance(input_json_text, bytes) else input_json_text.encode()
        for _ in range(2):
            ibuf = ctypes.create_string_buffer(32 * ninst.value)
            have = ninst.value
            rc = ctx.lib.zkfhe_bfv_prove(ctx.h, self.srs.h, self.h, text, seed, buf, cap, ctypes.byref(plen), ibuf, ctypes.byref(ninst), tm)
            if rc == 0 or ninst.value <= have:
                break      # otherwise: the library reported the instance count it needs -- retry once with that capacity
        ctx._check(rc)
        return buf.raw[: plen.value], Instances(ibuf.raw[: 32 * ninst.value]), list(tm)

    def witness_stream(self, input_json_text, gamma):
        """zkfhe_bfv_witness_stream: the phase-1 gate-context cells as the GPU generates them, as an (n_cells, 4) uint64 array
        of canonical values (little-endian limbs)."""
        lib = self.ctx.lib
        vp = ctypes.c_void_p
        lib.zkfhe_bfv_witness_stream.argtypes = [vp, vp, ctypes.c_char_p, ctypes.c_char_p, vp, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
        text = input_json_text if isinstance(input_json_text, bytes) else input_json_text.encode()
        g = int(gamma).to_bytes(32, "little")
        n = ctypes.c_size_t()
        self.ctx._check(lib.zkfhe_bfv_witness_stream(self.ctx.h, self.h, text, g, None, 0, ctypes.byref(n)))
        out = np.empty((n.value, 4), dtype=np.uint64)
        self.ctx._check(lib.zkfhe_bfv_witness_stream(self.ctx.h, self.h, text, g, out.ctypes.data_as(vp), n.value, ctypes.byref(n)))
        return out

    def prefix_cache(self, capacity=-1):
        """zkfhe_bfv_pk_prefix_cache: the per-public-key transcript cache of this key (state after vk digest | pk0 | pk1).
        capacity >= 0 sets the number of public keys remembered (0 = off); returns {"hits", "misses", "entries"}."""
        lib = self.ctx.lib
        u64p = ctypes.POINTER(ctypes.c_uint64)
        lib.zkfhe_bfv_pk_prefix_cache.argtypes = [ctypes.c_void_p, ctypes.c_int, u64p, u64p, u64p]
        h, m, e = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        self.ctx._check(lib.zkfhe_bfv_pk_prefix_cache(self.h, int(capacity), ctypes.byref(h), ctypes.byref(m), ctypes.byref(e)))
        return {"hits": h.value, "misses": m.value, "entries": e.value}

    def prehash(self, input_json_text=None):
        """zkfhe_bfv_pk_prehash: announce the input of a LATER proof -- its public inputs are absorbed into a transcript state on a
        helper thread now (host only) and the prove() of the same text starts from it.  One-shot.  Returns {"started", "taken",
        "pending"}; None as text only reads the counters."""
        lib = self.ctx.lib
        u64p = ctypes.POINTER(ctypes.c_uint64)
        lib.zkfhe_bfv_pk_prehash.argtypes = [ctypes.c_void_p, ctypes.c_char_p, u64p, u64p, u64p]
        s, t, p = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        text = None if input_json_text is None else (input_json_text if isinstance(input_json_text, bytes) else input_json_text.encode())
        rc = lib.zkfhe_bfv_pk_prehash(self.h, text, ctypes.byref(s), ctypes.byref(t), ctypes.byref(p))
        if rc != 0:
            raise ZkfheError("zkfhe_bfv_pk_prehash: too many announced proofs are pending (%d)" % rc)
        return {"started": s.value, "taken": t.value, "pending": p.value}

    def export_vk(self):
        lib = self.ctx.lib
        lib.zkfhe_bfv_pk_export_vk.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
        n = ctypes.c_size_t()
        lib.zkfhe_bfv_pk_export_vk(self.h, None, 0, ctypes.byref(n))
        buf = ctypes.create_string_buffer(n.value)
        self.ctx._check(lib.zkfhe_bfv_pk_export_vk(self.h, buf, n.value, ctypes.byref(n)))
        return buf.raw

    def destroy(self):
        if self.h:
            self.ctx.lib.zkfhe_bfv_pk_destroy(self.ctx.h, self.h)
            self.h = None


def make_vk_bytes(k, n_gate0, n_gate1, n_lookup, n_rlc, unusable_rows, lookup_bits, vk_digest, fixed_commit, sigma_commit, transcript="poseidon"):
    """Serialise a verifying key (same layout as zkfhe_bfv_pk_export_vk) from python values; points are (x, y) or None."""
    import struct
    out = b"ZKFHEVK2" + struct.pack("<10I", k, n_gate0, n_gate1, n_lookup, n_rlc, unusable_rows, lookup_bits, TRANSCRIPT_ID[transcript],
                                    len(fixed_commit), len(sigma_commit))
    out += int(vk_digest).to_bytes(32, "little")
    for p in list(fixed_commit) + list(sigma_commit):
        x, y = (0, 0) if p is None else p
        out += int(x).to_bytes(32, "little") + int(y).to_bytes(32, "little")
    return out


def bfv_verify(vk_bytes, instances, proof, srs_seed=b"zkfhe-unsafe-srs", g2=None, s_g2=None):
    """Host-only verifier (C++: transcript replay, quotient identity, one pairing-product check). Returns (accepted, reason).
    g2 / s_g2: the verifier's half of an external SRS, each ((x.c0, x.c1), (y.c0, y.c1)) as ints; default: the seeded setup."""
    lib = load_library()
    if g2 is not None or s_g2 is not None:
        lib.zkfhe_bfv_verify_g2.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t,
                                            ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int), ctypes.c_char_p, ctypes.c_size_t]

        def enc(p):
            (x0, x1), (y0, y1) = p
            return b"".join(int(v).to_bytes(32, "little") for v in (x0, x1, y0, y1))
        inst = instances.raw if isinstance(instances, Instances) else b"".join(int(v).to_bytes(32, "little") for v in instances)
        ok = ctypes.c_int(0)
        err = ctypes.create_string_buffer(256)
        rc = lib.zkfhe_bfv_verify_g2(vk_bytes, len(vk_bytes), inst, len(instances), proof, len(proof), enc(g2), enc(s_g2), ctypes.byref(ok), err, 256)
        if rc != 0:
            raise ZkfheError("zkfhe_bfv_verify_g2: bad arguments (%d)" % rc)
        return bool(ok.value), err.value.decode()
    lib.zkfhe_bfv_verify.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t,
                                     ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int), ctypes.c_char_p, ctypes.c_size_t]
    inst = instances.raw if isinstance(instances, Instances) else b"".join(int(v).to_bytes(32, "little") for v in instances)
    ok = ctypes.c_int(0)
    err = ctypes.create_string_buffer(256)
    rc = lib.zkfhe_bfv_verify(vk_bytes, len(vk_bytes), inst, len(instances), proof, len(proof), bytes(srs_seed), len(srs_seed), ctypes.byref(ok), err, 256)
    if rc != 0:
        raise ZkfheError("zkfhe_bfv_verify: bad arguments (%d)" % rc)
    return bool(ok.value), err.value.decode()
