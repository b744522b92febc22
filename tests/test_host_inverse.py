"""fp_inv (safegcd division steps, zk-fhe_amd/csrc/bn254.hip.hpp) against the binary-Euclid inversion and against
a * a^-1 == 1, for Fr and Fq: 200 k random values plus powers of two, p - 1, p - 2 and short values.  The field code is
host+device; this runs its host instantiation (no GPU)."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_safegcd_inverse_matches_euclid(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = str(tmp_path / "inverse_check")
    subprocess.run([hipcc, "-O2", "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "zk-fhe_amd", "csrc"),
                    os.path.join(HERE, "native", "inverse_check.hip"), "-o", exe], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert "Fr: 0 bad" in out and "Fq: 0 bad" in out, out


def test_radix29_field_and_point_arithmetic(tmp_path):
    """csrc/fq29.hip.hpp (nine 29-bit limbs, Montgomery constant 2^261, lazy reduction -- the arithmetic of the MSM kernels) against
    the standard 8 x 32-bit arithmetic: products, fused products with operands up to 11 p, weak / canonical reduction, the
    zero test on differences, pack / unpack, and mixed / full additions and doublings including the doubling-through-addition
    and cancellation cases.  Host instantiation of the same host+device code (no GPU)."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = str(tmp_path / "f29_check")
    subprocess.run([hipcc, "-O2", "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "zk-fhe_amd", "csrc"),
                    "-I", os.path.join(ROOT, "include"), os.path.join(HERE, "native", "f29_check.hip"), "-o", exe], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert "fq29: 0 bad" in out, out


def test_lazy_signed_limb_arithmetic(tmp_path):
    """csrc/lz29.hip.hpp (the arithmetic of the 2^13 NTT tile: signed lazy limbs with compile-time bounds, products against unpacked
    twiddles, weak reduction of values of either sign, canonical store) against the standard 8 x 32-bit Fr arithmetic.  Host
    instantiation of the same host+device code (no GPU)."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = str(tmp_path / "lz29_check")
    subprocess.run([hipcc, "-O2", "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "zk-fhe_amd", "csrc"),
                    "-I", os.path.join(ROOT, "include"), os.path.join(HERE, "native", "lz29_check.hip"), "-o", exe], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert "lz29: 0 bad" in out, out


def test_fr_nine_limb_sums_of_the_element_wise_kernels(tmp_path):
    """csrc/fr29.hip.hpp as k_eval_jobs / k_lincomb_ptrs / k_quotient_combine and the witness cell writer use it (round 4):
    fr29_to_mont against fp_to_mont, and lazy sums with 2^261-form constants, two terms per reduction, against the standard
    8 x 32-bit arithmetic -- 1 to 97 terms, short, negative-short and full-width operands.  Host instantiation (no GPU)."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = str(tmp_path / "fr29_sum_check")
    subprocess.run([hipcc, "-O2", "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "zk-fhe_amd", "csrc"),
                    "-I", os.path.join(ROOT, "include"), os.path.join(HERE, "native", "fr29_sum_check.hip"), "-o", exe], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert "fr29 sums: 0 bad" in out, out


def test_xyzz_batch_normalisation_on_the_host():
    """zkfhe_g1_xyzz_to_affine (host only: Montgomery's trick over the ZZZ of the array, one Bernstein-Yang inversion; the prover
    normalises a round's commitments with it) against the big-integer oracle: random multiples of the generator, each blown up
    to an accumulator form with its own random Z (X = x Z^2, Y = y Z^3, ZZ = Z^2, ZZZ = Z^3), identities (ZZ = 0) in between,
    n = 1 and n = 0 included."""
    import ctypes
    import random

    import numpy as np

    import zk_fhe_amd as zk
    from oracle import pyref
    lib = zk.load_library()
    lib.zkfhe_g1_xyzz_to_affine.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    rnd = random.Random(5)
    Q = pyref.Q

    def limbs(v):
        m = pyref.to_mont(v, Q)
        return [(m >> (64 * j)) & ((1 << 64) - 1) for j in range(4)]
    for n in (0, 1, 2, 37):
        pts, rows = [], []
        for i in range(n):
            if n > 2 and i % 5 == 3:
                pts.append(None)
                rows.append(limbs(rnd.randrange(Q)) + limbs(rnd.randrange(Q)) + [0] * 8)      # identity: only ZZ = 0 matters
                continue
            P = pyref.g1_mul((1, 2), rnd.randrange(1, pyref.R))
            z = rnd.randrange(1, Q)
            zz, zzz = z * z % Q, z * z * z % Q
            pts.append(P)
            rows.append(limbs(P[0] * zz % Q) + limbs(P[1] * zzz % Q) + limbs(zz) + limbs(zzz))
        raw = np.array(rows, dtype=np.uint64).reshape(n, 16)
        out = np.full((n, 8), 0xFFFFFFFFFFFFFFFF, dtype=np.uint64)
        assert lib.zkfhe_g1_xyzz_to_affine(raw.ctypes.data_as(ctypes.c_void_p), n, out.ctypes.data_as(ctypes.c_void_p)) == 0
        for i, P in enumerate(pts):
            want = [0] * 8 if P is None else limbs(P[0]) + limbs(P[1])
            assert [int(v) for v in out[i]] == want, (n, i)
    assert lib.zkfhe_g1_xyzz_to_affine(None, 3, None) != 0
