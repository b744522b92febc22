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
