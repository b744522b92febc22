"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads, and exports every symbol
that include/zkfhe.h declares (no compute calls: there is no GPU here)."""
import os
import re

import pytest

import zk_fhe_amd as zk

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "zkfhe.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(zkfhe_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = zk.load_library()
    syms = declared_symbols()
    assert len(syms) >= 30
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(zk.EXPORTS) == syms


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(zk.ZkfheError):
        zk.Context(0)


def test_product_does_not_reference_oracle():
    """The product tree must never import / include / link the oracle."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "zk-fhe_amd")):
        if "_build" in base:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", ".inc")):
                txt = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"(from|import)\s+oracle|oracle/[a-z_]+\.(h|c)\"|liboracle", txt):
                    bad.append(f)
    assert not bad, bad
