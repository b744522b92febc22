"""The `bfv` command-line driver with the reference's surface (README.md:18-52):
    bfv --name bfv -k 13 --input bfv/bfv.in {mock | keygen | prove | verify}
run as a subprocess in a scratch directory laid out like the reference's working tree (data/, configs/).
`mock` needs no GPU; keygen / prove / verify are one GPU-marked walk through README.md:28-52."""
import json
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
G = os.path.join(HERE, "golden", "bfv")
EXE = os.path.join(ROOT, "zk-fhe_amd", "bfv")


def _tree(tmp_path, with_pinning):
    os.makedirs(tmp_path / "data" / "bfv")
    os.makedirs(tmp_path / "configs")
    shutil.copy(os.path.join(G, "bfv.in"), tmp_path / "data" / "bfv" / "bfv.in")
    shutil.copy(os.path.join(G, "bfv_empty.in"), tmp_path / "data" / "bfv" / "bfv_empty.in")
    if with_pinning:
        shutil.copy(os.path.join(G, "bfv_config.json"), tmp_path / "configs" / "bfv.json")


def _run(tmp_path, *args):
    if not os.path.exists(EXE):
        pytest.skip("zk-fhe_amd/bfv is not built (python -c 'import __graft_entry__ as g; g.build()')")
    return subprocess.run([EXE, "--name", "bfv", "-k", "13"] + list(args), cwd=str(tmp_path), capture_output=True, text=True, timeout=900)


def test_mock_checks_the_reference_input(tmp_path):
    """README.md:18-22.  Host only: the circuit's asserts plus every gate / lookup / copy constraint row by row."""
    _tree(tmp_path, with_pinning=True)
    r = _run(tmp_path, "--input", "bfv/bfv.in", "mock")
    assert r.returncode == 0, r.stderr
    assert "every gate, lookup and copy constraint holds" in r.stdout
    # a wrong ciphertext coefficient: the mock prover reports the violated constraint and exits non-zero
    inp = json.load(open(tmp_path / "data" / "bfv" / "bfv.in"))
    inp["c1"][5] = str((int(inp["c1"][5]) + 1) % 536870909)
    json.dump(inp, open(tmp_path / "data" / "bfv" / "bad.in", "w"))
    r = _run(tmp_path, "--input", "bfv/bad.in", "mock")
    assert r.returncode != 0 and "NOT satisfied" in r.stderr
    # missing pinning: told to run keygen first
    os.remove(tmp_path / "configs" / "bfv.json")
    r = _run(tmp_path, "--input", "bfv/bfv.in", "mock")
    assert r.returncode != 0 and "keygen" in r.stderr


@pytest.mark.gpu
def test_keygen_prove_verify_walkthrough(tmp_path):
    """README.md:28-52 on the reference's own files: keygen from bfv_empty.in writes configs/bfv.json (equal to the reference's
    pinned file in params and break points), data/bfv.pk and data/bfv.vk; prove writes data/bfv.snark; verify accepts it and
    rejects a corrupted snark, a snark with one public input changed, and a foreign vk."""
    _tree(tmp_path, with_pinning=False)
    r = _run(tmp_path, "--input", "bfv/bfv_empty.in", "keygen")
    assert r.returncode == 0, r.stderr
    got = json.load(open(tmp_path / "configs" / "bfv.json"))
    want = json.load(open(os.path.join(G, "bfv_config.json")))
    assert got["params"] == want["params"] and got["break_points"] == want["break_points"]
    assert os.path.getsize(tmp_path / "data" / "bfv.pk") > 50 << 20 and os.path.exists(tmp_path / "data" / "bfv.vk")
    # README.md:34 / .gitignore:17: the unsafe test setup lands in params/kzg_bn254_13.srs (halo2 RawBytes frame) and is the
    # reference's own: g[1] = s G with s from ChaCha20Rng::from_seed([0; 32]) (tests/test_srs_file.py pins the keystream)
    srs_path = tmp_path / "params" / "kzg_bn254_13.srs"
    assert "kzg_bn254_13.srs" in r.stdout and os.path.getsize(srs_path) == 4 + 2 * 64 * 8192 + 256
    srs_blob = open(srs_path, "rb").read()
    from oracle import chacha20_ref, pyref
    to_int = lambda b: int.from_bytes(b, "little") * pow(1 << 256, -1, pyref.Q) % pyref.Q   # noqa: E731  raw Montgomery limbs
    assert (to_int(srs_blob[68:100]), to_int(srs_blob[100:132])) == pyref.g1_mul(pyref.G1_GEN, chacha20_ref.reference_srs_secret())
    stamp = os.stat(srs_path).st_mtime_ns
    r = _run(tmp_path, "--input", "bfv/bfv.in", "prove")
    assert r.returncode == 0 and "Proving time" in r.stdout, r.stderr
    assert "wrote" not in r.stdout and os.stat(srs_path).st_mtime_ns == stamp and open(srs_path, "rb").read() == srs_blob   # read, not rebuilt
    snark = open(tmp_path / "data" / "bfv.snark", "rb").read()
    assert snark[:8] == b"ZKFHESN2"
    r = _run(tmp_path, "--input", "bfv/bfv.in", "verify")
    assert r.returncode == 0 and "Snark verified successfully" in r.stdout, r.stderr
    # two proofs of the same statement differ (fresh blinding seed from the OS) and both verify
    r = _run(tmp_path, "--input", "bfv/bfv.in", "prove")
    assert r.returncode == 0
    assert open(tmp_path / "data" / "bfv.snark", "rb").read() != snark
    assert _run(tmp_path, "--input", "bfv/bfv.in", "verify").returncode == 0
    for pos in (len(snark) - 10, 32 + 32 * 100 + 3, len(snark) // 2):      # an opening point, a public input, an evaluation
        bad = bytearray(snark)
        bad[pos] ^= 1
        open(tmp_path / "data" / "bfv.snark", "wb").write(bytes(bad))
        r = _run(tmp_path, "--input", "bfv/bfv.in", "verify")
        assert r.returncode != 0 and "FAILED" in r.stderr
    open(tmp_path / "data" / "bfv.snark", "wb").write(snark)
    vk = bytearray(open(tmp_path / "data" / "bfv.vk", "rb").read())
    vk[200] ^= 1
    open(tmp_path / "data" / "bfv.vk", "wb").write(bytes(vk))
    assert _run(tmp_path, "--input", "bfv/bfv.in", "verify").returncode != 0
    # verify needs only the G2 tail of the params file; without the file it derives the same setup; with another setup's file it rejects
    open(tmp_path / "data" / "bfv.vk", "wb").write(bytes(vk[:200]) + bytes([vk[200] ^ 1]) + bytes(vk[201:]))
    assert _run(tmp_path, "--input", "bfv/bfv.in", "verify").returncode == 0
    os.rename(srs_path, str(srs_path) + ".away")
    assert _run(tmp_path, "--input", "bfv/bfv.in", "verify").returncode == 0
    tail = bytearray(srs_blob)
    tail[-128:] = tail[-256:-128]                     # s G2 := G2, i.e. s = 1
    open(srs_path, "wb").write(bytes(tail))
    assert _run(tmp_path, "--input", "bfv/bfv.in", "verify").returncode != 0
    # a damaged params file stops prove instead of being silently replaced
    open(srs_path, "wb").write(srs_blob[:-9])
    r = _run(tmp_path, "--input", "bfv/bfv.in", "prove")
    assert r.returncode != 0 and "params" in r.stderr
    os.remove(srs_path)
    os.rename(str(srs_path) + ".away", srs_path)
    # a wrong witness is refused by prove with a non-zero exit code
    inp = json.load(open(tmp_path / "data" / "bfv" / "bfv.in"))
    inp["e0"][0] = "25"
    json.dump(inp, open(tmp_path / "data" / "bfv" / "bad.in", "w"))
    assert _run(tmp_path, "--input", "bfv/bad.in", "prove").returncode != 0
