"""The reference's SRS and its on-disk form (params/kzg_bn254_<k>.srs, README.md:34, .gitignore:17) and the
data/<name>.snark container (8(f) row 3).

  * the ChaCha20 block function (library and oracle) against PUBLISHED vectors: the zero-key keystream block 0 and RFC 7539
    section 2.3.2 -- this is what pins `s` of ParamsKZG::setup(k, ChaCha20Rng::from_seed([0; 32]));
  * GPU: an SRS derived that way has g[1] = s G and the oracle's s G2; saved, the file has halo2's RawBytes frame; loaded, it
    gives the same key and the same proof bytes; the verifier accepts the proof with the G2 tail read from the file alone;
  * the snark container: the instances / proof fields exactly as bincode lays them out (Montgomery limbs), round trip, the
    previous container still readable, malformed input refused."""
import json
import os
import struct

import numpy as np
import pytest

import zk_fhe_amd as zk
from oracle import chacha20_ref as CH
from oracle import pyref

R = pyref.R
# draft-agl-tls-chacha20poly1305 / RFC 7539 A.1 test vector #1: key = 0, nonce = 0, block counter = 0
ZERO_KEY_BLOCK0 = bytes.fromhex("76b8e0ada0f13d90405d6ae55386bd28bdd219b8a08ded1aa836efcc8b770dc7"
                                "da41597c5157488d7724e03fb8d84a376a43b8f41518a11cc387b669b2ee6586")
# RFC 7539 section 2.3.2: key 00..1f, counter 1, nonce 00:00:00:09:00:00:00:4a:00:00:00:00
RFC_232 = bytes.fromhex("10f1e7e4d13b5915500fdd1fa32071c4c7d1f4c733c068030422aa9ac3d46c4e"
                        "d2826446079faa0914c2d705d98b02a2b5129cd1de164eb9cbd083e8a2503c4e")


def test_chacha20_block_matches_published_vectors():
    for impl in (CH.block, zk.chacha20_block):
        assert impl(bytes(32), (0, 0, 0, 0)) == ZERO_KEY_BLOCK0
        assert impl(bytes(range(32)), (1, 0x09000000, 0x4A000000, 0)) == RFC_232
    # the reference's secret: Fr::from_u512 of those 64 bytes
    s = int.from_bytes(ZERO_KEY_BLOCK0, "little") % R
    assert CH.reference_srs_secret() == s


def test_snark_container_layout_and_round_trip():
    inst = [0, 1, R - 1, 123456789 << 200]
    proof = bytes(range(200)) * 3
    blob = zk.snark_encode(inst, proof)
    assert blob[:8] == b"ZKFHESN2" and struct.unpack_from("<Q", blob, 8)[0] == 0      # protocol absent
    # from byte 16 on: what bincode writes for (Vec<Vec<Fr>>, Vec<u8>) with halo2curves' serde (raw Montgomery limbs)
    want = struct.pack("<QQ", 1, len(inst)) + b"".join((v * (1 << 256) % R).to_bytes(32, "little") for v in inst) + struct.pack("<Q", len(proof)) + proof
    assert blob[16:] == want
    assert zk.snark_decode(blob) == (inst, proof)
    old = b"ZKFHESN1" + struct.pack("<Q", len(inst)) + b"".join(v.to_bytes(32, "little") for v in inst) + proof
    assert zk.snark_decode(old) == (inst, proof)
    for bad in (blob[:-1], blob + b"x", b"ZKFHESN3" + blob[8:], blob[:40], b""):
        with pytest.raises(zk.ZkfheError):
            zk.snark_decode(bad)
    # an instance limb pattern that is not a reduced residue is refused
    evil = bytearray(blob)
    evil[32:64] = b"\xff" * 32
    with pytest.raises(zk.ZkfheError):
        zk.snark_decode(bytes(evil))
    with pytest.raises(zk.ZkfheError):
        zk.snark_encode([R], b"")


@pytest.mark.gpu
def test_reference_srs_derivation_file_and_reload(tmp_path):
    import torch  # noqa: F401
    from oracle import binding as orc
    from oracle import circuit_ref as C
    from oracle import halo2_ref as H
    from oracle import pairing_ref as PR
    from tests.test_proof_oracle import synth_input
    ctx = zk.Context(0)
    k, n = 9, 512
    s = CH.reference_srs_secret()
    srs = zk.Srs(ctx, k, seed=zk.SRS_HALO2_UNSAFE)
    path = str(tmp_path / ("kzg_bn254_%d.srs" % k))
    srs.save(path)
    blob = open(path, "rb").read()
    assert len(blob) == 4 + 2 * 64 * n + 256 and struct.unpack_from("<I", blob)[0] == k
    g = np.frombuffer(blob, dtype=np.uint64, count=8 * n, offset=4).reshape(n, 8)
    gl = np.frombuffer(blob, dtype=np.uint64, count=8 * n, offset=4 + 64 * n).reshape(n, 8)
    srs_o = H.make_srs(k, seed=H.SRS_HALO2_UNSAFE)       # the oracle's setup from the oracle's ChaCha20
    assert srs_o["s"] == s
    assert np.array_equal(g, srs_o["g"]) and np.array_equal(gl, srs_o["g_lagrange"])
    gen = orc.arr_to_points(g[:2])
    assert gen[0] == pyref.G1_GEN and gen[1] == pyref.g1_mul(pyref.G1_GEN, s)
    # G2 tail: raw Montgomery coordinates of the generator and of s G2
    mont = lambda v: (v * (1 << 256)) % pyref.Q    # noqa: E731
    t0 = len(blob) - 256
    tail = [int.from_bytes(blob[t0 + 32 * i:t0 + 32 * i + 32], "little") for i in range(8)]
    sg2 = PR.ec_mul(PR.G2_GEN, s)
    flat = lambda p: [int(p[0].c[0]), int(p[0].c[1]), int(p[1].c[0]), int(p[1].c[1])]   # noqa: E731
    assert tail == [mont(v) for v in flat(PR.G2_GEN) + flat(sg2)]
    fk, g2_f, sg2_f = zk.srs_file_g2(path)
    assert fk == k and (g2_f, sg2_f) == srs.g2()
    assert [sg2_f[0][0], sg2_f[0][1], sg2_f[1][0], sg2_f[1][1]] == flat(sg2)
    # the loaded SRS proves like the derived one, and the proof verifies against the file's G2 half only
    prm = C.BfvParams(N=8)
    inp = synth_input(8, prm.Q, prm.T, prm.B, 1)
    hcfg = H.auto_config(9, 9, H.BfvCircuit(inp, prm))
    zcfg = zk.BfvConfig(9, hcfg.n_gate0, hcfg.n_gate1, hcfg.n_lookup, hcfg.n_rlc, 9)
    out = []
    loaded = zk.Srs.load(ctx, path)
    for S in (srs, loaded):
        pk = zk.BfvProvingKey(ctx, S, json.dumps(inp), (8, prm.Q, prm.T, prm.B), zcfg)
        proof, inst, _ = pk.prove(json.dumps(inp), b"file")
        out.append((pk.info(), proof))
        vk = pk.export_vk()
        pk.destroy()
    assert out[0] == out[1]
    ok, why = zk.bfv_verify(vk, inst, out[0][1], g2=g2_f, s_g2=sg2_f)
    assert ok, why
    ok, _ = zk.bfv_verify(vk, inst, out[0][1], srs_seed=zk.SRS_HALO2_UNSAFE)
    assert ok
    ok, _ = zk.bfv_verify(vk, inst, out[0][1])            # the suite's other seed: a different s
    assert not ok
    loaded.save(str(tmp_path / "again.srs"))
    assert open(tmp_path / "again.srs", "rb").read() == blob
    # damaged files are refused: truncated, a point off the curve, a wrong k
    for name, data in (("short", blob[:-7]), ("offcurve", blob[:4 + 64 * 3 + 5] + bytes([blob[4 + 64 * 3 + 5] ^ 1]) + blob[4 + 64 * 3 + 6:]),
                       ("k", struct.pack("<I", k + 1) + blob[4:])):
        p = str(tmp_path / (name + ".srs"))
        open(p, "wb").write(data)
        with pytest.raises(zk.ZkfheError):
            zk.Srs.load(ctx, p)
    srs.destroy()
    loaded.destroy()
    ctx.close()
