"""Soak test (MI355X box): many concurrent k = 13 proofs over distinct inputs and seeds, every one checked by the C++
verifier (pairing) and the repeated (input, seed) pairs compared byte for byte.  Exercises the concurrent paths (several
contexts on one key, work stealing in the table sum, pinned result blocks) far longer than the unit tests do.

    python tools/soak.py [--proofs 2000] [--streams 16] [--transcript blake2b|poseidon]
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--proofs", type=int, default=2000)
    ap.add_argument("--streams", type=int, default=16)
    ap.add_argument("--transcript", default="blake2b")
    ap.add_argument("--inputs", type=int, default=12)
    ap.add_argument("--config", default="k13", choices=["k13", "k16"], help="k16: N = 4096, Q = 2^60 - 93 (auto-configured columns)")
    ap.add_argument("--hash-mode", default="latency", choices=["latency", "shared"], help="shared: the transcripts' long runs through the eight-lane Poseidon service")
    ap.add_argument("--gate", type=int, default=0, help="admission gate of the heavy middle of a proof (zkfhe_prover_gate)")
    ap.add_argument("--announce", type=int, default=0, help="announce every job's input this many jobs ahead (zkfhe_bfv_pk_prehash; 0 = off, at most 16 - streams): the "
                                                            "helper threads, the one-shot states and the per-key prefix cache under concurrency")
    args = ap.parse_args()
    import zk_fhe_amd as zk
    zk.host_hash_mode(args.hash_mode)
    zk.prover_gate(args.gate)
    import zk_fhe_amd.batch as batch
    from zk_fhe_amd import inputs as gen
    T, B = 7, 19
    ctx = zk.Context(0)
    if args.config == "k13":
        N, Q, k = 1024, 536870909, 13
        cfgj = json.load(open(os.path.join(ROOT, "tests", "golden", "bfv", "bfv_config.json")))
        zcfg = zk.BfvConfig.from_pinning(cfgj, transcript=args.transcript)
        ins = [open(os.path.join(ROOT, "tests", "golden", "bfv", "bfv.in"), "rb").read()]
    else:
        N, Q, k = 4096, (1 << 60) - 93, 16
        ins = []
    ins += [json.dumps(gen.generate(N, Q, T, B, seed=777 + i)).encode() for i in range(args.inputs - len(ins))]
    if args.config != "k13":
        zcfg = zk.bfv_auto_config(ins[0], (N, Q, T, B), k, transcript=args.transcript)
    srs = zk.Srs(ctx, k)
    empty = json.dumps(gen.empty(N))
    pk = zk.BfvProvingKey(ctx, srs, empty, (N, Q, T, B), zcfg, replay=args.config == "k13")
    vk = pk.export_vk()
    ctxs = [ctx] + [zk.Context(0) for _ in range(args.streams - 1)]
    first = {}
    lock = threading.Lock()
    bad = []
    done = [0]

    def one(c, j):
        # every third job repeats an earlier (input, seed) pair: the bytes must repeat too
        key = (j % len(ins), j if j % 3 else j // 3 % 50)
        if args.announce:
            try:
                pk.prehash(ins[(j + args.announce) % len(ins)])
            except zk.ZkfheError:
                pass   # sixteen announcements pending: this job's successor hashes for itself
        proof, inst, _ = pk.prove(ins[key[0]], b"soak-%d" % key[1], ctx=c)
        ok, why = zk.bfv_verify(vk, inst, proof)
        with lock:
            done[0] += 1
            if not ok:
                bad.append((j, why))
            if key in first and first[key] != proof:
                bad.append((j, "bytes differ from the first proof of the same input and seed"))
            first.setdefault(key, proof)

    t0 = time.time()
    batch.run_concurrent(list(range(args.proofs)), ctxs, one)
    dt = time.time() - t0
    print(json.dumps({"proofs": done[0], "failed": len(bad), "first_failures": bad[:5], "seconds": round(dt, 1),
                      "distinct_pairs": len(first), "streams": args.streams, "transcript": args.transcript,
                      "hash_mode": zk.host_hash_mode(), "gate": args.gate, "announce": args.announce, "prehash": pk.prehash(), "table_bits": srs.table_bits()[0]}))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
