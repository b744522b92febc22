"""The N > 1 path (independent proofs sharded over ranks, gather of proof bytes, max-over-ranks timing) with two
gloo processes on CPU.  The prover is replaced by a stand-in that hashes its input: the sharding / gathering /
concurrency logic is what is under test here; the GPU prover itself is covered by the -m gpu tests."""
import hashlib
import os
import socket

import torch.distributed as dist
import torch.multiprocessing as mp

import zk_fhe_amd.batch as B


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class FakeKey:
    def prove(self, text, seed, ctx=None):
        return hashlib.sha256(text.encode() + seed + str(ctx).encode()[:0]).digest(), [], [0] * 5


def _worker(rank, world, port, n_items, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    inputs = ["input-%d" % i for i in range(n_items)]
    seeds = [b"seed-%d" % i for i in range(n_items)]
    mine = B.shard_indices(n_items, rank, world)
    proofs = B.prove_batch(FakeKey(), [inputs[i] for i in mine], [seeds[i] for i in mine], contexts=["s0", "s1", "s2"])
    allp = B.gather_proofs(dict(zip(mine, proofs)), n_items, rank, world)
    t = B.max_over_ranks(1.0 + rank)
    if rank == 0:
        out.put((allp, t))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_batch_matches_single_process():
    n_items, world = 11, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in ps:
        p.start()
    allp, t = q.get(timeout=120)
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    want = [hashlib.sha256(("input-%d" % i).encode() + b"seed-%d" % i).digest() for i in range(n_items)]
    assert allp == want            # every proof present, in order, identical to the 1-rank result
    assert t == 2.0                # max over ranks


def test_shard_indices_cover_everything_once():
    for world in (1, 2, 4, 8):
        seen = sorted(i for r in range(world) for i in B.shard_indices(64, r, world))
        assert seen == list(range(64))
        assert all(len(B.shard_indices(64, r, world)) == 64 // world for r in range(world))


def test_run_concurrent_propagates_errors():
    import pytest

    def fn(w, j):
        if j == 3:
            raise ValueError("boom")
        return j
    with pytest.raises(ValueError):
        B.run_concurrent(list(range(8)), ["a", "b"], fn)


def _gather_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    local = bytes([rank]) * 64 * 3          # three 64-byte "points" per rank
    parts = B.all_gather_bytes(local, world)
    if rank == 1:
        out.put(parts)
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_bytes_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    parts = q.get(timeout=120)
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert parts == [bytes([0]) * 192, bytes([1]) * 192]


def test_point_ranges_partition():
    for n, w in ((524288, 8), (8192, 3), (10, 4)):
        rs = [B.point_range(n, r, w) for r in range(w)]
        assert rs[0][0] == 0 and rs[-1][1] == n
        assert all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
