"""Batches of independent proofs over the GPUs of one node (BASELINE config 3).

Proofs are independent objects: proof i goes to rank i mod W (one process per GPU), inside a rank the proofs run
concurrently on several contexts (HIP streams) against one proving key.  There is NO data-path collective; the only
communication is collecting the <= 62 KB proofs (all_gather_object) and the max-over-ranks of the wall time.
Works with the gloo backend on CPU (tests) and nccl (= RCCL) on GPUs.
"""
import os
import threading
import time


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


# A k = 13 proof costs 28 ms of host CPU with every Poseidon transcript hashing on its own and ~16 ms through the shared
# eight-lane service, which in turn doubles the hashing latency of a proof.  One GPU proves ~220 proofs/s,
# i.e. keeps ~6 cores busy in the first mode: with fewer CPUs per rank than that the host is the bottleneck and the service pays.
CPUS_PER_GPU_FOR_LATENCY_MODE = 6


def configure_host(zk, ranks_on_this_host=1):
    """Picks the transcript hashing mode for a batch run from the CPUs each rank can count on (unless ZKFHE_HASH_MODE is set);
    returns {"usable_cpus", "cpus_per_rank", "hash_mode"}."""
    cpus = usable_cpus()
    per = cpus / max(1, ranks_on_this_host)
    if "ZKFHE_HASH_MODE" not in os.environ:
        zk.host_hash_mode("shared" if per < CPUS_PER_GPU_FOR_LATENCY_MODE else "latency")
    return {"usable_cpus": cpus, "cpus_per_rank": per, "hash_mode": zk.host_hash_mode()}


def shard_indices(n_items, rank, world):
    """proof i -> rank i mod world"""
    return list(range(rank, n_items, world))


def run_concurrent(jobs, workers, fn, stagger_s=0.0):
    """Run fn(worker, job) for every job, `len(workers)` at a time (one thread per worker, jobs pulled from a shared
    counter).  Returns results in job order; the first exception is re-raised.  stagger_s > 0 starts worker i that many
    seconds after worker i - 1: proofs that start together reach every Fiat-Shamir round together and leave the GPU idle
    while all of them hash; a small offset keeps some of them in a kernel phase while others are on the host."""
    results = [None] * len(jobs)
    lock = threading.Lock()
    nxt = [0]
    errs = []

    def loop(w, delay=0.0):
        if delay > 0:
            time.sleep(delay)
        while True:
            with lock:
                j = nxt[0]
                if j >= len(jobs) or errs:
                    return
                nxt[0] += 1
            try:
                results[j] = fn(w, jobs[j])
            except Exception as e:  # noqa: BLE001
                with lock:
                    errs.append(e)
                return
    ths = [threading.Thread(target=loop, args=(w, i * stagger_s)) for i, w in enumerate(workers)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    if errs:
        raise errs[0]
    return results


def prove_batch(pk, inputs, seeds, contexts, with_instances=False):
    """All proofs of this rank: inputs[i] (JSON text), seeds[i]; `contexts` = zk.Context objects of this rank's GPU (one proof in
    flight per context).  Returns the proof bytes in input order -- (proof, public inputs) pairs with with_instances."""
    if with_instances:
        return run_concurrent(list(range(len(inputs))), contexts, lambda c, i: pk.prove(inputs[i], seeds[i], ctx=c)[:2])
    return run_concurrent(list(range(len(inputs))), contexts, lambda c, i: pk.prove(inputs[i], seeds[i], ctx=c)[0])


def gather_proofs(local, n_items, rank, world):
    """local: {global index: proof bytes} of this rank -> list of all n_items proofs (on every rank)."""
    import torch.distributed as dist
    if world == 1 and not dist.is_initialized():
        return [local[i] for i in range(n_items)]
    parts = [None] * world
    dist.all_gather_object(parts, local)
    merged = {}
    for p in parts:
        merged.update(p)
    missing = [i for i in range(n_items) if i not in merged]
    if missing:
        raise RuntimeError("proofs missing after gather: %s" % missing[:8])
    return [merged[i] for i in range(n_items)]


def max_over_ranks(seconds, device=None):
    """Wall time of the slowest rank (what bench.py reports)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():      # a one-rank process GROUP still runs the collective (bench.py ZKFHE_BENCH_FORCE_DIST)
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_floats(value, device=None):
    """one float per rank -> list over ranks (on every rank); what bench.py reports per rank (host CPU per proof)"""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        return [float(value)]
    t = torch.tensor([value], dtype=torch.float64, device=device or "cpu")
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


# ----------------------------------------------------------------------------------------------------------------
# One huge MSM split over the GPUs of a node (BASELINE config 5): rank r owns the point range
# [r*n/W, (r+1)*n/W) of the basis (its own per-window table) and the matching scalars; a partial MSM is a group
# element, so  sum_r partial_r  is the MSM.  EC addition is not an RCCL reduction op: the 64-byte affine partials
# are all-gathered as raw bytes (n_cols * 64 B per rank -- latency-bound on xGMI) and every rank adds them locally.
def point_range(n, rank, world):
    lo = (n * rank) // world
    hi = (n * (rank + 1)) // world
    return lo, hi


def all_gather_bytes(local, world, device=None):
    """local: bytes of equal length on every rank -> list of `world` byte strings (uint8 all_gather; gloo or nccl)."""
    if world == 1:
        return [bytes(local)]
    import torch
    import torch.distributed as dist
    t = torch.frombuffer(bytearray(local), dtype=torch.uint8)
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return [bytes(o.cpu().numpy().tobytes()) for o in out]


class ShardedMsm:
    """basis_slice: zk.Basis built from this rank's point range; combine() adds the gathered partials on the GPU."""

    def __init__(self, ctx, basis_slice, rank, world, device=None):
        self.ctx, self.basis, self.rank, self.world, self.device = ctx, basis_slice, rank, world, device

    def msm(self, scalars_slice):
        """scalars_slice: (n_cols, n_local, 4) Montgomery Fr of this rank's range -> (n_cols, 8) affine result (every rank)."""
        import numpy as np
        part = self.ctx.msm(self.basis, scalars_slice)
        parts = all_gather_bytes(part.tobytes(), self.world, self.device)
        acc = np.frombuffer(parts[0], dtype=np.uint64).reshape(-1, 8).copy()
        for p in parts[1:]:
            acc = self.ctx.g1_add(acc, np.frombuffer(p, dtype=np.uint64).reshape(-1, 8))
        return acc
