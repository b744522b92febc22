"""bench.py as the driver launches it for N > 1 (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`):
two ranks, here sharing GPU 0 with the gloo backend (ZKFHE_BENCH_BACKEND=gloo; on a multi-GPU node the same script runs one
rank per GPU over RCCL).  Checks the contract of the JSON line: one line, from rank 0, n_gpus = 2, the whole-job rate."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.gpu
def test_bench_two_ranks_one_gpu_gloo():
    env = dict(os.environ)
    env["ZKFHE_BENCH_BACKEND"] = "gloo"
    env["ZKFHE_TABLE_GB"] = "4"          # two SRS on one device: the suite's budget, not a service's table profile
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--no-cpu-baseline",
           "--steady-seconds", "0"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["scaling"] == "weak" and d["unit"] == "proofs/s"
    assert d["value"] > 0 and abs(d["value"] - 2 * 6 / (d["ms_per_step"] * 6 / 1e3)) / d["value"] < 1e-6   # whole job: both ranks' proofs over the slowest rank's time
    assert d["vs_baseline"] is None and d["cpu_baseline"] is None
    assert "roofline" in d and d["roofline"]["bound"] == "hbm"
    assert d["config"]["mode"] == "batch" and d["config"]["verified"] is True
    assert len(d["config"]["host_cpu_ms_per_proof_by_rank"]) == 2 and d["config"]["host"]["hash_mode"] in ("latency", "shared")


@pytest.mark.gpu
def test_bench_one_proof_sharded_two_ranks_one_gpu_gloo():
    """bench.py --mode one-proof-sharded (BASELINE configs[4]'s shape, here at k = 13 for time): both ranks make EVERY proof
    together -- sharded SRS, commitments gathered across the ranks (gloo callback here, RCCL on a node), quotient by column --
    and the line says so: strong scaling, value = proofs of the job, not per rank; the timed proofs verify."""
    env = dict(os.environ)
    env["ZKFHE_BENCH_BACKEND"] = "gloo"
    env["ZKFHE_TABLE_GB"] = "4"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
           "--steady-seconds", "0", "--mode", "one-proof-sharded", "--transcript", "blake2b"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["mode"] == "one-proof-sharded" and d["config"]["verified"] is True
    assert abs(d["value"] - 3 / (d["ms_per_step"] * 3 / 1e3)) / d["value"] < 1e-6      # three proofs of the JOB over the wall time
    assert len(d["config"]["host_cpu_ms_per_proof_by_rank"]) == 2 and d["config"]["host"]["usable_cpus"] >= 1


@pytest.mark.gpu
def test_bench_eight_ranks_one_gpu_gloo():
    """The driver's N = 8 launch of bench.py (BASELINE configs[2]: independent proofs, one replica per rank) with the eight ranks
    sharing GPU 0 over gloo: one JSON line from rank 0 with n_gpus = 8, the whole-job rate, one host-CPU figure per rank, and the
    hashing mode batch.configure_host chose from the CPUs each rank can count on (fewer than six per rank: the shared eight-lane
    service).  The rate itself means nothing here (eight ranks on one chip); the contract and the control flow are under test."""
    env = dict(os.environ)
    env["ZKFHE_BENCH_BACKEND"] = "gloo"
    env["ZKFHE_TABLE_GB"] = "1"
    env.pop("ZKFHE_HASH_MODE", None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "8", "--warmup", "1", "--streams", "4",
           "--no-cpu-baseline", "--steady-seconds", "0"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["steps"] == 8 and d["scaling"] == "weak" and d["config"]["mode"] == "batch"
    assert abs(d["value"] - 8 * 8 / (d["ms_per_step"] * 8 / 1e3)) / d["value"] < 1e-6          # 64 proofs of the job over the slowest rank's time
    assert len(d["config"]["host_cpu_ms_per_proof_by_rank"]) == 8 and all(v > 0 for v in d["config"]["host_cpu_ms_per_proof_by_rank"])
    host = d["config"]["host"]
    assert host["cpus_per_rank"] == host["usable_cpus"] / 8
    assert host["hash_mode"] == ("shared" if host["cpus_per_rank"] < 6 else "latency")
    assert d["config"]["verified"] is True and d["vs_baseline"] is None


@pytest.mark.gpu
def test_bench_one_k19_proof_sharded_over_eight_ranks_gloo():
    """bench.py --gpus 8 --mode one-proof-sharded --config k19: BASELINE configs[4] as the driver would launch it on an 8-GPU node
    (there: one rank per GPU, RCCL), here with the eight ranks on GPU 0 over gloo -- control flow, the JSON contract (strong
    scaling: value = proofs of the JOB per second) and a verified k = 19 proof made by eight ranks.  The rate means nothing here."""
    env = dict(os.environ)
    env["ZKFHE_BENCH_BACKEND"] = "gloo"
    env["ZKFHE_TABLE_GB"] = "1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
           "--steady-seconds", "0", "--mode", "one-proof-sharded", "--config", "k19", "--transcript", "blake2b"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=2400)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["config"]["mode"] == "one-proof-sharded" and d["config"]["verified"] is True
    assert d["metric"] == "BFV proofs/sec (k=19)" and "k=19, N=16384" in d["config"]["workload"]
    assert abs(d["value"] - 1 / (d["ms_per_step"] / 1e3)) / d["value"] < 1e-6            # one proof of the JOB per step
    assert len(d["config"]["host_cpu_ms_per_proof_by_rank"]) == 8
