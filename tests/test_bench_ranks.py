"""bench.py for N > 1, both ways it can be started: plainly (`python bench.py --gpus N`: the script launches itself again under
torch.distributed.run with N local ranks) and the way the task's contract words it (`python -m torch.distributed.run
--nproc-per-node N bench.py --gpus N ...`).  The ranks share GPU 0 here with the gloo backend (ZKFHE_BENCH_BACKEND=gloo; on a
multi-GPU node the same script runs one rank per GPU over RCCL); the RCCL branch itself runs as a ONE-rank process group
(ZKFHE_BENCH_FORCE_DIST=1).  Checks the contract of the JSON line: one line, from rank 0, n_gpus = the N asked for, the whole-job
rate -- and that a launch whose size is not the N asked for prints no line at all."""
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


def _clean_env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_PORT", "ZKFHE_BENCH_FORCE_DIST")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(kw)
    return env


@pytest.mark.gpu
def test_bench_plain_command_with_gpus_2_runs_two_ranks():
    """`python bench.py --gpus 2` with NO launcher around it must be a two-rank job (the script starts itself
    again under torch.distributed.run) and print n_gpus = 2 -- not a single-GPU line that ignores the flag."""
    env = _clean_env(ZKFHE_BENCH_BACKEND="gloo", ZKFHE_TABLE_GB="4")          # two SRS on one device: the suite's budget, not a service's table profile
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--steady-seconds", "0"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["gpus_requested"] == 2 and d["steps"] == 6 and d["scaling"] == "weak" and d["unit"] == "proofs/s"
    assert d["config"]["process_group"] == "gloo" and d["config"]["proofs_gathered_and_verified"] == 2   # each rank's last proof, verified on rank 0
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
    env = _clean_env(ZKFHE_BENCH_BACKEND="gloo", ZKFHE_TABLE_GB="4")
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
    env = _clean_env(ZKFHE_BENCH_BACKEND="gloo", ZKFHE_TABLE_GB="1")
    env.pop("ZKFHE_HASH_MODE", None)
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
    env = _clean_env(ZKFHE_BENCH_BACKEND="gloo", ZKFHE_TABLE_GB="1")
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
    assert d["config"]["sharded_proof_identical_on_all_ranks"] is True                   # all eight ranks ended with the same bytes
    # ... and they are the SINGLE-GPU proof of the same input and seed, made here after the ranks have given the device back (eight
    # k = 19 keys and workspaces next to a ninth do not fit one device): the sharded path -- an eighth of both SRS halves per rank,
    # commitments by point range, quotient / evaluations / SHPLONK sums by column and index -- changes no byte
    import hashlib
    import torch  # noqa: F401
    import zk_fhe_amd as zk
    from zk_fhe_amd import inputs as gen
    lp = d["config"]["last_timed_proof"]
    N, Q = 16384, (1 << 60) - 93
    text = json.dumps(gen.generate(N, Q, 7, 19, seed=20240613 + lp["input_index"])).encode()    # bench.py main(): inputs[i] of rank 0
    ctx = zk.Context(0)
    srs = zk.Srs(ctx, 19)
    pk = zk.BfvProvingKey(ctx, srs, json.dumps(gen.empty(N)), (N, Q, 7, 19), zk.bfv_auto_config(text, (N, Q, 7, 19), 19, transcript="blake2b"), replay=False)
    assert "%064x" % pk.info()["vk_digest"] == lp["vk_digest"]
    proof, inst, _ = pk.prove(text, lp["seed"].encode())
    ok, why = zk.bfv_verify(pk.export_vk(), inst, proof)
    pk.destroy()
    srs.destroy()
    ctx.close()
    assert ok, why
    assert len(inst) == 5 * N + 1 and hashlib.sha256(proof).hexdigest() == lp["sha256"]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["batch", "one-proof-sharded"])
def test_bench_rccl_process_group_of_one_rank(mode):
    """The RCCL branch of bench.py on the hardware there is: ZKFHE_BENCH_FORCE_DIST=1 makes a ONE-rank job build its process group
    (backend "nccl" = RCCL, device_id = cuda:0) and send every collective of the N > 1 path through it -- the barriers, the
    max-over-ranks of the clock (all_reduce on a device tensor), the per-rank host-CPU figures (all_gather), the proofs
    (all_gather_object) -- and, in one-proof-sharded mode, broadcast the communicator id and run every commitment through the
    library's RCCL communicator (ncclAllGather of the partials)."""
    env = _clean_env(ZKFHE_BENCH_FORCE_DIST="1", ZKFHE_TABLE_GB="4")
    env.pop("ZKFHE_BENCH_BACKEND", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--streams", "2", "--no-cpu-baseline", "--no-traffic-pass",
           "--steady-seconds", "0", "--mode", mode]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["gpus_requested"] == 1 and d["config"]["process_group"] == "nccl" and d["config"]["verified"] is True
    assert d["config"]["mode"] == mode and len(d["config"]["host_cpu_ms_per_proof_by_rank"]) == 1
    if mode == "batch":
        assert d["scaling"] == "weak" and d["config"]["proofs_gathered_and_verified"] == 1
    else:
        assert d["scaling"] == "strong" and d["config"]["sharded_proof_identical_on_all_ranks"] is True and d["config"]["concurrent_proofs_per_gpu"] == 1


@pytest.mark.gpu
def test_bench_refuses_a_launch_that_is_not_the_n_asked_for():
    """Never a line whose n_gpus differs from --gpus: two ranks launched with --gpus 1 exit non-zero naming both numbers, and so
    does the default (RCCL) backend with more local ranks than visible GPUs -- one process per GPU is the contract."""
    import torch
    env = _clean_env(ZKFHE_BENCH_BACKEND="gloo", ZKFHE_TABLE_GB="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "--gpus 1" in r.stderr and "WORLD_SIZE=2" in r.stderr
    if torch.cuda.device_count() < 2:
        env = _clean_env(ZKFHE_TABLE_GB="1")
        env.pop("ZKFHE_BENCH_BACKEND", None)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--no-cpu-baseline"], cwd=ROOT, env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert "2 local ranks but 1 visible GPU" in r.stderr
