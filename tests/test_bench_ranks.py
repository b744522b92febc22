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
    env["ZKFHE_TABLE_GB"] = "4"          # two SRS on one device: the suite's budget, not the 86 GB default
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
