"""The N > 1 code path of bench.py on the GPU box: two FRESH rank processes under torch.distributed.run share the one
GPU (gloo rendezvous; on an 8-GPU node the driver uses RCCL), each takes its contiguous shard of the 64 seeded pairs
(visual_slam_amd.parallel.shard_range), and the records gathered on rank 0 must equal a single-process run of the same
seeds bit for bit.  North star: "independent frame pairs ... shard naturally" (/root/reference/vo.py:80 chains poses only
after the fact)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(cmd, timeout=300):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-4000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


def test_two_ranks_on_one_gpu_equal_one_process(tmp_path):
    common = ["--pairs", "64", "--chunk", "8", "--steps", "2", "--warmup", "1", "--no-cpu", "--no-extras", "--workers", "2"]
    two = tmp_path / "two.npy"
    one = tmp_path / "one.npy"
    j2 = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
               "--dump-records", str(two)] + common)
    j1 = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dump-records", str(one)] + common)
    assert j2["n_gpus"] == 2 and j1["n_gpus"] == 1
    assert j2["config"]["pairs_ok"] == 64 and j2["config"]["pairs_total"] == 64 and j2["config"]["pairs_on_rank0"] == 32
    assert j1["config"]["pairs_ok"] == 64 and j2["scaling"] == "strong" and j2["value"] > 0
    a, b = np.load(two), np.load(one)
    assert a.dtype == b.dtype and len(a) == len(b) == 64
    for f in a.dtype.names:
        assert np.array_equal(a[f], b[f]), f
