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


_RCCL_PROBE = r"""
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
from visual_slam_amd import _lib, ba, parallel
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
rec = np.zeros(37, _lib.PAIR_DTYPE)
rec["T"] = np.random.default_rng(0).normal(size=(37, 4, 4)); rec["n_match"] = np.arange(37); rec["status"] = -np.arange(37)
got = parallel.gather_pair_results(rec, 37, force=True)
assert got.dtype == rec.dtype and all(np.array_equal(got[f], rec[f]) for f in rec.dtype.names)
assert parallel.max_over_ranks(1.25, force=True) == 1.25
H = torch.arange(5 * 36, dtype=torch.float64, device="cuda").reshape(5, 6, 6); b = torch.arange(30, dtype=torch.float64, device="cuda").reshape(5, 6)
H2, b2 = ba.allreduce_pose_blocks(H, b, force=True)
assert torch.equal(H2, H) and torch.equal(b2, b) and H2.is_cuda
Hn, bn = ba.allreduce_pose_blocks(H.cpu().numpy(), b.cpu().numpy(), force=True)
assert np.array_equal(Hn, H.cpu().numpy()) and np.array_equal(bn, b.cpu().numpy())
dist.barrier(); torch.cuda.synchronize()
dist.destroy_process_group()
print("rccl-ok", dist.is_nccl_available())
"""


def test_the_collectives_of_the_n_gt_1_path_run_on_rccl():
    """The three collectives bench.py uses at N > 1 -- all_gather of the packed 152-byte records (uint8), MAX all-reduce of the
    step time, SUM all-reduce of the [P, 42] float64 pose blocks -- through RCCL itself (backend "nccl") in a one-rank group
    on this box's GPU: two ranks cannot share a GPU under RCCL, so this is the part of the N > 1 path a one-GPU box can run;
    the sharding arithmetic around it runs under gloo (tests/test_parallel_cpu.py and the test above)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", _RCCL_PROBE], cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-4000:]
    assert "rccl-ok True" in out.stdout
