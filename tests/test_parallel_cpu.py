"""N>1 path on CPU: world_size-2 gloo processes shard the pairs, gather the per-pair records and
reduce the step time exactly as bench.py / a multi-GPU caller does (no data-path collective)."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from visual_slam_amd import parallel
from visual_slam_amd._lib import PAIR_DTYPE


def test_shard_range_partitions():
    for total in (0, 1, 7, 4096, 4097):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, total, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = parallel.shard_range(total, rank, world)
    local = np.zeros(hi - lo, PAIR_DTYPE)
    local["n_match"] = np.arange(lo, hi)          # stand-in for the per-pair results of this shard
    local["T"][:, 0, 0] = np.arange(lo, hi) * 0.5
    local["status"] = -rank
    allres = parallel.gather_pair_results(local, total)
    tmax = parallel.max_over_ranks(1.0 + rank)
    dist.barrier()
    if rank == 0:
        q.put((allres["n_match"].tolist(), allres["T"][:, 0, 0].tolist(), allres["status"].tolist(), tmax))
    dist.destroy_process_group()


def test_two_rank_gloo_shard_and_gather():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    total, world = 11, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    n_match, t00, status, tmax = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert n_match == list(range(total))                       # every pair exactly once, in order
    assert t00 == [i * 0.5 for i in range(total)]
    assert status == [0] * 6 + [-1] * 5
    assert tmax == 2.0


def _rigid(rng):
    """a KITTI-like relative transform: small rotation, unit-norm translation"""
    w = rng.normal(0, 0.02, 3)
    th = np.linalg.norm(w); k = w / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    t = rng.normal(0, 1, 3); t /= np.linalg.norm(t)
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = t
    return T


def _seq_worker(rank, world, port, transfs, q):
    """One long sequence sharded over ranks: rank r chains ITS block of consecutive pairs from identity
    (that is what it can do without waiting for the ranks before it), the per-pair records are all_gathered, and
    trajectory.join_shards applies the carried pose of the preceding blocks."""
    import torch.distributed as dist
    from visual_slam_amd import trajectory as tj
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    total = len(transfs)
    lo, hi = parallel.shard_range(total, rank, world)
    local = np.zeros(hi - lo, PAIR_DTYPE)
    local["T"] = transfs[lo:hi]                     # stand-in for sequence_run_device on frames [lo, hi]
    allres = parallel.gather_pair_results(local, total)
    shards = [tj.chain_poses(allres["T"][a:b]) for a, b in (parallel.shard_range(total, r, world) for r in range(world))]
    joined = tj.join_shards(shards)
    dist.barrier()
    if rank == 0:
        q.put(np.array(joined))
    dist.destroy_process_group()


def test_two_rank_sharded_sequence_joins_to_the_unsharded_chain():
    from visual_slam_amd import trajectory as tj
    rng = np.random.default_rng(3)
    transfs = np.array([_rigid(rng) for _ in range(175)])          # KITTI 02 snippet length: 176 frames
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_seq_worker, args=(r, 2, port, transfs, q)) for r in range(2)]
    for p in procs:
        p.start()
    joined = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    whole = np.array(tj.chain_poses(transfs))
    assert joined.shape == whole.shape == (175, 4, 4)
    assert np.abs(joined - whole).max() <= 1e-12
