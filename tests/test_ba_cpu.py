"""BA residual/Jacobian oracle (oracle/ba_oracle.py) against finite differences, the synthetic
config-5 generator, and the sharded all-reduce with 2 gloo ranks -- no GPU."""
import os
import socket

import numpy as np
import pytest

from oracle import ba_oracle as BO
from visual_slam_amd import ba


def test_synth_problem_is_consistent():
    pr = ba.synth_problem(n_poses=50, n_points=400, seed=0, noise=0.0)
    assert pr["poses"].shape == (50, 6) and pr["points"].shape == (400, 3) and len(pr["obs_pose"]) == 2000
    assert (np.diff(pr["obs_point"]) >= 0).all()                     # sorted by point
    for o in range(0, 2000, 97):                                      # noise-free observations reproject exactly
        p = BO.project(pr["poses"][pr["obs_pose"][o]], pr["points"][pr["obs_point"][o]], pr["K"])
        assert np.allclose(p, pr["uv"][o], atol=1e-9)
    # points are in front of their cameras
    out = BO.ba_pass(pr["poses"], pr["points"], pr["obs_pose"][:200], pr["obs_point"][:200], pr["uv"][:200], pr["K"])
    assert np.abs(out["r"]).max() < 1e-8


def test_analytic_jacobian_matches_finite_differences():
    rng = np.random.default_rng(0)
    K = np.array([[718.856, 0, 607.1928], [0, 718.856, 185.2157], [0, 0, 1.0]])
    for trial in range(30):
        pose = np.concatenate([rng.normal(0, 0.4, 3), rng.normal(0, 1, 3)])
        if trial == 0:
            pose[:3] = 0           # the theta -> 0 branch
        X = np.array([rng.uniform(-3, 3), rng.uniform(-2, 2), rng.uniform(6, 30)])
        _, Jp, Jl = BO.obs_jacobians(pose, X, K)
        h = 1e-6
        for a in range(6):
            d = np.zeros(6); d[a] = h
            fd = (BO.project(pose + d, X, K) - BO.project(pose - d, X, K)) / (2 * h)
            assert np.allclose(Jp[:, a], fd, rtol=1e-6, atol=1e-5), (trial, a)
        for a in range(3):
            d = np.zeros(3); d[a] = h
            fd = (BO.project(pose, X + d, K) - BO.project(pose, X - d, K)) / (2 * h)
            assert np.allclose(Jl[:, a], fd, rtol=1e-6, atol=1e-5)


def test_scalar_residual_is_the_references():
    """norm == VisualOdometry.calc_reprojection_error (libs/vo_utils.py:184-216) for one camera."""
    from visual_slam_amd.vo_utils import VisualOdometry
    pr = ba.synth_problem(n_poses=8, n_points=60, seed=3)
    sel = pr["obs_pose"] == 4
    out = BO.ba_pass(pr["poses"], pr["points"], pr["obs_pose"][sel], pr["obs_point"][sel], pr["uv"][sel], pr["K"])
    vo = VisualOdometry.__new__(VisualOdometry)
    ref = vo.calc_reprojection_error(pr["K"], np.zeros(5), pr["poses"][4, :3], pr["poses"][4, 3:], pr["points"][pr["obs_point"][sel]], pr["uv"][sel])
    assert np.allclose(out["norm"], ref, rtol=1e-12, atol=1e-12)


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pr = ba.synth_problem(n_poses=20, n_points=101, seed=1)
    sh = ba.shard_observations(pr, rank, world)
    part = BO.ba_pass(sh["poses"], sh["points"], sh["obs_pose"], sh["obs_point"], sh["uv"], sh["K"])   # oracle stands in for the GPU pass
    H, b = ba.allreduce_pose_blocks(part["Hpp"], part["bp"])
    q.put((rank, H, b, len(sh["obs_pose"])))
    dist.destroy_process_group()


def test_sharded_pose_blocks_allreduce_gloo():
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    got = [q.get(timeout=120) for _ in ps]
    [p.join(60) for p in ps]
    pr = ba.synth_problem(n_poses=20, n_points=101, seed=1)
    full = BO.ba_pass(pr["poses"], pr["points"], pr["obs_pose"], pr["obs_point"], pr["uv"], pr["K"])
    assert sum(g[3] for g in got) == len(pr["obs_pose"])
    for _, H, b, _ in got:
        assert np.allclose(H, full["Hpp"], rtol=1e-12, atol=1e-9) and np.allclose(b, full["bp"], rtol=1e-12, atol=1e-9)


def test_schur_step_solves_the_normal_equations():
    """ba.schur_step (host logic of the BA caller side, vo.py:112-135) on the numpy oracle's blocks: the step equals the solution
    of the full damped normal equations assembled densely, and a few steps reach the noise floor."""
    P, N, noise = 4, 60, 0.2
    pr = ba.synth_problem(n_poses=P, n_points=N, obs_per_point=3, seed=4, noise=noise)
    rng = np.random.default_rng(1)
    poses = pr["poses"].copy(); points = pr["points"] + rng.normal(0, 0.1, pr["points"].shape)
    poses[1:, 3:] += rng.normal(0, 0.03, (P - 1, 3))
    run = lambda ps, xs: BO.ba_pass(ps, xs, pr["obs_pose"], pr["obs_point"], pr["uv"], pr["K"])
    out = run(poses, points)
    lam = 1e-2
    dp, dl = ba.schur_step(out, pr["obs_pose"], pr["obs_point"], lam, fixed_poses=(0,))
    # dense reference: unknowns = poses 1..P-1 and all points
    n = 6 * P + 3 * N
    H = np.zeros((n, n)); g = np.zeros(n)
    for j in range(P):
        H[6 * j:6 * j + 6, 6 * j:6 * j + 6] = out["Hpp"][j]; g[6 * j:6 * j + 6] = -out["bp"][j]
    for i in range(N):
        a = 6 * P + 3 * i
        H[a:a + 3, a:a + 3] = out["Hll"][i]; g[a:a + 3] = -out["bl"][i]
    for o, (j, i) in enumerate(zip(pr["obs_pose"], pr["obs_point"])):
        a = 6 * P + 3 * i
        H[6 * j:6 * j + 6, a:a + 3] += out["Hpl"][o]; H[a:a + 3, 6 * j:6 * j + 6] += out["Hpl"][o].T
    H += lam * np.diag(np.maximum(np.diag(H), 1e-12))
    free = np.ones(n, bool); free[:6] = False
    x = np.zeros(n); x[free] = np.linalg.solve(H[np.ix_(free, free)], g[free])
    assert np.allclose(dp.reshape(-1), x[:6 * P], rtol=1e-7, atol=1e-10) and np.allclose(dl.reshape(-1), x[6 * P:], rtol=1e-7, atol=1e-10)
    cost = lambda o: float((o["r"] ** 2).sum())
    c = [cost(out)]
    for _ in range(8):
        dp, dl = ba.schur_step(out, pr["obs_pose"], pr["obs_point"], 1e-4, fixed_poses=(0,))
        poses, points = poses + dp, points + dl
        out = run(poses, points); c.append(cost(out))
    assert np.sqrt(c[-1] / (2 * len(pr["uv"]))) < 1.2 * noise < np.sqrt(c[0] / (2 * len(pr["uv"])))
