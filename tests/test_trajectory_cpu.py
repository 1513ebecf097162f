"""Pose chaining / trajectory / GT evaluation -- host logic, no GPU."""
import os

import numpy as np
from scipy.spatial.transform import Rotation

from visual_slam_amd import trajectory as tj

from conftest import GOLDEN


def _rand_transf(rng):
    T = np.eye(4)
    T[:3, :3] = Rotation.from_rotvec(rng.normal(0, 0.05, 3)).as_matrix()
    t = rng.normal(0, 1, 3)
    T[:3, 3] = t / np.linalg.norm(t)
    return T


def test_chain_matches_the_reference_loop():
    rng = np.random.default_rng(0)
    Ts = [_rand_transf(rng) for _ in range(20)]
    poses = tj.chain_poses(Ts)
    cur = np.eye(4)
    for T, p in zip(Ts, poses):   # vo.py:77-81 verbatim arithmetic
        cur = cur @ np.linalg.inv(np.nan_to_num(T, neginf=0, posinf=0))
        assert np.array_equal(cur, p)


def test_join_shards_equals_unsharded_chain():
    rng = np.random.default_rng(1)
    Ts = [_rand_transf(rng) for _ in range(31)]
    full = tj.chain_poses(Ts)
    for cuts in ([0, 16, 31], [0, 8, 16, 24, 31], [0, 1, 31]):
        shards = [tj.chain_poses(Ts[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
        joined = tj.join_shards(shards)
        assert len(joined) == len(full)
        assert np.allclose(np.array(joined), np.array(full), rtol=0, atol=1e-12)


def test_rodrigues_against_scipy():
    rng = np.random.default_rng(2)
    for _ in range(50):
        rv = rng.normal(0, 1, 3)
        R = Rotation.from_rotvec(rv).as_matrix()
        got = tj.rodrigues(R).ravel()
        want = Rotation.from_matrix(R).as_rotvec()
        assert np.allclose(got, want, atol=1e-9)
    assert np.array_equal(tj.rodrigues(np.eye(3)), np.zeros((3, 1)))
    Rpi = Rotation.from_rotvec([np.pi, 0, 0]).as_matrix()
    assert np.allclose(np.abs(tj.rodrigues(Rpi).ravel()), [np.pi, 0, 0], atol=1e-6)


def test_gt_relative_transforms_reproduce_the_gt_path():
    from visual_slam_amd.vo_utils import VisualOdometry
    gt = VisualOdometry.load_poses(os.path.join(GOLDEN, "kitti02", "poses.txt"))
    n = min(len(gt), 4)
    Ts = [tj.relative_gt(gt, i) for i in range(1, n)]
    poses = tj.chain_poses(Ts, start=gt[0])
    est = tj.estimated_path(poses)
    g = tj.gt_path(gt, 1, n - 1)
    err, diff = tj.gt_error(est, g)
    assert err < 1e-9 and diff.shape == (n - 1,)


def test_sanitize_and_failure_mode():
    T = np.eye(4); T[0, 3] = np.inf; T[1, 3] = np.nan
    S = tj.sanitize(T)
    assert S[0, 3] == 0 and S[1, 3] == 0
    import pytest
    with pytest.raises(np.linalg.LinAlgError):
        tj.chain_poses([np.zeros((4, 4))])   # a failed pair is singular, as in the reference


def test_projection_pairs_follow_the_reference_loop():
    """P1 = K @ prev_pose[:3], P2 = K @ cur_pose[:3] with prev_pose = the chained pose before the pair (vo.py:65-67,92,176;
    libs/vo_utils.py:153-157), for a chain that starts at identity and for one continued from a carried pose (second wave)."""
    from visual_slam_amd import trajectory as tj
    rng = np.random.default_rng(5)
    K = np.array([[718.856, 0, 607.1928], [0, 718.856, 185.2157], [0, 0, 1.0]])
    transfs = []
    for _ in range(7):
        w = rng.normal(0, 0.03, 3); th = np.linalg.norm(w); k = w / th
        Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        T = np.eye(4); T[:3, :3] = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx; T[:3, 3] = rng.normal(0, 1, 3)
        transfs.append(T)
    # the reference loop, literally
    cur = prev = np.eye(4)
    want = []
    for T in transfs:
        cur = cur @ np.linalg.inv(np.nan_to_num(T, neginf=0, posinf=0))
        want.append((K @ prev[:3, :], K @ cur[:3, :]))
        prev = cur
    poses = tj.chain_poses(transfs)
    P1, P2 = tj.projection_pairs(K, poses)
    assert P1.shape == P2.shape == (7, 3, 4)
    for i, (a, b) in enumerate(want):
        assert np.array_equal(P1[i], a) and np.array_equal(P2[i], b)
    # two waves: the second continues from the first wave's last pose
    first = tj.chain_poses(transfs[:3])
    second = tj.chain_poses(transfs[3:], start=first[-1])
    Q1, Q2 = tj.projection_pairs(K, second, start=first[-1])
    assert np.array_equal(Q1, P1[3:]) and np.array_equal(Q2, P2[3:])
