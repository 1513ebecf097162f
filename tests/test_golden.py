"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py from the
oracle): the oracle must keep reproducing them (CPU), and the GPU path must match them (gpu)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_gray


def _inputs(name):
    if name == "kitti02_pair_0_1":
        return load_gray(os.path.join(GOLDEN, "kitti02", "000000.png")), load_gray(os.path.join(GOLDEN, "kitti02", "000001.png"))
    from visual_slam_amd import synth
    a, b, *_ = synth.make_pair(0)
    return a, b


def _check(g, k1, f1, k2, f2, qidx, tidx, d1, d2, E, mask, R, t, X4, exact_pose):
    assert np.array_equal(k1.view(np.uint8), g["kps1"].view(np.uint8)) and np.array_equal(f1, g["desc1"])
    assert np.array_equal(k2.view(np.uint8), g["kps2"].view(np.uint8)) and np.array_equal(f2, g["desc2"])
    assert np.array_equal(qidx, g["qidx"]) and np.array_equal(tidx, g["tidx"])
    assert np.array_equal(d1, g["d1"]) and np.array_equal(d2, g["d2"])
    assert np.array_equal(mask, g["mask"])
    tol = 0 if exact_pose else 1e-9
    assert np.abs(E - g["E"]).max() <= tol
    # north-star: rotation / translation direction below 1e-3 rad; we hold 1e-6
    if exact_pose:
        assert np.array_equal(R, g["R"]) and np.array_equal(t, g["t"])
    assert np.arccos(np.clip((np.trace(R.T @ g["R"]) - 1) / 2, -1, 1)) <= 1e-6
    assert np.arccos(np.clip(t @ g["t"], -1, 1)) <= 1e-6
    pg = (g["X4"] / g["X4"][3])[:3]; p = (X4 / X4[3])[:3]
    assert np.abs(p - pg).max() / np.abs(pg).max() <= (0 if exact_pose else 1e-5)


@pytest.mark.parametrize("name", ["kitti02_pair_0_1", "synth_seed0"])
def test_oracle_reproduces_golden(oracle, name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    a, b = _inputs(name)
    K = g["K"]
    k1, f1 = oracle.orb_detect_compute(a); k2, f2 = oracle.orb_detect_compute(b)
    i1, d1, i2, d2 = oracle.match_knn2(f1, f2)
    keep = oracle.ratio_filter(d1, d2, 0.45)
    q1 = np.stack([k1["x"][keep], k1["y"][keep]], 1); q2 = np.stack([k2["x"][i1[keep]], k2["y"][i1[keep]]], 1)
    E, mask, _, _ = oracle.find_essential_ransac(q1, q2, K)
    R, t, _ = oracle.recover_pose(E, q1, q2, K)
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = t
    X4 = oracle.triangulate_f32(K @ np.eye(4)[:3], K @ np.linalg.inv(T)[:3], q1, q2)
    _check(g, k1, f1, k2, f2, keep, i1[keep], d1[keep], d2[keep], E, mask, R, t, X4, True)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["kitti02_pair_0_1", "synth_seed0"])
def test_gpu_matches_golden(name):
    from visual_slam_amd import _lib
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    a, b = _inputs(name)
    K = g["K"]
    ctx = _lib.Context(max_w=a.shape[1], max_h=a.shape[0])
    k1, f1 = ctx.orb_detect_compute(a); k2, f2 = ctx.orb_detect_compute(b)
    qidx, tidx, d1, d2 = ctx.match_knn2_ratio(f1, f2, 0.45)
    q1 = np.stack([k1["x"][qidx], k1["y"][qidx]], 1); q2 = np.stack([k2["x"][tidx], k2["y"][tidx]], 1)
    E, mask, _ = ctx.find_essential_ransac(q1, q2, K)
    R, t, _ = ctx.recover_pose(E, q1, q2, K)
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = t
    X4 = ctx.triangulate(K @ np.eye(4)[:3], K @ np.linalg.inv(T)[:3], q1, q2)
    ctx.close()
    _check(g, k1, f1, k2, f2, qidx, tidx, d1, d2, E, mask, R, t, X4, False)
