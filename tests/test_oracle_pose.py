"""Known-answer tests for the oracle's matcher, five-point RANSAC, recoverPose and DLT
triangulation, built from the mathematical definitions (parity unpinned)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN


def rodrigues(rvec):
    th = np.linalg.norm(rvec)
    if th < 1e-15:
        return np.eye(3)
    k = rvec / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx


def rot_angle(Ra, Rb):
    c = (np.trace(Ra.T @ Rb) - 1) / 2
    return float(np.arccos(np.clip(c, -1, 1)))


def dir_angle(a, b):
    c = np.dot(a, b) / (np.linalg.norm(a) * np.linalg.norm(b))
    return float(np.arccos(np.clip(c, -1, 1)))


def two_view(seed, n, K, noise=0.0, outliers=0):
    rng = np.random.default_rng(seed)
    X = np.stack([rng.uniform(-8, 8, n), rng.uniform(-3, 3, n), rng.uniform(6, 40, n)], 1)
    R = rodrigues(np.array([0.01, -0.03, 0.005]) * (1 + seed))
    t = np.array([0.1, -0.05, -1.0]); t /= np.linalg.norm(t)
    x1 = (K @ X.T).T; x1 = x1[:, :2] / x1[:, 2:]
    X2 = (R @ X.T).T + t
    x2 = (K @ X2.T).T; x2 = x2[:, :2] / x2[:, 2:]
    x1 = x1 + rng.normal(0, noise, x1.shape) if noise else x1
    x2 = x2 + rng.normal(0, noise, x2.shape) if noise else x2
    if outliers:
        x2[:outliers] = rng.uniform(0, 1000, (outliers, 2))
    return x1.astype(np.float32), x2.astype(np.float32), R, t, X


def test_hamming_knn2_vs_numpy(oracle):
    rng = np.random.default_rng(0)
    q = rng.integers(0, 256, (300, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (257, 32), dtype=np.uint8)
    t[10] = t[200]          # exact duplicate: tie -> lower train index first
    q[5] = t[10]
    i1, d1, i2, d2 = oracle.match_knn2(q, t)
    D = np.unpackbits(q[:, None, :] ^ t[None, :, :], axis=2).sum(2).astype(np.int32)
    order = np.argsort(D, axis=1, kind="stable")
    assert np.array_equal(i1, order[:, 0]) and np.array_equal(i2, order[:, 1])
    assert np.array_equal(d1, D[np.arange(300), order[:, 0]]) and np.array_equal(d2, D[np.arange(300), order[:, 1]])
    assert i1[5] == 10 and d1[5] == 0 and i2[5] == 200 and d2[5] == 0
    with pytest.raises(ValueError):
        oracle.match_knn2(q, t[:1])


def test_ratio_is_python_double_semantics(oracle):
    d1, d2 = np.meshgrid(np.arange(257), np.arange(257), indexing="ij")
    d1 = d1.ravel().astype(np.int32); d2 = d2.ravel().astype(np.int32)
    keep = oracle.ratio_filter(d1, d2, 0.45)
    ref = np.array([i for i in range(len(d1)) if float(np.float32(d1[i])) < 0.45 * float(np.float32(d2[i]))], np.int32)
    assert np.array_equal(keep, ref)
    # equivalent to 20*d1 < 9*d2 over the whole Hamming range
    assert np.array_equal(keep, np.nonzero(20 * d1 < 9 * d2)[0])


def test_five_point_minimal_solver(oracle, K02):
    for seed in range(5):
        x1, x2, R, t, _ = two_view(seed, 5, K02)
        Kinv = np.linalg.inv(K02)
        n1 = (Kinv @ np.c_[x1.astype(np.float64), np.ones(5)].T).T[:, :2]
        n2 = (Kinv @ np.c_[x2.astype(np.float64), np.ones(5)].T).T[:, :2]
        Es = oracle.five_point(n1, n2)
        assert 1 <= len(Es) <= 10
        tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
        Etrue = tx @ R; Etrue /= np.linalg.norm(Etrue)
        best = 1e9
        for E in Es:
            assert abs(np.linalg.norm(E) - 1) < 1e-12
            # every solution satisfies the epipolar constraint on the 5 points and the cubic constraints
            for a, b in zip(n1, n2):
                assert abs(np.r_[b, 1] @ E @ np.r_[a, 1]) < 1e-9
            assert abs(np.linalg.det(E)) < 1e-9
            assert np.abs(2 * E @ E.T @ E - np.trace(E @ E.T) * E).max() < 1e-8
            best = min(best, np.abs(E - Etrue).max(), np.abs(E + Etrue).max())
        assert best < 1e-4  # float32 pixel coordinates limit the match to the true E


def test_ransac_pose_noise_free(oracle, K02):
    for seed in range(3):
        x1, x2, R, t, _ = two_view(seed, 200, K02)
        E, mask, ninl, iters = oracle.find_essential_ransac(x1, x2, K02)
        assert ninl >= 195 and mask.sum() == ninl
        Rr, tr, ngood = oracle.recover_pose(E, x1, x2, K02)
        assert rot_angle(R, Rr) < 1e-4 and dir_angle(t, tr) < 1e-3
        assert abs(np.linalg.norm(tr) - 1) < 1e-12 and abs(np.linalg.det(Rr) - 1) < 1e-9
        assert ngood >= 195
        assert iters <= 20  # adaptive stopping at ~100% inliers


def test_ransac_with_outliers_and_noise(oracle, K02):
    x1, x2, R, t, _ = two_view(7, 400, K02, noise=0.3, outliers=120)
    E, mask, ninl, iters = oracle.find_essential_ransac(x1, x2, K02)
    assert mask[:120].sum() <= 12 and mask[120:].sum() >= 200
    Rr, tr, _ = oracle.recover_pose(E, x1, x2, K02)
    assert rot_angle(R, Rr) < 5e-3 and dir_angle(t, tr) < 5e-2
    assert 5 < iters < 1000


def test_ransac_degenerate_counts(oracle, K02):
    x1, x2, *_ = two_view(0, 4, K02)
    with pytest.raises(RuntimeError):
        oracle.find_essential_ransac(x1, x2, K02)
    x1, x2, *_ = two_view(0, 5, K02)
    E, mask, ninl, _ = oracle.find_essential_ransac(x1, x2, K02)
    assert ninl == 5 and mask.all()


def test_triangulate_reprojects(oracle, K02):
    x1, x2, R, t, X = two_view(2, 50, K02)
    P1 = K02 @ np.eye(4)[:3]
    P2 = K02 @ np.c_[R, t]
    X4 = oracle.triangulate_f32(P1, P2, x1, x2)
    assert X4.dtype == np.float32 and X4.shape == (4, 50)
    Xh = (X4 / X4[3])[:3].T
    assert np.abs(Xh - X).max() / np.abs(X).max() < 2e-3  # float32 pixels, float32 output
    # a single point through the reference's call shape (libs/vo_utils.py:160-164)
    one = oracle.triangulate_f32(P1, P2, x1[3], x2[3])
    assert np.allclose((one / one[3])[:3, 0], Xh[3], rtol=1e-6)


def test_kitti_pair_against_ground_truth(oracle, K02):
    """Plausibility gate (not the parity gate): recovered relative pose vs KITTI GT."""
    from conftest import load_gray
    poses = [np.vstack([np.fromstring(l, sep=" ").reshape(3, 4), [0, 0, 0, 1]])
             for l in open(os.path.join(GOLDEN, "kitti02", "poses.txt"))]
    imgs = [load_gray(os.path.join(GOLDEN, "kitti02", "%06d.png" % i)) for i in range(3)]
    for i in (1, 2):
        rc, T, counts = oracle.pair_run(imgs[i - 1], imgs[i], K02)
        assert rc == 0 and counts[2] > 100 and counts[3] > 50
        rel = np.linalg.inv(poses[i]) @ poses[i - 1]  # maps frame i-1 coordinates into frame i
        assert rot_angle(rel[:3, :3], T[:3, :3]) < 1e-2
        assert dir_angle(rel[:3, 3], T[:3, 3]) < 6e-2


def test_pose_rounding_variants_agree_to_rounding(oracle):
    """oracle.set_pose_variant (pin-readiness): std::hypot in the Jacobi rotations and the LU-inverse +
    product form of the 10 x 10 elimination are the same mathematics as the defaults, so on ordinary five-point samples they
    give the same number of models, equal to 1e-9, while not (always) the same bits -- which is why they are switches."""
    K = np.array([[718.856, 0, 607.1928], [0, 718.856, 185.2157], [0, 0, 1.0]])
    rng = np.random.default_rng(3)
    x1, x2, *_ = two_view(4, 80, K, noise=0.2)
    n1 = (x1 - K[:2, 2]) / K[0, 0]; n2 = (x2 - K[:2, 2]) / K[0, 0]
    differ = 0
    for _ in range(40):
        idx = rng.choice(80, 5, replace=False)
        base = oracle.five_point(n1[idx], n2[idx])
        for hm, am in ((1, 0), (0, 1), (1, 1)):
            with oracle.pose_variant(hypot_mode=hm, ainv_mode=am):
                v = oracle.five_point(n1[idx], n2[idx])
            assert len(v) == len(base)
            if len(v):
                # the models of a sample come out in root order: equal up to rounding, possibly with the sign E -> -E kept
                assert np.abs(v - base).max() < 1e-6
                differ += int(not np.array_equal(v, base))
        assert np.array_equal(oracle.five_point(n1[idx], n2[idx]), base)      # the context manager restored the defaults
    assert differ > 0
    # and whole RANSAC runs land on the same consensus set on a clean problem
    x1, x2, *_ = two_view(6, 300, K, noise=0.3, outliers=60)
    E0, m0, n0, _ = oracle.find_essential_ransac(x1, x2, K)
    for hm, am in ((1, 0), (0, 1), (1, 1)):
        with oracle.pose_variant(hypot_mode=hm, ainv_mode=am):
            Ev, mv, nv, _ = oracle.find_essential_ransac(x1, x2, K)
        assert abs(nv - n0) <= 3 and np.abs(np.abs(Ev) - np.abs(E0)).max() < 1e-3
