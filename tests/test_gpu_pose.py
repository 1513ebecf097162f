"""GPU parity of matcher, RANSAC, recoverPose, triangulation and the whole pair path
against the CPU oracle, through the C ABI."""
import os

import numpy as np
import pytest

from test_oracle_pose import two_view, rot_angle, dir_angle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from visual_slam_amd import _lib
    c = _lib.Context(max_w=1241, max_h=376, chunk_pairs=2, max_pairs=4)
    yield c
    c.close()


def test_match_knn2_bit_exact(ctx, oracle):
    rng = np.random.default_rng(1)
    for nq, nt in [(300, 257), (3000, 2973), (1, 2), (65, 4000), (4024, 4024)]:
        q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
        t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
        if nt > 210 and nq > 5:
            t[10] = t[200]; q[5] = t[10]   # exact ties: lower train index first
        got = ctx.match_knn2(q, t)
        ref = oracle.match_knn2(q, t)
        for a, b in zip(got, ref):
            assert np.array_equal(a, b), (nq, nt)


def test_match_extremes_and_collisions():
    """Matrix-core matcher corner cases vs a numpy brute force: distances 0 and 256 (all-zero / all-one
    descriptors), every train identical (ties broken by lowest trainIdx), train counts around the 32-row
    block and the 2016-train segment limits, query counts around the 384-query workgroup."""
    from visual_slam_amd import _lib
    rng = np.random.default_rng(7)
    LUT = np.array([bin(i).count("1") for i in range(256)], np.int32)

    def brute(q, t):
        d = LUT[q[:, None, :] ^ t[None, :, :]].sum(2)
        key = d.astype(np.int64) * 65536 + np.arange(t.shape[0])[None, :]
        o = np.argsort(key, axis=1, kind="stable")[:, :2]
        return o[:, 0], np.take_along_axis(d, o[:, :1], 1)[:, 0], o[:, 1], np.take_along_axis(d, o[:, 1:2], 1)[:, 0]

    c = _lib.Context(max_w=640, max_h=480, chunk_pairs=1, max_pairs=1, nfeatures=8000)   # kp_cap 9024 -> 5 train segments
    try:
        cases = []
        q = rng.integers(0, 256, (383, 32), dtype=np.uint8); t = np.zeros((33, 32), np.uint8); t[:] = 0xFF; q[0] = 0; q[1] = 0xFF
        cases.append((q, t))                                             # identical trains, d = 256 and d = 0
        q = rng.integers(0, 256, (385, 32), dtype=np.uint8); t = rng.integers(0, 256, (2, 32), dtype=np.uint8)
        cases.append((q, t))                                             # the minimum train set
        for nt in (31, 32, 63, 64, 65, 2016 * 2 + 1, 8063):
            q = rng.integers(0, 256, (97, 32), dtype=np.uint8)
            t = rng.integers(0, 4, (nt, 32), dtype=np.uint8)              # few distinct bits -> many distance ties
            t[nt - 1] = q[3]                                              # an exact match in the ragged last block
            cases.append((q, t))
        for q, t in cases:
            got = c.match_knn2(q, t)
            ref = brute(q, t)
            for a, b in zip(got, ref):
                assert np.array_equal(np.asarray(a, np.int64), np.asarray(b, np.int64)), (q.shape, t.shape)
    finally:
        c.close()


def test_match_on_real_descriptors_and_ratio(ctx, oracle, kitti02):
    k1, f1 = oracle.orb_detect_compute(kitti02[0])
    k2, f2 = oracle.orb_detect_compute(kitti02[1])
    i1, d1, i2, d2 = oracle.match_knn2(f1, f2)
    for ratio in (0.45, 0.7, 1.0):
        keep = oracle.ratio_filter(d1, d2, ratio)
        qi, ti, a, b = ctx.match_knn2_ratio(f1, f2, ratio)
        assert np.array_equal(qi, keep) and np.array_equal(ti, i1[keep])
        assert np.array_equal(a, d1[keep]) and np.array_equal(b, d2[keep])


def test_match_errors(ctx):
    from visual_slam_amd._lib import VoError, VO_ERR_TOO_FEW, VO_ERR_CAPACITY
    q = np.zeros((10, 32), np.uint8)
    with pytest.raises(VoError) as e:
        ctx.match_knn2(q, q[:1])
    assert e.value.code == VO_ERR_TOO_FEW
    with pytest.raises(VoError) as e:
        ctx.match_knn2(np.zeros((5000, 32), np.uint8), q)
    assert e.value.code == VO_ERR_CAPACITY
    assert len(ctx.match_knn2_ratio(np.zeros((0, 32), np.uint8), q)[0]) == 0


def test_ransac_matches_oracle(ctx, oracle, K02):
    for seed, n, noise, outl in [(0, 200, 0.0, 0), (3, 400, 0.3, 120), (5, 900, 0.5, 500), (6, 50, 0.2, 10)]:
        x1, x2, R, t, _ = two_view(seed, n, K02, noise=noise, outliers=outl)
        Eo, mo, no, iters = oracle.find_essential_ransac(x1, x2, K02)
        E, m, ni = ctx.find_essential_ransac(x1, x2, K02)
        assert ni == no and np.array_equal(m, mo), (seed, iters)
        assert np.abs(E - Eo).max() < 1e-9
        Ro, to, go = oracle.recover_pose(Eo, x1, x2, K02)
        Rg, tg, gg = ctx.recover_pose(E, x1, x2, K02)
        assert gg == go
        assert rot_angle(Ro, Rg) < 1e-6 and dir_angle(to, tg) < 1e-6   # north-star bound is 1e-3 rad


def test_all_four_cheirality_counts_match_the_oracle(ctx, oracle, K02):
    """recoverPose (libs/vo_utils.py:143) counts, for each of [R1|t], [R2|t], [R1|-t], [R2|-t], the matches that triangulate
    in front of both cameras.  The GPU triangulates with +t only and derives the -t counts from the sign symmetry of the
    Jacobi SVD; the oracle triangulates all four.  Same E in, all four counts equal -- on clean scenes,
    on scenes that are mostly outliers (every candidate gets votes) and with a depth threshold that cuts through the points."""
    from visual_slam_amd import _lib as L
    rng = np.random.default_rng(77)
    nonzero = np.zeros(4, np.int64)
    for k in range(40):
        n = int(rng.integers(8, 1500))
        x1, x2, R, t, _ = two_view(100 + k, n, K02, noise=float(rng.uniform(0, 1.0)), outliers=int(n * rng.uniform(0, 0.9)))
        if k % 4 == 3:          # pure garbage correspondences
            x2 = rng.uniform(0, 1200, x2.shape).astype(np.float32)
        E, _, _ = ctx.find_essential_ransac(x1, x2, K02)
        dist = 50.0 if k % 3 else float(rng.uniform(2, 30))
        Ro, to, go, votes_o = oracle.recover_pose_votes(E, x1, x2, K02, dist)
        Rg, tg, gg = ctx.recover_pose(E, x1, x2, K02, dist)
        votes_g = ctx.debug_read(L.VO_DBG_VOTES, 0, 0, np.int32)
        assert np.array_equal(votes_g, votes_o), (k, n, votes_g, votes_o)
        assert gg == go and np.array_equal(Rg, Ro) and np.array_equal(np.ravel(tg), np.ravel(to)), k
        nonzero += votes_o > 0
    assert (nonzero >= 5).all(), nonzero          # every candidate did collect votes in several scenes


def test_ransac_degenerate_counts(ctx, oracle, K02):
    from visual_slam_amd._lib import VoError, VO_ERR_TOO_FEW
    x1, x2, *_ = two_view(0, 4, K02)
    with pytest.raises(VoError) as e:
        ctx.find_essential_ransac(x1, x2, K02)
    assert e.value.code == VO_ERR_TOO_FEW
    x1, x2, *_ = two_view(0, 5, K02)
    E, mask, ninl = ctx.find_essential_ransac(x1, x2, K02)
    Eo, mo, no, _ = oracle.find_essential_ransac(x1, x2, K02)
    assert ninl == 5 and mask.all() and np.abs(E - Eo).max() < 1e-9


def test_ransac_many_iterations(ctx, oracle, K02):
    """Low inlier ratio: several rounds of 64 hypotheses, adaptive stopping must agree."""
    x1, x2, *_ = two_view(11, 300, K02, noise=0.2, outliers=200)
    Eo, mo, no, iters = oracle.find_essential_ransac(x1, x2, K02)
    assert iters > 64
    E, m, ni = ctx.find_essential_ransac(x1, x2, K02)
    assert ni == no and np.array_equal(m, mo) and np.abs(E - Eo).max() < 1e-9


def test_triangulate_matches_oracle(ctx, oracle, K02):
    x1, x2, R, t, X = two_view(2, 777, K02, noise=0.1)
    P1 = K02 @ np.eye(4)[:3]
    P2 = K02 @ np.c_[R, t]
    Xo = oracle.triangulate_f32(P1, P2, x1, x2)
    Xg = ctx.triangulate(P1, P2, x1, x2)
    po = (Xo / Xo[3])[:3]; pg = (Xg / Xg[3])[:3]
    assert np.abs(po - pg).max() / np.abs(po).max() < 1e-5   # north-star: relative error below 1e-5
    assert np.abs(np.abs(Xo) - np.abs(Xg)).max() < 1e-6


def test_pairs_run_matches_oracle(ctx, oracle, kitti02, kitti09, K02):
    from visual_slam_amd import _lib as L
    A = np.stack([kitti02[0], kitti02[1], kitti02[2]])
    B = np.stack([kitti02[1], kitti02[2], kitti02[3]])
    res = ctx.pairs_run(A, B, K02)
    for i in range(3):
        rc, T, counts = oracle.pair_run(A[i], B[i], K02)
        r = res[i]
        assert rc == 0 and r["status"] == 0
        assert [r["n_kp1"], r["n_kp2"], r["n_match"], r["n_inlier"]] == counts.tolist()
        assert rot_angle(T[:3, :3], r["T"][:3, :3]) < 1e-6 and dir_angle(T[:3, 3], r["T"][:3, 3]) < 1e-6
        assert np.array_equal(r["T"][3], [0, 0, 0, 1])
    # match indices are bit exact too: the context (chunk_pairs 2, max_pairs 4) has two lanes, so the host-image path stages
    # all three pairs at once and they are rows 0..2 of the last pose wave, written by two different lanes
    for i in range(3):
        k1, f1 = oracle.orb_detect_compute(A[i]); k2, f2 = oracle.orb_detect_compute(B[i])
        i1, d1, i2, d2 = oracle.match_knn2(f1, f2)
        keep = oracle.ratio_filter(d1, d2, 0.45)
        mq = ctx.debug_read(L.VO_DBG_MATCHES, i, 0, np.int32).reshape(-1, 4)
        assert np.array_equal(mq[:, 0], keep) and np.array_equal(mq[:, 1], i1[keep])


def test_pairs_run_reports_failures_per_pair(ctx, kitti02, K02):
    from visual_slam_amd._lib import VO_ERR_TOO_FEW
    flat = np.full_like(kitti02[0], 90)
    res = ctx.pairs_run(np.stack([kitti02[0], flat]), np.stack([kitti02[1], flat]), K02)
    assert res[0]["status"] == 0 and res[1]["status"] == VO_ERR_TOO_FEW
    assert res[1]["n_kp1"] == 0 and not res[1]["T"].any()


def test_dropin_visual_odometry(oracle, K02):
    """The reference's call sequence (vo.py:74-93) through the drop-in class."""
    from conftest import GOLDEN
    from visual_slam_amd import VisualOdometry
    vo = VisualOdometry(os.path.join(GOLDEN, "kitti02"), False)
    assert np.allclose(vo.K, K02) and vo.D.shape == (5,) and len(vo.image_paths) == 4 and len(vo.gt_poses) == 4
    q1, q2 = vo.get_matches(1, "ORB", None, None)
    assert q1.dtype == np.float32 and q1.shape == q2.shape and q1.shape[1] == 2
    transf = np.nan_to_num(vo.get_pose(q1, q2), neginf=0, posinf=0)
    from conftest import load_gray
    rc, T, counts = oracle.pair_run(load_gray(vo.image_paths[0]), load_gray(vo.image_paths[1]), K02)
    assert len(q1) == counts[2]
    assert rot_angle(T[:3, :3], transf[:3, :3]) < 1e-6 and dir_angle(T[:3, 3], transf[:3, 3]) < 1e-6
    prev_pose = np.eye(4); cur_pose = prev_pose @ np.linalg.inv(transf)
    p = vo.triangulate(q1[0], q2[0], prev_pose, cur_pose)
    assert p.shape == (3,) and p.dtype == np.float32
    pb = vo.triangulate(q1, q2, prev_pose, cur_pose)
    assert pb.shape == (len(q1), 3) and np.allclose(pb[0], p, rtol=1e-6)
    P1 = vo.K @ prev_pose[:3]; P2 = vo.K @ cur_pose[:3]
    Xo = oracle.triangulate_f32(P1, P2, q1, q2)
    ref = (Xo / Xo[3])[:3].T
    assert np.abs(pb - ref).max() / np.abs(ref).max() < 1e-5
    with pytest.raises(NotImplementedError):
        vo.get_matches(1, "SIFT", None, None)


def test_config4_4k_20k_features(oracle):
    """BASELINE.json configs[3]: 3840x2160 synthetic pair, 20k ORB features (O(N^2) matcher stress)."""
    from visual_slam_amd import _lib, synth
    a, b, K, R, t = synth.make_pair(1, 3840, 2160)
    p = oracle.default_params(nfeatures=20000)
    k1, f1 = oracle.orb_detect_compute(a, p)
    k2, f2 = oracle.orb_detect_compute(b, p)
    assert len(k1) >= 20000 and len(k2) >= 20000
    c = _lib.Context(max_w=3840, max_h=2160, nfeatures=20000)
    g1, d1 = c.orb_detect_compute(a)
    g2, d2 = c.orb_detect_compute(b)
    assert np.array_equal(g1.view(np.uint8), k1.view(np.uint8)) and np.array_equal(d1, f1)
    assert np.array_equal(g2.view(np.uint8), k2.view(np.uint8)) and np.array_equal(d2, f2)
    oi1, od1, oi2, od2 = oracle.match_knn2(f1, f2)
    gi1, gd1, gi2, gd2 = c.match_knn2(d1, d2)
    assert np.array_equal(gi1, oi1) and np.array_equal(gd1, od1) and np.array_equal(gi2, oi2) and np.array_equal(gd2, od2)
    res = c.pairs_run(a[None], b[None], K)[0]
    rc, T, counts = oracle.pair_run(a, b, K, p=p)
    c.close()
    assert rc == 0 and res["status"] == 0
    assert [res["n_kp1"], res["n_kp2"], res["n_match"], res["n_inlier"]] == counts.tolist()
    assert rot_angle(T[:3, :3], res["T"][:3, :3]) < 1e-6 and dir_angle(T[:3, 3], res["T"][:3, 3]) < 1e-6


def test_batching_is_invisible(K02):
    """Size-independent properties on a ragged batch: results do not depend on chunk_pairs / max_pairs, repeated pairs
    give identical records (determinism), and two contexts driven from two host threads agree with one."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from visual_slam_amd import _lib, synth
    a, b = synth.make_batch(range(6), workers=1)
    reps = 7                                             # 42 pairs: not a multiple of any chunk size below
    A = np.concatenate([a] * reps); B = np.concatenate([b] * reps)
    n, h, w = A.shape
    K = synth.K_KITTI02
    dA = torch.from_numpy(A).cuda(); dB = torch.from_numpy(B).cuda()

    def run(chunk, maxp, lo=0, hi=n):
        c = _lib.Context(max_w=w, max_h=h, chunk_pairs=chunk, max_pairs=maxp)
        try:
            return c.pairs_run_device(dA.data_ptr() + lo * h * w, dB.data_ptr() + lo * h * w, hi - lo, w, h, w, K)
        finally:
            c.close()

    ref = run(8, n)
    assert (ref["status"] == 0).all()
    for f in ref.dtype.names:                            # determinism: the 7 repetitions are identical
        v = ref[f].reshape(reps, 6, *ref[f].shape[1:])
        assert all(np.array_equal(v[0], v[r]) for r in range(1, reps)), f
    for chunk, maxp in ((5, 16), (32, 42), (1, 3)):
        got = run(chunk, maxp, 0, 18 if chunk == 1 else n)
        for f in ref.dtype.names:
            assert np.array_equal(got[f], ref[f][: len(got)]), (chunk, maxp, f)
    with ThreadPoolExecutor(2) as ex:                    # two contexts / streams at once
        parts = list(ex.map(lambda r: run(8, 21, r[0], r[1]), [(0, 21), (21, 42)]))
    both = np.concatenate(parts)
    for f in ref.dtype.names:
        assert np.array_equal(both[f], ref[f]), f


def test_many_synthetic_pairs_match_oracle(oracle):
    """A sweep of seeded config-2 pairs (the bench workload) through the batched path vs the oracle, pair by pair:
    status, keypoint / match / inlier counts exactly, pose within 1e-6 rad.  Catches rare-path divergences
    (retainBest ties, ragged matcher blocks, second RANSAC rounds) that single fixtures cannot."""
    from concurrent.futures import ThreadPoolExecutor
    from visual_slam_amd import _lib, synth
    seeds = list(range(100, 196))
    a, b = synth.make_batch(seeds, workers=1)
    K = synth.K_KITTI02
    c = _lib.Context(max_w=a.shape[2], max_h=a.shape[1], chunk_pairs=7, max_pairs=len(seeds))
    try:
        res = c.pairs_run(a, b, K)
    finally:
        c.close()
    with ThreadPoolExecutor(8) as ex:          # ctypes releases the GIL: the oracle runs on 8 host threads
        ref = list(ex.map(lambda i: oracle.pair_run(a[i], b[i], K), range(len(seeds))))
    for i, (rc, T, counts) in enumerate(ref):
        r = res[i]
        assert rc == r["status"] == 0, seeds[i]
        assert [r["n_kp1"], r["n_kp2"], r["n_match"], r["n_inlier"]] == counts.tolist(), seeds[i]
        assert rot_angle(T[:3, :3], r["T"][:3, :3]) < 1e-6 and dir_angle(T[:3, 3], r["T"][:3, 3]) < 1e-6, seeds[i]


def test_device_frames_with_stride_and_odd_base(K02):
    """vo_pairs_run_device / vo_sequence_run_device read level 0 in place: frames whose rows are `stride` bytes apart
    (odd, so rows and the base pointer are unaligned) give the records of the dense layout."""
    import torch
    from visual_slam_amd import _lib, synth
    a, b = synth.make_batch(range(3), workers=1)
    n, h, w = a.shape
    K = synth.K_KITTI02
    c = _lib.Context(max_w=w, max_h=h, chunk_pairs=2, max_pairs=n)
    try:
        dA = torch.from_numpy(a).cuda(); dB = torch.from_numpy(b).cuda()
        dense = c.pairs_run_device(dA.data_ptr(), dB.data_ptr(), n, w, h, w, K)
        stride = w + 2                                        # 1243
        pa = torch.full((n * h * stride + 7,), 255, dtype=torch.uint8, device="cuda")
        pb = torch.full((n * h * stride + 7,), 255, dtype=torch.uint8, device="cuda")
        for src, dst in ((dA, pa), (dB, pb)):
            dst[1:1 + n * h * stride].view(n, h, stride)[:, :, :w] = src      # base pointer 1 byte past an aligned address
        strided = c.pairs_run_device(pa.data_ptr() + 1, pb.data_ptr() + 1, n, w, h, stride, K)
        for f in dense.dtype.names:
            assert np.array_equal(dense[f], strided[f]), f
        seq_dense = c.sequence_run_device(dA.data_ptr(), n, w, h, w, K)
        seq_strided = c.sequence_run_device(pa.data_ptr() + 1, n, w, h, stride, K)
        for f in seq_dense.dtype.names:
            assert np.array_equal(seq_dense[f], seq_strided[f]), f
        assert (dense["status"] == 0).all()
    finally:
        c.close()


def test_get_matches_one_call_equals_the_three_calls(ctx, kitti02, kitti09):
    """vo_get_matches (both frames in one launch wave, matched on the device) == detectAndCompute x2 + knnMatch + ratio."""
    for a, b in ((kitti02[0], kitti02[1]), (kitti09[0], kitti09[1])):
        from visual_slam_amd import _lib
        c = _lib.Context(max_w=a.shape[1], max_h=a.shape[0])
        k1, f1 = c.orb_detect_compute(a); k2, f2 = c.orb_detect_compute(b)
        qi, ti, _, _ = c.match_knn2_ratio(f1, f2, 0.45)
        q1, q2 = c.get_matches(a, b, 0.45)
        c.close()
        assert len(q1) == len(qi) > 100
        assert np.array_equal(q1, np.stack([k1["x"][qi], k1["y"][qi]], 1)) and np.array_equal(q2, np.stack([k2["x"][ti], k2["y"][ti]], 1))
    flat = np.full_like(kitti02[0], 77)
    with pytest.raises(_lib.VoError) as e:
        ctx.get_matches(kitti02[0], flat)
    assert e.value.code == _lib.VO_ERR_TOO_FEW


def _check_points(oracle, K, poses_prev, poses_cur, X, q1, q2, off):
    worst = 0.0
    for p in range(len(off) - 1):
        a, b = int(off[p]), int(off[p + 1])
        if a == b:
            continue
        Xo = oracle.triangulate_f32(K @ poses_prev[p][:3], K @ poses_cur[p][:3], q1[a:b], q2[a:b])
        ref = (Xo / Xo[3])[:3].T                                    # libs/vo_utils.py:161-164 in float32
        worst = max(worst, float(np.abs(X[a:b] - ref).max() / np.abs(ref).max()))
    return worst


def test_batched_triangulation_sequence_and_pairs(oracle, kitti02, K02):
    """Row T in the batched path (vo.py:80-93): every match of every pair in ONE launch, against the oracle's
    triangulatePoints on the same correspondences and projection matrices, relative error < 1e-5 (north star)."""
    from conftest import GOLDEN
    from visual_slam_amd import _lib, synth, trajectory as tj
    from visual_slam_amd.vo_utils import VisualOdometry
    # (a) the bundled KITTI 02 frames as a sequence, poses chained as vo.py:80 does
    vo = VisualOdometry(os.path.join(GOLDEN, "kitti02"), False)
    out = tj.run_sequence(vo, triangulate=True, wave=2, chunk=1)     # two pose waves, three lanes
    off = out["point_offsets"]
    assert len(off) == 4 and off[-1] == len(out["points"]) == out["results"]["n_match"].sum()
    prevs = [np.eye(4)] + out["poses"][:-1]
    assert _check_points(oracle, vo.K, prevs, out["poses"], out["points"], out["q1"], out["q2"], off) < 1e-5
    # the drop-in's per-match call gives the same rows (vo.py:92)
    p0 = vo.triangulate(out["q1"][0], out["q2"][0], np.eye(4), out["poses"][0])
    assert np.allclose(p0, out["points"][0], rtol=1e-6, atol=0)
    # (b) 96 independent synthetic pairs: prev = identity, cur = inv(T) per pair
    A, B = synth.make_batch(range(500, 596), workers=4)
    K = synth.K_KITTI02
    ctx = _lib.Context(max_w=1241, max_h=376, chunk_pairs=16, max_pairs=96)
    res = ctx.pairs_run(A, B, K)
    assert (res["status"] == 0).all()
    curs = [np.linalg.inv(t) for t in res["T"]]
    P1 = np.array([K @ np.eye(4)[:3]] * 96); P2 = np.array([K @ c[:3] for c in curs])
    X, q1, q2, off = ctx.pairs_triangulate(P1, P2)
    assert np.array_equal(np.diff(off), res["n_match"]) and len(X) == off[-1] > 96 * 100
    assert _check_points(oracle, K, [np.eye(4)] * 96, curs, X, q1, q2, off) < 1e-5
    # a failed pair owns no rows
    flat = np.full_like(A[0], 50)
    res = ctx.pairs_run(np.stack([A[0], flat, A[1]]), np.stack([B[0], flat, B[1]]), K)
    assert res["status"][1] != 0
    X2, _, _, off2 = ctx.pairs_triangulate(P1[:3], P2[[0, 1, 1]])
    assert off2[1] == off2[2] and off2[3] - off2[2] == res["n_match"][2] and np.array_equal(X2[: off2[1]], X[: off[1]])
    ctx.close()


def test_full_size_batch_is_a_permutation_of_its_pairs():
    """BASELINE.json configs[2] at full size (4096 pairs, one context, six lanes, 128 chunks): the batch is 16 distinct
    synthetic pairs in a random order, so every record must equal the record of its pair computed alone -- wherever in the
    batch, on whichever lane and in whichever chunk it ran -- for the synchronous call and for submit / wait."""
    import torch
    from visual_slam_amd import _lib, synth
    a, b = synth.make_batch(range(300, 316), workers=4)
    n, h, w = a.shape
    K = synth.K_KITTI02
    small = _lib.Context(max_w=w, max_h=h, chunk_pairs=1, max_pairs=1)
    alone = np.concatenate([small.pairs_run(a[i:i + 1], b[i:i + 1], K) for i in range(n)])
    small.close()
    assert (alone["status"] == 0).all()
    rng = np.random.default_rng(11)
    idx = rng.integers(0, n, 4096)
    tA, tB = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    ti = torch.from_numpy(idx).cuda()
    dA, dB = tA[ti].contiguous(), tB[ti].contiguous()
    ctx = _lib.Context(max_w=w, max_h=h, chunk_pairs=32, max_pairs=4096)
    sync = ctx.pairs_run_device(dA.data_ptr(), dB.data_ptr(), 4096, w, h, w, K)
    ctx.pairs_submit_device(dA.data_ptr(), dB.data_ptr(), 4096, w, h, w, K)
    ctx.pairs_submit_device(dA.data_ptr() + 1000 * h * w, dB.data_ptr() + 1000 * h * w, 3096, w, h, w, K)
    first, second = ctx.pairs_wait(), ctx.pairs_wait()
    ctx.close()
    for f in alone.dtype.names:
        assert np.array_equal(sync[f], alone[f][idx]), f
        assert np.array_equal(first[f], alone[f][idx]), f
        assert np.array_equal(second[f], alone[f][idx[1000:]]), f


def test_both_solver_forms_give_the_same_records(K02):
    """A pose wave of few pairs runs the 16-lanes-per-hypothesis five-point solver, a large one the one-lane kernel (switch at
    2048 hypotheses per RANSAC round, vo_pose.hip): the same pairs must come out bit for bit the same either way."""
    import torch
    from visual_slam_amd import _lib, synth
    a, b = synth.make_batch(range(6), workers=1)
    reps = 27                                            # 162 pairs x 16 first-round hypotheses = 2592 > 2048
    A = np.concatenate([a] * reps); B = np.concatenate([b] * reps)
    n, h, w = A.shape
    K = synth.K_KITTI02
    dA = torch.from_numpy(A).cuda(); dB = torch.from_numpy(B).cuda()
    big = _lib.Context(max_w=w, max_h=h, chunk_pairs=32, max_pairs=n)
    small = _lib.Context(max_w=w, max_h=h, chunk_pairs=6, max_pairs=6)
    try:
        one_lane = big.pairs_run_device(dA.data_ptr(), dB.data_ptr(), n, w, h, w, K)
        wide = small.pairs_run_device(dA.data_ptr(), dB.data_ptr(), 6, w, h, w, K)
    finally:
        big.close(); small.close()
    assert (wide["status"] == 0).all()
    for f in wide.dtype.names:
        v = one_lane[f].reshape(reps, 6, *one_lane[f].shape[1:])
        assert all(np.array_equal(v[r], wide[f]) for r in range(reps)), f


def test_get_pose_one_call_equals_the_two_calls(ctx, K02):
    for seed, n, noise, outl in [(0, 200, 0.0, 0), (3, 400, 0.3, 120), (11, 300, 0.2, 200)]:
        x1, x2, *_ = two_view(seed, n, K02, noise=noise, outliers=outl)
        E, mask, ninl = ctx.find_essential_ransac(x1, x2, K02)
        R, t, ng = ctx.recover_pose(E, x1, x2, K02)
        T, ni2, ng2 = ctx.get_pose(x1, x2, K02)
        assert (ni2, ng2) == (ninl, ng)
        assert np.array_equal(T[:3, :3], R) and np.array_equal(T[:3, 3], t.reshape(3)) and np.array_equal(T[3], [0, 0, 0, 1])
    from visual_slam_amd._lib import VoError, VO_ERR_TOO_FEW
    x1, x2, *_ = two_view(0, 4, K02)
    with pytest.raises(VoError) as e:
        ctx.get_pose(x1, x2, K02)
    assert e.value.code == VO_ERR_TOO_FEW
