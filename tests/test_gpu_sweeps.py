"""Bounded versions of the one-off parity sweeps (tools/sweep_pose.py, sweep_sizes.py, sweep_real.py) inside the
driver-run GPU suite: seeded two-view problems, random frame sizes, the KITTI-09 pair with its own intrinsics, and the
keypoint-capacity cliff.  Everything through the C ABI against the CPU oracle (parity unpinned: the oracle restates
OpenCV 4.8.0)."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from conftest import GOLDEN
from test_oracle_pose import rodrigues, rot_angle, dir_angle

pytestmark = pytest.mark.gpu

N_POSE = 150
N_SIZES = 40


def _threads():
    n = os.cpu_count() or 1
    env = os.environ.get("OMP_NUM_THREADS", "")
    if env.isdigit() and int(env) > 0:
        n = min(n, int(env))
    return max(1, min(16, n))


def _scene(rng, K, n=None, kind=None, outl=None, noise=None):
    """One two-view problem (cf. tools/sweep_pose.py): general / planar / pure-rotation / forward-motion scenes,
    5..3000 correspondences, up to 85 % outliers, pixel noise up to 1.5."""
    n = int(rng.choice([5, 6, 7, 8, 12, 30, 100, 300, 700, 1500, 3000])) if n is None else n
    kind = str(rng.choice(["general", "planar", "rotation", "forward"])) if kind is None else kind
    X = np.stack([rng.uniform(-8, 8, n), rng.uniform(-3, 3, n), rng.uniform(6, 40, n)], 1)
    if kind == "planar":
        X[:, 2] = 20 + 0.1 * X[:, 0]
    R = rodrigues(rng.normal(0, 0.05, 3))
    t = rng.normal(0, 1, 3); t /= np.linalg.norm(t)
    if kind == "rotation":
        t = t * 1e-9
    if kind == "forward":
        t = np.array([0.02, -0.01, -1.0]); t /= np.linalg.norm(t)
    x1 = (K @ X.T).T; x1 = x1[:, :2] / x1[:, 2:]
    X2 = (R @ X.T).T + t
    x2 = (K @ X2.T).T; x2 = x2[:, :2] / x2[:, 2:]
    noise = float(rng.choice([0, 0.1, 0.5, 1.5])) if noise is None else noise
    if noise:
        x1 = x1 + rng.normal(0, noise, x1.shape); x2 = x2 + rng.normal(0, noise, x2.shape)
    frac = float(rng.choice([0, 0.1, 0.3, 0.6, 0.85])) if outl is None else outl
    no = int(n * frac) if n > 8 else 0
    if no:
        x2[:no] = rng.uniform(0, 1200, (no, 2))
    return x1.astype(np.float32), x2.astype(np.float32), (kind, n, no, noise)


def test_pose_sweep_against_the_oracle(oracle, K02):
    """findEssentialMat + recoverPose + triangulatePoints (libs/vo_utils.py:141-143,160) on 150 seeded problems: inlier
    masks and counts exact, E / R / t to 1e-9, triangulated points to 1e-5; some problems need several RANSAC rounds
    (more than the 16 + 64 hypotheses of the first two), some have no model at all on either side."""
    from visual_slam_amd import _lib
    rng = np.random.default_rng(2025)
    cases = [_scene(rng, K02) for _ in range(N_POSE - 10)]
    # forced corners of the space: every scene kind at 85 % outliers (hundreds of iterations), the minimum sample, a large clean set
    for kind in ("general", "planar", "rotation", "forward"):
        cases.append(_scene(rng, K02, n=300, kind=kind, outl=0.85, noise=0.5))
    cases += [_scene(rng, K02, n=5, kind="general"), _scene(rng, K02, n=3000, kind="general", outl=0.0, noise=0.0)]
    cases += [_scene(rng, K02, n=700, kind=k, outl=0.6, noise=1.5) for k in ("general", "planar", "rotation", "forward")]
    assert len(cases) == N_POSE

    def ref(c):
        try:
            E, m, ni, it = oracle.find_essential_ransac(c[0], c[1], K02)
        except RuntimeError:
            return None
        R, t, g = oracle.recover_pose(E, c[0], c[1], K02)
        return E, m, ni, it, R, t, g, oracle.triangulate_f32(K02 @ np.eye(4)[:3], K02 @ np.c_[R, t], c[0], c[1])

    with ThreadPoolExecutor(_threads()) as ex:
        refs = list(ex.map(ref, cases))
    ctx = _lib.Context(max_w=1241, max_h=376, chunk_pairs=1, max_pairs=1)
    many_rounds = 0
    try:
        for c, r in zip(cases, refs):
            tag = c[2]
            try:
                E, m, ni = ctx.find_essential_ransac(c[0], c[1], K02)
            except _lib.VoError:
                assert r is None, tag
                continue
            assert r is not None, tag
            Eo, mo, no_, it, Ro, to, go, Xo = r
            many_rounds += it > 80
            assert ni == no_ and np.array_equal(m, mo), (tag, it)
            assert np.abs(E - Eo).max() < 1e-9, tag
            Rg, tg, gg = ctx.recover_pose(E, c[0], c[1], K02)
            assert gg == go, tag
            assert np.abs(Rg - Ro).max() < 1e-9 and np.abs(np.ravel(tg) - np.ravel(to)).max() < 1e-9, tag
            Xg = ctx.triangulate(K02 @ np.eye(4)[:3], K02 @ np.c_[Ro, to], c[0], c[1])
            assert np.abs(np.abs(Xg) - np.abs(Xo)).max() < 1e-5, tag
            # the one-call form (get_pose of libs/vo_utils.py:136-145) must land on the same numbers
            T, ni2, ng2 = ctx.get_pose(c[0], c[1], K02)
            assert (ni2, ng2) == (ni, gg) and np.array_equal(T[:3, :3], Rg) and np.array_equal(T[:3, 3], np.ravel(tg)), tag
    finally:
        ctx.close()
    assert many_rounds >= 8          # the sweep does reach past the first two RANSAC rounds


def test_frame_size_sweep_against_the_oracle(oracle, kitti02, kitti09):
    """detectAndCompute (libs/vo_utils.py:94-95) on 40 frame sizes between 64x64 and 1400x640 (odd widths, ragged last
    tiles, levels that lose their interior) and on flipped / negated / cropped real frames: keypoint records and
    descriptors bit for bit."""
    from visual_slam_amd import _lib, synth
    rng = np.random.default_rng(7)
    sizes = [(int(rng.integers(64, 1400)), int(rng.integers(64, 640))) for _ in range(N_SIZES - 10)]
    sizes[:6] = [(64, 64), (65, 67), (1399, 639), (1400, 64), (64, 640), (333, 111)]
    imgs = []
    for k, (w, h) in enumerate(sizes):
        big = synth.frame_a(np.random.default_rng(1000 + k), max(w, 256), max(h, 256))
        imgs.append(np.ascontiguousarray(big[:h, :w]))
    g2, g9 = kitti02[2], kitti09[1]
    real = [g2[:, ::-1], g2[::-1], 255 - g9, (g9 * 0.6 + 40).astype(np.uint8), np.clip(g2.astype(np.int32) * 2 - 100, 0, 255).astype(np.uint8)]
    for v in real:
        v = np.ascontiguousarray(v)
        imgs.append(v)
        x0, y0 = int(rng.integers(0, 600)), int(rng.integers(0, 150))
        cw, ch = int(rng.integers(64, v.shape[1] - x0)), int(rng.integers(64, v.shape[0] - y0))
        imgs.append(np.ascontiguousarray(v[y0:y0 + ch, x0:x0 + cw]))
    assert len(imgs) == N_SIZES
    with ThreadPoolExecutor(_threads()) as ex:
        ref = list(ex.map(lambda im: oracle.orb_detect_compute(im), imgs))
    ctx = _lib.Context(max_w=1400, max_h=640, chunk_pairs=1, max_pairs=1)
    try:
        total = 0
        for im, (ko, do) in zip(imgs, ref):
            kg, dg = ctx.orb_detect_compute(im)
            assert len(kg) == len(ko), im.shape
            assert all(np.array_equal(kg[f], ko[f]) for f in ko.dtype.names), im.shape
            assert np.array_equal(dg, do), im.shape
            total += len(kg)
    finally:
        ctx.close()
    assert total > 20000


def test_batch_contexts_on_odd_sizes_against_the_oracle(oracle, kitti02):
    """Batch contexts (several lanes; the blur there only computes the tiles a descriptor patch reaches, and the pair records come
    from the multi-lane path) on odd frame sizes: every pair record against the oracle's -- keypoint, match and inlier counts
    exact, which they can only be if every sampled blurred pixel is right -- on synthetic frames and on crops of a real
    frame with large keypoint-free regions (most blur tiles skipped)."""
    from visual_slam_amd import _lib, synth
    K = synth.K_KITTI02
    cases = []
    for k, (w, h) in enumerate([(333, 201), (641, 129), (1023, 257), (450, 376)]):
        a = np.stack([np.ascontiguousarray(synth.frame_a(np.random.default_rng(500 + 10 * k + j), max(w, 256), max(h, 256))[:h, :w]) for j in range(5)])
        b = np.roll(a, 3, axis=2)
        cases.append((a, b))
    g = kitti02[0]
    flat = g.copy(); flat[:, 400:900] = 90; flat[:150] = 120            # keypoints only in two corners of the frame
    real = np.stack([flat[y0:y0 + 300, x0:x0 + 1100] for (x0, y0) in [(0, 0), (64, 40), (100, 76), (141, 33), (17, 5)]])
    cases.append((np.ascontiguousarray(real), np.ascontiguousarray(np.roll(real, 2, axis=2))))
    for a, b in cases:
        n, h, w = a.shape
        c = _lib.Context(max_w=w, max_h=h, chunk_pairs=2, max_pairs=n, nfeatures=1500)
        try:
            res = c.pairs_run(a, b, K)
        finally:
            c.close()
        with ThreadPoolExecutor(_threads()) as ex:
            ref = list(ex.map(lambda i: oracle.pair_run(a[i], b[i], K, p=oracle.default_params(nfeatures=1500)), range(n)))
        for i, (rc, T, counts) in enumerate(ref):
            r = res[i]
            assert rc == r["status"], (w, h, i, rc, r["status"])
            assert [r["n_kp1"], r["n_kp2"], r["n_match"], r["n_inlier"]] == counts.tolist(), (w, h, i)
            if rc == 0:
                assert np.abs(T - r["T"]).max() < 1e-6, (w, h, i)


def test_kitti09_pair_pose_with_its_own_intrinsics(oracle, kitti09):
    """Sequence 09 has its own calibration (assets/test_imgs/09/calib.txt:1: fx 707.0912, 1226x370), hence its own RANSAC
    threshold (thr^2 = 2.0000882e-06): get_matches -> get_pose on frames 0, 1 against the oracle, and against the
    ground-truth relative pose (plausibility gate)."""
    from visual_slam_amd import _lib, trajectory as tj
    from visual_slam_amd.vo_utils import VisualOdometry
    d = os.path.join(GOLDEN, "kitti09")
    K, _ = VisualOdometry.load_calib(os.path.join(d, "calib.txt"))
    assert abs(K[0, 0] - 707.0912) < 1e-9 and kitti09[0].shape == (370, 1226)
    a, b = kitti09
    ctx = _lib.Context(max_w=a.shape[1], max_h=a.shape[0], chunk_pairs=1, max_pairs=1)
    try:
        q1, q2 = ctx.get_matches(a, b, 0.45)
        T, ninl, ngood = ctx.get_pose(q1, q2, K)
        E, mask, ni = ctx.find_essential_ransac(q1, q2, K)
        rec = ctx.pairs_run(a[None], b[None], K)[0]
    finally:
        ctx.close()
    # oracle: the same chain call for call
    k1, f1 = oracle.orb_detect_compute(a); k2, f2 = oracle.orb_detect_compute(b)
    i1, d1, i2, d2 = oracle.match_knn2(f1, f2)
    keep = oracle.ratio_filter(d1, d2, 0.45)
    o1 = np.stack([k1["x"][keep], k1["y"][keep]], 1); o2 = np.stack([k2["x"][i1[keep]], k2["y"][i1[keep]]], 1)
    assert np.array_equal(q1, o1) and np.array_equal(q2, o2) and len(q1) > 100
    Eo, mo, no_, _ = oracle.find_essential_ransac(o1, o2, K)
    Ro, to, go = oracle.recover_pose(Eo, o1, o2, K)
    assert ni == no_ == ninl and np.array_equal(mask, mo) and np.abs(E - Eo).max() < 1e-9
    assert ngood == go and rot_angle(T[:3, :3], Ro) < 1e-6 and dir_angle(T[:3, 3], np.ravel(to)) < 1e-6
    assert rec["status"] == 0 and [rec["n_match"], rec["n_inlier"], rec["n_good"]] == [len(q1), ninl, ngood]
    assert np.array_equal(rec["T"], T)
    # ground truth of sequence 09 (poses.txt): rotation within 1e-2 rad; the step is only 0.29 m and E is a minimal-sample model
    # (no refit), so the translation direction is a loose gate (0.23 rad here, identical for the oracle)
    gt = VisualOdometry.load_poses(os.path.join(d, "poses.txt"))
    Tgt = tj.relative_gt(gt, 1)
    assert tj.rotation_angle(T[:3, :3], Tgt[:3, :3]) < 1e-2 and tj.direction_angle(T[:3, 3], Tgt[:3, 3]) < 0.35


def test_more_response_ties_than_the_keypoint_capacity_is_an_error(oracle):
    """retainBest keeps every keypoint that ties with the n-th best (KeyPointsFilter::retainBest), so a frame of identical
    corners returns far more than nfeatures keypoints in cv2.  The library holds nfeatures + 1024 per frame and reports a
    frame beyond that as VO_ERR_CAPACITY -- loudly, never truncated (a behavioural difference from
    libs/vo_utils.py:94-95, which would return them all)."""
    from visual_slam_amd import _lib
    h, w = 376, 1241
    img = np.full((h, w), 40, np.uint8)
    img[4::8, 4::8] = 220                       # isolated one-pixel dots on an 8-px grid: identical FAST score and Harris response
    ko, do = oracle.orb_detect_compute(img)
    ctx = _lib.Context(max_w=w, max_h=h, chunk_pairs=1, max_pairs=1)
    try:
        assert len(ko) > ctx.kp_cap             # the premise: the reference semantics give more keypoints than the workspace holds
        with pytest.raises(_lib.VoError) as e:
            ctx.orb_detect_compute(img)
        assert e.value.code == _lib.VO_ERR_CAPACITY
        rec = ctx.pairs_run(img[None], img[None], np.array([[700.0, 0, 620], [0, 700.0, 188], [0, 0, 1]]))[0]
        assert rec["status"] == _lib.VO_ERR_CAPACITY
        # the context stays usable and exact afterwards
        ok = np.full((h, w), 40, np.uint8); ok[4::8, 4::64] = 220
        kg, dg = ctx.orb_detect_compute(ok)
        k2, d2 = oracle.orb_detect_compute(ok)
        assert len(kg) == len(k2) and all(np.array_equal(kg[f], k2[f]) for f in k2.dtype.names) and np.array_equal(dg, d2)
    finally:
        ctx.close()


def test_a_larger_keypoint_capacity_returns_what_cv2_would(oracle):
    """The same frame of identical corners with vo_orb_params.keypoint_capacity raised: every tied keypoint comes back, bit
    for bit the oracle's (= cv2's retainBest semantics); and the drop-in class does that retry by itself, so
    get_matches behaves like libs/vo_utils.py:94-108 on such frames too."""
    from visual_slam_amd import _lib
    from visual_slam_amd.vo_utils import VisualOdometry
    h, w = 376, 1241
    img = np.full((h, w), 40, np.uint8)
    img[4::8, 4::8] = 220
    img[:, w // 2:] = np.where(img[:, w // 2:] == 220, 180, 40)      # two kinds of corners, so that descriptors can tell frames apart a little
    ko, do = oracle.orb_detect_compute(img)
    assert len(ko) > 3000 + 1024
    ctx = _lib.Context(max_w=w, max_h=h, chunk_pairs=1, max_pairs=1, keypoint_capacity=12000)
    try:
        assert ctx.kp_cap == 12000
        kg, dg = ctx.orb_detect_compute(img)
        assert len(kg) == len(ko) and all(np.array_equal(kg[f], ko[f]) for f in ko.dtype.names) and np.array_equal(dg, do)
        # the matcher and ratio test on that many descriptors (6 train segments)
        i1, d1, i2, d2 = oracle.match_knn2(do, do)
        g = ctx.match_knn2(dg, dg)
        assert all(np.array_equal(a, b) for a, b in zip(g, (i1, d1, i2, d2)))
    finally:
        ctx.close()
    vo = VisualOdometry.__new__(VisualOdometry)
    vo.K, vo.D, vo.debug, vo._ctx, vo._ctx_shape = np.eye(3), np.zeros(5), False, None, None
    shifted = np.roll(img, 8, axis=1)
    try:
        q1, q2 = vo.get_matches(1, "ORB", shifted, img)              # raises VO_ERR_CAPACITY internally, retries with a larger workspace
    except ValueError:
        q1 = q2 = np.zeros((0, 2), np.float32)                        # (no match may survive the ratio test on identical corners)
    assert vo._ctx.kp_cap > 3000 + 1024 and q1.shape == q2.shape
    k1, f1 = oracle.orb_detect_compute(img); k2, f2 = oracle.orb_detect_compute(shifted)
    a1, b1, a2, b2 = oracle.match_knn2(f1, f2)
    keep = oracle.ratio_filter(b1, b2, 0.45)
    assert len(q1) == len(keep)
    if len(keep):
        assert np.array_equal(q1, np.stack([k1["x"][keep], k1["y"][keep]], 1)) and np.array_equal(q2, np.stack([k2["x"][a1[keep]], k2["y"][a1[keep]]], 1))
    vo._ctx.close()
