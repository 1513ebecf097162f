"""Live pin of the oracle against the real thing (SURVEY.md 8(c) "Live cv2 check").

opencv-python (4.8.0.76 in the reference's requirements.txt:2) is NOT installed in this image, so here
every test below is skipped and the oracle stays "parity unpinned".  On any box where `import cv2`
works these tests run by themselves (they are ordinary CPU tests) and compare the oracle with cv2 on
the bundled KITTI frames, call for call as /root/reference/libs/vo_utils.py makes them.  Nothing is
ever fetched or installed.  The ORB test enumerates the full product of the switchable variants
(blur_mode x fma_atan x fma_blur x trig_mode, 16 tuples) and reports every tuple that reproduces cv2,
which is the information needed to re-pin the defaults.
"""
import itertools

import numpy as np
import pytest

cv2 = pytest.importorskip("cv2", reason="opencv-python is not installed here (SURVEY.md 8(c)): the oracle stays unpinned")


def _cv_kps(kps):
    return (np.array([[k.pt[0], k.pt[1], k.size, k.angle, k.response] for k in kps], np.float32).reshape(-1, 5),
            np.array([k.octave for k in kps], np.int32))


def _orb_equal(oracle, img, **variant):
    kps, desc = cv2.ORB_create(3000).detectAndCompute(img, None)      # libs/vo_utils.py:23,94-95
    ko, do = oracle.orb_detect_compute(img, oracle.default_params(**variant))
    if len(kps) != len(ko):
        return False
    f, o = _cv_kps(kps)
    mine = np.stack([ko["x"], ko["y"], ko["size"], ko["angle"], ko["response"]], 1)
    return np.array_equal(f, mine) and np.array_equal(o, ko["octave"]) and np.array_equal(desc, do)


def test_versions_are_reported():
    print("cv2", cv2.__version__)
    print(cv2.getBuildInformation()[:2000])


@pytest.mark.parametrize("idx", [0, 1, 2, 3])
def test_orb_detect_compute_bit_exact(oracle, kitti02, idx):
    img = kitti02[idx]
    names = ("blur_mode", "fma_atan", "fma_blur", "trig_mode")
    ok = [dict(zip(names, t)) for t in itertools.product((0, 1), repeat=4) if _orb_equal(oracle, img, **dict(zip(names, t)))]
    print("cv2 %s frame %d: oracle variants that reproduce detectAndCompute bit for bit: %s" % (cv2.__version__, idx, ok or "none"))
    if not _orb_equal(oracle, img):
        pytest.fail("oracle defaults differ from cv2 %s on frame %d; variants that DO match: %s" % (cv2.__version__, idx, ok or "none"))


def test_bfmatcher_knn_and_ratio(oracle, kitti02):
    orb = cv2.ORB_create(3000)
    _, f1 = orb.detectAndCompute(kitti02[0], None)
    _, f2 = orb.detectAndCompute(kitti02[1], None)
    matches = cv2.BFMatcher(cv2.NORM_HAMMING).knnMatch(f1, f2, k=2)   # north-star matcher
    i1, d1, i2, d2 = oracle.match_knn2(f1, f2)
    assert [m[0].trainIdx for m in matches] == i1.tolist() and [m[1].trainIdx for m in matches] == i2.tolist()
    assert [int(m[0].distance) for m in matches] == d1.tolist() and [int(m[1].distance) for m in matches] == d2.tolist()
    good = [k for k, (m, n) in enumerate(matches) if m.distance < 0.45 * n.distance]   # libs/vo_utils.py:101-104
    assert good == oracle.ratio_filter(d1, d2, 0.45).tolist()


def _matched_points(oracle, a, b):
    k1, f1 = oracle.orb_detect_compute(a)
    k2, f2 = oracle.orb_detect_compute(b)
    i1, d1, i2, d2 = oracle.match_knn2(f1, f2)
    keep = oracle.ratio_filter(d1, d2, 0.45)
    q1 = np.stack([k1["x"][keep], k1["y"][keep]], 1).astype(np.float32)
    q2 = np.stack([k2["x"][i1[keep]], k2["y"][i1[keep]]], 1).astype(np.float32)
    return q1, q2


def test_find_essential_and_recover_pose(oracle, kitti02, K02):
    """findEssentialMat against the oracle's pose-side rounding variants (hypot_mode x ainv_mode, oracle.set_pose_variant):
    the tuples that reproduce cv2's inlier mask and E are printed, exactly as the ORB test does for its flags; the HIP
    solver implements (0, 0) and follows whichever tuple a cv2 box names."""
    q1, q2 = _matched_points(oracle, kitti02[0], kitti02[1])
    E, mask = cv2.findEssentialMat(q1, q2, K02, cv2.RANSAC, 0.999, 1.0, None)     # libs/vo_utils.py:141
    ok, bits = [], []
    for hm, am in itertools.product((0, 1), repeat=2):
        with oracle.pose_variant(hypot_mode=hm, ainv_mode=am):
            Ev, mv, _, _ = oracle.find_essential_ransac(q1, q2, K02)
        if np.array_equal(mask.ravel().astype(bool), np.asarray(mv).ravel().astype(bool)) and np.abs(E - Ev).max() < 1e-9:
            ok.append({"hypot_mode": hm, "ainv_mode": am})
            if np.array_equal(E, Ev):
                bits.append({"hypot_mode": hm, "ainv_mode": am})
    print("cv2 %s: pose variants that reproduce findEssentialMat's mask and E (1e-9): %s; bit for bit: %s" % (cv2.__version__, ok or "none", bits or "none"))
    Eo, mo, no, _ = oracle.find_essential_ransac(q1, q2, K02)
    assert np.array_equal(mask.ravel().astype(bool), np.asarray(mo).ravel().astype(bool)), "defaults differ; variants that match: %s" % (ok or "none")
    assert np.abs(E - Eo).max() < 1e-9
    _, R, t, _ = cv2.recoverPose(E, q1, q2, K02, mask)                             # libs/vo_utils.py:143
    Ro, to, go = oracle.recover_pose(E, q1, q2, K02)
    assert np.abs(R - Ro).max() < 1e-9 and np.abs(t.ravel() - np.asarray(to).ravel()).max() < 1e-9


def test_triangulate_points(oracle, kitti02, K02):
    q1, q2 = _matched_points(oracle, kitti02[0], kitti02[1])
    P1 = K02 @ np.eye(4)[:3]
    T = np.eye(4); T[2, 3] = -1.0
    P2 = K02 @ T[:3]
    X = cv2.triangulatePoints(P1, P2, q1.T, q2.T)                                  # libs/vo_utils.py:160
    Xo = oracle.triangulate_f32(P1, P2, q1, q2)
    assert np.abs(np.abs(X) - np.abs(Xo)).max() < 1e-5
