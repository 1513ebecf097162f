"""Known-answer tests that pin the CPU oracle's detect stages to the mathematical
definitions (the reference has no tests or fixtures: parity unpinned)."""
import numpy as np
import pytest

CIRCLE = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3),
          (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def fast_bruteforce(img, thr):
    """FAST-9/16 by definition: score = largest t >= thr such that 9 contiguous circle pixels
    are all > v+t or all < v-t (strict); stored score = that t ... OpenCV stores t_max - 1 + ... see below."""
    h, w = img.shape
    im = img.astype(np.int32)
    score = np.zeros((h, w), np.int32)
    ring = np.stack([im[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx] for dx, dy in CIRCLE], 0)  # [16, h-6, w-6]
    v = im[3:h - 3, 3:w - 3]
    d = v[None] - ring  # positive: neighbour darker
    best = np.full(v.shape, -10 ** 6)
    for s in range(16):
        idx = [(s + k) % 16 for k in range(9)]
        best = np.maximum(best, d[idx].min(0))
        best = np.maximum(best, (-d[idx]).min(0))
    # corner iff some arc has all |diff| > thr; cornerScore = max over arcs of min |diff| - 1
    sc = np.where(best > thr, best - 1, 0)
    score[3:h - 3, 3:w - 3] = sc
    return score


def nms_border(score, edge):
    h, w = score.shape
    out = []
    p = np.pad(score, 1)
    c = p[1:-1, 1:-1]
    keep = c > 0
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            if dx == 0 and dy == 0:
                continue
            keep &= c > p[1 + dy:h + 1 + dy, 1 + dx:w + 1 + dx]
    ys, xs = np.nonzero(keep)
    for y, x in zip(ys, xs):
        if edge <= x < w - edge and edge <= y < h - edge:
            out.append((x, y, score[y, x]))
    return np.array(out, np.int32).reshape(-1, 3)


def test_level_geometry(oracle):
    lw, lh, sc, q = oracle.level_geometry(1241, 376)
    assert lw.tolist() == [1241, 1034, 862, 718, 598, 499, 416, 346]
    assert lh.tolist() == [376, 313, 261, 218, 181, 151, 126, 105]
    assert q.tolist() == [652, 543, 452, 377, 314, 262, 218, 182]
    lw, lh, _, _ = oracle.level_geometry(1226, 370)
    assert lw.tolist() == [1226, 1022, 851, 709, 591, 493, 411, 342]
    assert lh.tolist() == [370, 308, 257, 214, 178, 149, 124, 103]
    lw, lh, _, q = oracle.level_geometry(3840, 2160, oracle.default_params(nfeatures=20000))
    assert lw.tolist() == [3840, 3200, 2667, 2222, 1852, 1543, 1286, 1072]
    assert q.tolist() == [4343, 3620, 3016, 2514, 2095, 1746, 1455, 1211]


@pytest.mark.parametrize("seed", [0, 1])
def test_fast_matches_definition(oracle, seed):
    rng = np.random.default_rng(seed)
    # smooth-ish random image so that corners exist but are not everywhere
    base = rng.integers(0, 256, (40, 60)).astype(np.float32)
    img = np.kron(base, np.ones((4, 4), np.float32))
    img = (img + rng.normal(0, 12, img.shape)).clip(0, 255).astype(np.uint8)
    p = oracle.default_params(nlevels=1, nfeatures=100000)
    _, _, dump = oracle.orb_detect_compute(img, p, debug=True)
    n = dump["fast_count"][0]
    got = dump["fast_xys"][:n]
    ref = nms_border(fast_bruteforce(img, 20), 31)
    assert n > 20
    assert np.array_equal(got, ref)


def test_fast_on_kitti(oracle, kitti02):
    img = kitti02[0]
    _, _, dump = oracle.orb_detect_compute(img, debug=True)
    n0 = dump["fast_count"][0]
    ref = nms_border(fast_bruteforce(img, 20), 31)
    assert np.array_equal(dump["fast_xys"][:n0], ref)


def test_pyramid_properties(oracle, kitti02):
    img = kitti02[0]
    _, _, dump = oracle.orb_detect_compute(img, debug=True)
    lw, lh = dump["lw"], dump["lh"]
    ofs = np.concatenate([[0], np.cumsum(lw.astype(np.int64) * lh)])
    lv = [dump["pyr"][ofs[i]:ofs[i + 1]].reshape(lh[i], lw[i]) for i in range(8)]
    assert np.array_equal(lv[0], img)
    # level l is a bilinear resample of level l-1: compare with float bilinear (|diff| <= 1)
    for l in range(1, 8):
        src = lv[l - 1].astype(np.float64)
        sh, sw = src.shape
        dh, dw = lv[l].shape
        fx = (np.arange(dw) + 0.5) * (sw / dw) - 0.5
        fy = (np.arange(dh) + 0.5) * (sh / dh) - 0.5
        x0 = np.floor(fx).astype(int); y0 = np.floor(fy).astype(int)
        ax = fx - x0; ay = fy - y0
        top = src[y0][:, x0] * (1 - ax) + src[y0][:, x0 + 1] * ax
        bot = src[y0 + 1][:, x0] * (1 - ax) + src[y0 + 1][:, x0 + 1] * ax
        ref = top * (1 - ay)[:, None] + bot * ay[:, None]
        assert np.abs(ref - lv[l]).max() <= 1.0
    # a constant image stays constant through the fixed-point resize
    const = np.full((100, 130), 173, np.uint8)
    _, _, d2 = oracle.orb_detect_compute(const, oracle.default_params(nlevels=4), debug=True)
    assert (d2["pyr"] == 173).all()


def test_retain_best_set_and_libstdcxx(oracle):
    rng = np.random.default_rng(3)
    for n, k in [(10, 3), (1000, 100), (5000, 1304), (300, 300), (300, 400), (4, 2)]:
        keys = rng.integers(20, 60, n).astype(np.float32)
        idx = oracle.retain_best(keys, k)
        if n <= k:
            assert np.array_equal(idx, np.arange(n))
            continue
        thr = np.sort(keys)[::-1][k - 1]
        assert set(idx.tolist()) == set(np.nonzero(keys >= thr)[0].tolist())
        assert len(set(idx.tolist())) == len(idx)


def test_harris_and_angle_known_answers(oracle):
    # vertical step edge: Ix large, Iy = 0 -> response = -k*(a)^2*scale^4 < 0; corner > 0
    img = np.zeros((120, 120), np.uint8)
    yy, xx = np.mgrid[0:40, 0:40]
    img[40:80, 40:80] = (150 + xx + yy).astype(np.uint8)  # ramp breaks the score ties under NMS
    p = oracle.default_params(nlevels=1, nfeatures=50)
    kps, desc, dump = oracle.orb_detect_compute(img, p, debug=True)
    assert len(kps) >= 3
    # every detected square corner has a positive Harris response
    assert (kps["response"] > 0).all()
    # the intensity centroid points from the corner into the bright square
    k = kps[np.argmin((kps["x"] - 79) ** 2 + (kps["y"] - 79) ** 2)]
    assert abs(k["angle"] - 225.0) < 6.0
    k = kps[np.argmin((kps["x"] - 79) ** 2 + (kps["y"] - 42) ** 2)]
    assert abs(k["angle"] - 135.0) < 6.0
    k = kps[np.argmin((kps["x"] - 42) ** 2 + (kps["y"] - 79) ** 2)]
    assert abs(k["angle"] - 315.0) < 6.0


def test_blur_modes(oracle, kitti02):
    from scipy.ndimage import correlate1d

    img = kitti02[1][:200, :300].copy()
    x = np.arange(-3, 4)
    g = np.exp(-x * x / 8.0); g /= g.sum()
    ref = correlate1d(correlate1d(img.astype(np.float64), g, axis=1, mode="mirror"), g, axis=0, mode="mirror")
    for mode, fma, tol in [(0, 1, 0.51), (0, 0, 0.51), (1, 0, 1.1)]:
        p = oracle.default_params(nlevels=1, blur_mode=mode, fma_blur=fma)
        _, _, dump = oracle.orb_detect_compute(img, p, debug=True)
        b = dump["blurred"].reshape(img.shape).astype(np.float64)
        # the x256 integer kernel sums to 257, i.e. a gain of (257/256)^2
        r = ref if mode == 0 else np.minimum(ref * (257 / 256.0) ** 2, 255.0)
        assert np.abs(b - r).max() <= tol, (mode, fma)
    # the integer kernel sums to 257: a constant 255 image saturates, 100 becomes 100*257*257>>16
    const = np.full((64, 64), 100, np.uint8)
    _, _, d = oracle.orb_detect_compute(const, oracle.default_params(nlevels=1, blur_mode=1), debug=True)
    assert (d["blurred"] == ((100 * 257 * 257 + 32768) >> 16)).all()


def test_descriptor_definition(oracle, kitti02):
    """Recompute rBRIEF for a few keypoints from the blurred level dump with numpy."""
    import re, os
    img = kitti02[0]
    kps, desc, dump = oracle.orb_detect_compute(img, debug=True)
    lw, lh = dump["lw"], dump["lh"]
    ofs = np.concatenate([[0], np.cumsum(lw.astype(np.int64) * lh)])
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "vo_orb_pattern_31.h")).read()
    body = hdr[hdr.index("{") + 1: hdr.rindex("}")]
    pat = np.array([int(v) for v in re.findall(r"-?\d+", body)], np.int32).reshape(512, 2)
    _, _, sc, _ = oracle.level_geometry(img.shape[1], img.shape[0])
    for i in list(range(0, len(kps), 97)):
        k = kps[i]
        l = int(k["octave"])
        blur = dump["blurred"][ofs[l]:ofs[l + 1]].reshape(lh[l], lw[l])
        s = np.float32(1.0) / sc[l]
        cx = int(np.rint(np.float32(k["x"]) * s)); cy = int(np.rint(np.float32(k["y"]) * s))
        ang = np.float32(k["angle"]) * np.float32(np.pi / np.float32(180.0))
        a = np.float32(np.cos(np.float64(ang))); b = np.float32(np.sin(np.float64(ang)))
        px = pat[:, 0].astype(np.float32); py = pat[:, 1].astype(np.float32)
        x = px * a - py * b
        y = px * b + py * a
        ix = np.rint(x).astype(int); iy = np.rint(y).astype(int)
        vals = blur[cy + iy, cx + ix].astype(int)
        bits = (vals[0::2] < vals[1::2]).astype(np.uint8)
        ref = np.packbits(bits.reshape(32, 8), axis=1, bitorder="little").ravel()
        assert np.array_equal(ref, desc[i]), i


def test_keypoint_fields(oracle, kitti02):
    kps, desc = oracle.orb_detect_compute(kitti02[0])
    assert 2900 <= len(kps) <= 3100 and desc.shape == (len(kps), 32)
    _, _, sc, _ = oracle.level_geometry(1241, 376)
    assert np.allclose(kps["size"], 31 * sc[kps["octave"]])
    assert (np.diff(kps["octave"]) >= 0).all()  # levels concatenated 0..7
    assert ((kps["angle"] >= 0) & (kps["angle"] < 360.0001)).all()
    # level coordinates are integers at least 31 px inside the level
    lw, lh, _, _ = oracle.level_geometry(1241, 376)
    lx = np.rint(kps["x"] / sc[kps["octave"]]); ly = np.rint(kps["y"] / sc[kps["octave"]])
    assert (lx >= 31).all() and (lx < lw[kps["octave"]] - 31).all()
    assert (ly >= 31).all() and (ly < lh[kps["octave"]] - 31).all()
