"""GPU parity of the detect stages against the CPU oracle, stage by stage, through the C ABI
(bit exact: pyramid, FAST lists, retainBest order, Harris, keypoints, blur, descriptors)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from visual_slam_amd import _lib
    c = _lib.Context(max_w=1241, max_h=376, chunk_pairs=1, max_pairs=1)
    yield c
    c.close()


def _levels(dump):
    lw, lh = dump["lw"], dump["lh"]
    ofs = np.concatenate([[0], np.cumsum(lw.astype(np.int64) * lh)])
    return lw, lh, ofs


@pytest.mark.parametrize("which", ["kitti02_0", "kitti02_1", "kitti09_0"])
def test_detect_stages_bit_exact(ctx, oracle, kitti02, kitti09, which):
    from visual_slam_amd import _lib as L
    img = {"kitti02_0": kitti02[0], "kitti02_1": kitti02[1], "kitti09_0": kitti09[0]}[which]
    kps_o, desc_o, dump = oracle.orb_detect_compute(img, debug=True)
    lw, lh, ofs = _levels(dump)

    # stage 1: stop after the FAST lists
    ctx.set_debug_stop(1)
    ctx.orb_detect_compute(img)
    fo = np.concatenate([[0], np.cumsum(dump["fast_count"])])
    for l in range(8):
        pyr = ctx.debug_read(L.VO_DBG_PYR_LEVEL, 0, l)
        assert np.array_equal(pyr, dump["pyr"][ofs[l]:ofs[l + 1]]), "pyramid level %d" % l
        fl = ctx.debug_read(L.VO_DBG_FAST_LIST, 0, l, np.int32).reshape(-1, 3)
        assert np.array_equal(fl, dump["fast_xys"][fo[l]:fo[l + 1]]), "FAST list level %d" % l

    # stage 2: retainBest #1 order + Harris
    ctx.set_debug_stop(2)
    ctx.orb_detect_compute(img)
    so = np.concatenate([[0], np.cumsum(dump["sel1_count"])])
    for l in range(8):
        sl = ctx.debug_read(L.VO_DBG_SEL1_LIST, 0, l, np.int32).reshape(-1, 3)
        assert np.array_equal(sl, dump["sel1_xys"][so[l]:so[l + 1]]), "retainBest#1 order level %d" % l
        hr = ctx.debug_read(L.VO_DBG_SEL1_HARRIS, 0, l, np.float32)
        assert np.array_equal(hr.view(np.uint32), dump["sel1_harris"][so[l]:so[l + 1]].view(np.uint32)), "Harris level %d" % l

    # full run
    ctx.set_debug_stop(0)
    kps, desc = ctx.orb_detect_compute(img)
    for l in range(8):
        bl = ctx.debug_read(L.VO_DBG_BLUR_LEVEL, 0, l)
        assert np.array_equal(bl, dump["blurred"][ofs[l]:ofs[l + 1]]), "blur level %d" % l
    assert len(kps) == len(kps_o)
    for f in ("x", "y", "size", "response", "octave", "angle"):
        assert np.array_equal(kps[f].view(np.uint32), kps_o[f].view(np.uint32)), f
    assert np.array_equal(desc, desc_o)


# every switchable variant of recalled OpenCV behaviour (SURVEY.md App. A "[L]" items), each on its own and combined:
# (blur_mode, fma_atan, fma_blur, trig_mode); the default (0, 1, 1, 0) is what every other test runs
VARIANTS = [(0, 0, 0, 0), (0, 0, 1, 0), (0, 1, 0, 0), (1, 0, 0, 0), (1, 1, 1, 0), (0, 1, 1, 1), (1, 0, 1, 1), (0, 0, 0, 1)]


@pytest.mark.parametrize("mode,fma_atan,fma_blur,trig", VARIANTS)
def test_blur_fma_and_trig_variants(oracle, kitti02, mode, fma_atan, fma_blur, trig):
    from visual_slam_amd import _lib
    v = dict(blur_mode=mode, fma_atan=fma_atan, fma_blur=fma_blur, trig_mode=trig)
    c = _lib.Context(max_w=1241, max_h=376, **v)
    kps, desc = c.orb_detect_compute(kitti02[2])
    kps_o, desc_o = oracle.orb_detect_compute(kitti02[2], oracle.default_params(**v))
    c.close()
    assert np.array_equal(kps.view(np.uint8), kps_o.view(np.uint8))
    assert np.array_equal(desc, desc_o)


def test_detect_other_sizes_and_params(oracle, kitti02):
    from visual_slam_amd import _lib
    rng = np.random.default_rng(5)
    for (h, w, nf, nl) in [(200, 320, 500, 8), (376, 1241, 3000, 4), (96, 128, 200, 3), (150, 150, 1000, 8),
                           (233, 317, 800, 8), (75, 1241, 300, 2), (376, 71, 100, 2), (301, 997, 2000, 6)]:   # odd sizes, thin frames
        img = kitti02[3][:h, :w].copy() if h <= 376 else None
        if rng.integers(0, 2):
            img = (img.astype(np.int32) + rng.integers(-20, 20, img.shape)).clip(0, 255).astype(np.uint8)
        c = _lib.Context(max_w=w, max_h=h, nfeatures=nf, nlevels=nl)
        kps, desc = c.orb_detect_compute(img)
        kps_o, desc_o = oracle.orb_detect_compute(img, oracle.default_params(nfeatures=nf, nlevels=nl))
        c.close()
        assert len(kps) == len(kps_o), (h, w, nf, nl)
        assert np.array_equal(kps.view(np.uint8), kps_o.view(np.uint8)), (h, w, nf, nl)
        assert np.array_equal(desc, desc_o), (h, w, nf, nl)


def test_pyramid_and_blur_levels_whole_on_many_widths(oracle):
    """Every pyramid level and every blurred level, WHOLE (the descriptor tests only see a blurred level 12 px inside its
    border), on frame widths of every residue mod 4 and around the 64-px tile boundaries: the lanes at a row's right edge
    (pulled-back 8-byte window of k_pyramid, reflected dwords of k_blur) and level widths down to 18 px."""
    from visual_slam_amd import _lib as L, synth
    sizes = [(64, 64), (65, 70), (66, 99), (67, 64), (127, 80), (128, 65), (129, 131), (191, 64), (257, 100), (258, 67), (259, 75), (260, 90),
             (319, 201), (383, 70), (449, 133), (511, 64), (513, 77), (640, 96), (1021, 70), (1279, 66)]
    c = L.Context(max_w=1280, max_h=256, chunk_pairs=1, max_pairs=1)
    try:
        for k, (w, h) in enumerate(sizes):
            img = np.ascontiguousarray(synth.frame_a(np.random.default_rng(300 + k), max(w, 256), max(h, 256))[:h, :w])
            _, _, dump = oracle.orb_detect_compute(img, debug=True)
            lw, lh, ofs = _levels(dump)
            c.orb_detect_compute(img)
            for l in range(8):
                assert np.array_equal(c.debug_read(L.VO_DBG_PYR_LEVEL, 0, l), dump["pyr"][ofs[l]:ofs[l + 1]]), ("pyramid", w, h, l, int(lw[l]))
                assert np.array_equal(c.debug_read(L.VO_DBG_BLUR_LEVEL, 0, l), dump["blurred"][ofs[l]:ofs[l + 1]]), ("blur", w, h, l, int(lw[l]))
    finally:
        c.close()


def test_flat_image_has_no_keypoints(ctx):
    kps, desc = ctx.orb_detect_compute(np.full((376, 1241), 128, np.uint8))
    assert len(kps) == 0 and desc.shape == (0, 32)


def test_row_stride_is_honoured(ctx, oracle, kitti02):
    big = np.zeros((376, 1300), np.uint8)
    big[:, :1241] = kitti02[0]
    view = big[:, :1241]
    from visual_slam_amd import _lib as L
    import ctypes as C
    kps = np.zeros(ctx.kp_cap, L.KP_DTYPE); desc = np.zeros((ctx.kp_cap, 32), np.uint8); n = C.c_int(0)
    rc = ctx.lib.vo_orb_detect_compute(ctx.h, view.ctypes.data_as(C.c_void_p), 1241, 376, 1300, kps.ctypes.data_as(C.c_void_p),
                                       desc.ctypes.data_as(C.c_void_p), ctx.kp_cap, C.byref(n))
    assert rc == 0
    ko, do = oracle.orb_detect_compute(kitti02[0])
    assert n.value == len(ko) and np.array_equal(desc[: n.value], do)
