"""ctypes wrapper over oracle/libvo_oracle.so.

TEST INFRASTRUCTURE ONLY -- the CPU parity oracle (PARITY UNPINNED, see vo_oracle.h).
Imported only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Never imported by the product package (visual_slam_amd/).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libvo_oracle.so")


class OrbParams(C.Structure):
    _fields_ = [
        ("nfeatures", C.c_int32),
        ("scale_factor", C.c_float),
        ("nlevels", C.c_int32),
        ("edge_threshold", C.c_int32),
        ("patch_size", C.c_int32),
        ("fast_threshold", C.c_int32),
        ("blur_mode", C.c_int32),
        ("fma_atan", C.c_int32),
        ("fma_blur", C.c_int32),
        ("trig_mode", C.c_int32),
    ]


class OrbDebug(C.Structure):
    _fields_ = [
        ("pyr", C.c_void_p),
        ("blurred", C.c_void_p),
        ("fast_count", C.c_void_p),
        ("fast_xys", C.c_void_p),
        ("fast_cap", C.c_int32),
        ("sel1_count", C.c_void_p),
        ("sel1_xys", C.c_void_p),
        ("sel1_harris", C.c_void_p),
        ("sel1_cap", C.c_int32),
        ("sel2_count", C.c_void_p),
    ]


KP_DTYPE = np.dtype(
    [("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4")]
)


def build(force=False):
    """Compile the oracle with the committed Makefile (gcc only, a few seconds)."""
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "vo_oracle.cpp")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_orb_detect_compute.restype = C.c_int
        _lib.orc_match_knn2.restype = C.c_int
        _lib.orc_find_essential_ransac.restype = C.c_int
        _lib.orc_find_essential_ransac.argtypes = [
            C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_double, C.c_int,
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
        ]
        _lib.orc_recover_pose.argtypes = [
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p,
        ]
        _lib.orc_recover_pose_votes.argtypes = [
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
        ]
        _lib.orc_ratio_filter.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
        _lib.orc_pair_run.argtypes = [
            C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p,
        ]
    return _lib


def default_params(**kw):
    p = OrbParams()
    lib().orc_default_orb_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def level_geometry(w, h, p=None):
    p = p or default_params()
    n = p.nlevels
    lw = np.zeros(n, np.int32); lh = np.zeros(n, np.int32); sc = np.zeros(n, np.float32); q = np.zeros(n, np.int32)
    lib().orc_level_geometry(C.c_int(w), C.c_int(h), C.byref(p), _ptr(lw), _ptr(lh), _ptr(sc), _ptr(q))
    return lw, lh, sc, q


def orb_detect_compute(img, p=None, debug=False):
    """Returns (kps structured array, desc uint8 [n,32]) and, with debug=True, a dict of stage dumps."""
    p = p or default_params()
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    cap = p.nfeatures * 2 + 8192
    kps = np.zeros(cap, KP_DTYPE)
    desc = np.zeros((cap, 32), np.uint8)
    n = C.c_int(0)
    dbg = None
    dump = None
    if debug:
        lw, lh, _, _ = level_geometry(w, h, p)
        total = int((lw.astype(np.int64) * lh).sum())
        fast_cap = total // 4 + 16
        dump = dict(
            pyr=np.zeros(total, np.uint8), blurred=np.zeros(total, np.uint8),
            fast_count=np.zeros(p.nlevels, np.int32), fast_xys=np.zeros((fast_cap, 3), np.int32),
            sel1_count=np.zeros(p.nlevels, np.int32), sel1_xys=np.zeros((cap * 2, 3), np.int32),
            sel1_harris=np.zeros(cap * 2, np.float32), sel2_count=np.zeros(p.nlevels, np.int32),
            lw=lw, lh=lh,
        )
        dbg = OrbDebug(
            _ptr(dump["pyr"]), _ptr(dump["blurred"]), _ptr(dump["fast_count"]), _ptr(dump["fast_xys"]), fast_cap,
            _ptr(dump["sel1_count"]), _ptr(dump["sel1_xys"]), _ptr(dump["sel1_harris"]), cap * 2, _ptr(dump["sel2_count"]),
        )
    rc = lib().orc_orb_detect_compute(
        _ptr(img), C.c_int(w), C.c_int(h), C.c_int(img.strides[0]), C.byref(p),
        C.byref(dbg) if dbg is not None else None, _ptr(kps), _ptr(desc), C.c_int(cap), C.byref(n),
    )
    if rc != 0:
        raise RuntimeError("orc_orb_detect_compute failed: %d" % rc)
    if debug:
        return kps[: n.value].copy(), desc[: n.value].copy(), dump
    return kps[: n.value].copy(), desc[: n.value].copy()


def match_knn2(q, t):
    q = np.ascontiguousarray(q, np.uint8); t = np.ascontiguousarray(t, np.uint8)
    nq, nt = len(q), len(t)
    out = [np.zeros(nq, np.int32) for _ in range(4)]
    rc = lib().orc_match_knn2(_ptr(q), C.c_int(nq), _ptr(t), C.c_int(nt), *[_ptr(o) for o in out])
    if rc != 0:
        raise ValueError("knnMatch needs at least 2 train descriptors")
    return tuple(out)  # idx1, d1, idx2, d2


def ratio_filter(d1, d2, ratio=0.45):
    d1 = np.ascontiguousarray(d1, np.int32); d2 = np.ascontiguousarray(d2, np.int32)
    keep = np.zeros(len(d1), np.int32)
    n = C.c_int(0)
    lib().orc_ratio_filter(_ptr(d1), _ptr(d2), len(d1), float(ratio), _ptr(keep), C.byref(n))
    return keep[: n.value].copy()


def find_essential_ransac(q1, q2, K, prob=0.999, thr=1.0, max_iters=1000):
    q1 = np.ascontiguousarray(q1, np.float32).reshape(-1, 2); q2 = np.ascontiguousarray(q2, np.float32).reshape(-1, 2)
    K = np.ascontiguousarray(K, np.float64)
    n = len(q1)
    E = np.zeros((3, 3)); mask = np.zeros(max(n, 1), np.uint8)
    ninl = C.c_int(0); iters = C.c_int(0)
    rc = lib().orc_find_essential_ransac(_ptr(q1), _ptr(q2), n, _ptr(K), prob, thr, max_iters, _ptr(E), _ptr(mask), C.byref(ninl), C.byref(iters))
    if rc != 0:
        raise RuntimeError("findEssentialMat: no model")
    return E, mask[:n].copy(), ninl.value, iters.value


def set_pose_variant(hypot_mode=0, ainv_mode=0):
    """Rounding variants of the pose side (process-wide, see vo_oracle.cpp): hypot_mode 1 = std::hypot in the Jacobi
    rotations; ainv_mode 1 = LU inverse + matrix product for the 10 x 10 elimination.  (0, 0) = defaults = the HIP solver."""
    lib().orc_set_pose_variant(int(hypot_mode), int(ainv_mode))


class pose_variant:
    """with oracle.pose_variant(hypot_mode=1): ...   (restores the defaults on exit)"""

    def __init__(self, hypot_mode=0, ainv_mode=0):
        self.v = (hypot_mode, ainv_mode)

    def __enter__(self):
        set_pose_variant(*self.v)
        return self

    def __exit__(self, *exc):
        set_pose_variant(0, 0)


def five_point(x1, x2):
    x1 = np.ascontiguousarray(x1, np.float64); x2 = np.ascontiguousarray(x2, np.float64)
    Eb = np.zeros((10, 3, 3))
    lib().orc_five_point.restype = C.c_int
    n = lib().orc_five_point(_ptr(x1), _ptr(x2), _ptr(Eb))
    return Eb[:n].copy()


def solve_poly10(coeffs):
    """cv::solvePoly(coeffs, roots, 300) for one degree-10 coefficient vector (ascending powers) -> complex128 [10]"""
    c = np.ascontiguousarray(coeffs, np.float64).reshape(11)
    out = np.zeros(20)
    lib().orc_solve_poly10(_ptr(c), _ptr(out))
    return out.view(np.complex128).copy()


def recover_pose(E, q1, q2, K, dist_thresh=50.0):
    q1 = np.ascontiguousarray(q1, np.float32).reshape(-1, 2); q2 = np.ascontiguousarray(q2, np.float32).reshape(-1, 2)
    E = np.ascontiguousarray(E, np.float64); K = np.ascontiguousarray(K, np.float64)
    R = np.zeros((3, 3)); t = np.zeros(3); ng = C.c_int(0)
    lib().orc_recover_pose(_ptr(E), _ptr(q1), _ptr(q2), len(q1), _ptr(K), dist_thresh, _ptr(R), _ptr(t), C.byref(ng))
    return R, t, ng.value


def recover_pose_votes(E, q1, q2, K, dist_thresh=50.0):
    """recover_pose plus the cheirality counts of the four candidates [R1|t], [R2|t], [R1|-t], [R2|-t] (int32 [4])."""
    q1 = np.ascontiguousarray(q1, np.float32).reshape(-1, 2); q2 = np.ascontiguousarray(q2, np.float32).reshape(-1, 2)
    E = np.ascontiguousarray(E, np.float64); K = np.ascontiguousarray(K, np.float64)
    R = np.zeros((3, 3)); t = np.zeros(3); ng = C.c_int(0); votes = np.zeros(4, np.int32)
    lib().orc_recover_pose_votes(_ptr(E), _ptr(q1), _ptr(q2), len(q1), _ptr(K), dist_thresh, _ptr(R), _ptr(t), C.byref(ng), _ptr(votes))
    return R, t, ng.value, votes


def triangulate_f32(P1, P2, x1, x2):
    P1 = np.ascontiguousarray(P1, np.float64); P2 = np.ascontiguousarray(P2, np.float64)
    x1 = np.ascontiguousarray(x1, np.float32).reshape(-1, 2); x2 = np.ascontiguousarray(x2, np.float32).reshape(-1, 2)
    n = len(x1)
    X4 = np.zeros((4, n), np.float32)
    lib().orc_triangulate_f32(_ptr(P1), _ptr(P2), _ptr(x1), _ptr(x2), C.c_int(n), _ptr(X4))
    return X4


def pair_run(imgA, imgB, K, ratio=0.45, p=None):
    p = p or default_params()
    imgA = np.ascontiguousarray(imgA, np.uint8); imgB = np.ascontiguousarray(imgB, np.uint8)
    h, w = imgA.shape
    K = np.ascontiguousarray(K, np.float64)
    T = np.zeros((4, 4)); counts = np.zeros(4, np.int32)
    rc = lib().orc_pair_run(_ptr(imgA), _ptr(imgB), w, h, imgA.strides[0], C.byref(p), _ptr(K), float(ratio), _ptr(T), _ptr(counts))
    return rc, T, counts


def retain_best(keys, n_points):
    keys = np.ascontiguousarray(keys, np.float32)
    out = np.zeros(len(keys), np.int32); n = C.c_int(0)
    lib().orc_retain_best(_ptr(keys), C.c_int(len(keys)), C.c_int(n_points), _ptr(out), C.byref(n))
    return out[: n.value].copy()
