"""ctypes binding of libvo_mi355.so (include/vo_mi355.h).

The library is the product; there is no CPU fallback.  If the shared object is missing
or no GPU is present, the calls fail loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("VO_MI355_LIB") or os.path.join(_HERE, "libvo_mi355.so")   # the env override is for A/B builds (tools/exp_build.sh)

VO_OK, VO_ERR_INVALID, VO_ERR_HIP, VO_ERR_CAPACITY, VO_ERR_TOO_FEW, VO_ERR_NO_MODEL = 0, -1, -2, -3, -4, -5

VO_DBG_PYR_LEVEL, VO_DBG_BLUR_LEVEL, VO_DBG_FAST_LIST, VO_DBG_SEL1_LIST, VO_DBG_SEL1_HARRIS = 1, 2, 3, 4, 5
VO_DBG_KEYPOINTS, VO_DBG_DESCRIPTORS, VO_DBG_MATCHES, VO_DBG_E_MASK, VO_DBG_VOTES = 6, 7, 8, 9, 10
VO_N_KERNELS = 14
KERNEL_NAMES = ["k_pyramid", "k_fast_tiles", "k_fast_scan", "k_fast_emit", "k_select#1", "k_harris", "k_select#2", "k_orient",
                "k_blur", "k_rbrief", "k_match", "k_ratio", "ransac(stage)", "recover_pose(stage)"]


class OrbParams(C.Structure):
    _fields_ = [
        ("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32), ("edge_threshold", C.c_int32),
        ("patch_size", C.c_int32), ("fast_threshold", C.c_int32), ("blur_mode", C.c_int32), ("fma_atan", C.c_int32),
        ("fma_blur", C.c_int32), ("trig_mode", C.c_int32), ("keypoint_capacity", C.c_int32),
    ]


KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4")])
PAIR_DTYPE = np.dtype([("T", "<f8", (4, 4)), ("n_kp1", "<i4"), ("n_kp2", "<i4"), ("n_match", "<i4"), ("n_inlier", "<i4"),
                       ("n_good", "<i4"), ("status", "<i4")])

# every symbol include/vo_mi355.h declares
EXPORTS = [
    "vo_default_orb_params", "vo_ctx_create", "vo_ctx_destroy", "vo_last_error", "vo_orb_detect_compute", "vo_match_knn2_ratio",
    "vo_match_knn2", "vo_find_essential_ransac", "vo_recover_pose", "vo_triangulate", "vo_pairs_run_device", "vo_pairs_run", "vo_sequence_run_device", "vo_sequence_run", "vo_ba_plan_create", "vo_ba_plan_destroy", "vo_ba_pass_device", "vo_ba_pass", "vo_ba_set_kernel_timing", "vo_ba_get_kernel_timing",
    "vo_match_knn2_ratio_noalloc", "vo_get_matches", "vo_get_pose", "vo_pairs_triangulate", "vo_pairs_submit_device", "vo_sequence_submit_device", "vo_pairs_wait", "vo_debug_read", "vo_debug_minimal_solver", "vo_set_debug_stop", "vo_set_kernel_timing", "vo_get_kernel_timing",
]

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise RuntimeError(
            "libvo_mi355.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'`; "
            "there is no CPU fallback." % SO_PATH)
    # PyTorch-ROCm wheels bundle their own HIP runtime (torch/lib/libamdhip64.so).  The dynamic linker shares one copy
    # by SONAME only if torch's is loaded first; loading this library (linked against /opt/rocm) first and torch
    # afterwards leaves torch with "No HIP GPUs are available".  So where torch exists, import it before the library.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(SO_PATH)
    vp, i, d = C.c_void_p, C.c_int, C.c_double
    lib.vo_ctx_create.argtypes = [i, i, i, i, i, vp, C.POINTER(vp)]
    lib.vo_ctx_destroy.argtypes = [vp]
    lib.vo_ctx_destroy.restype = None
    lib.vo_last_error.argtypes = [vp]
    lib.vo_last_error.restype = C.c_char_p
    lib.vo_orb_detect_compute.argtypes = [vp, vp, i, i, i, vp, vp, i, vp]
    lib.vo_match_knn2_ratio.argtypes = [vp, vp, i, vp, i, d, vp, vp, vp, vp, i, vp]
    lib.vo_match_knn2.argtypes = [vp, vp, i, vp, i, vp, vp, vp, vp]
    lib.vo_match_knn2_ratio_noalloc.argtypes = [vp, vp, i, vp, i, d, vp, i, vp, vp]
    lib.vo_find_essential_ransac.argtypes = [vp, vp, vp, i, vp, d, d, i, vp, vp, vp]
    lib.vo_recover_pose.argtypes = [vp, vp, vp, vp, i, vp, d, vp, vp, vp]
    lib.vo_triangulate.argtypes = [vp, vp, vp, vp, vp, i, vp]
    lib.vo_get_matches.argtypes = [vp, vp, vp, i, i, i, d, vp, vp, i, vp]
    lib.vo_pairs_triangulate.argtypes = [vp, i, vp, vp, vp, vp, vp, C.c_long, vp]
    lib.vo_pairs_run_device.argtypes = [vp, vp, vp, i, i, i, i, vp, d, vp, vp]
    lib.vo_pairs_submit_device.argtypes = [vp, vp, vp, i, i, i, i, vp, d, vp, vp]
    lib.vo_pairs_wait.argtypes = [vp]
    lib.vo_sequence_submit_device.argtypes = [vp, vp, i, i, i, i, vp, d, vp, vp]
    lib.vo_pairs_run.argtypes = [vp, vp, vp, i, i, i, i, vp, d, vp]
    lib.vo_sequence_run_device.argtypes = [vp, vp, i, i, i, i, vp, d, vp, vp]
    lib.vo_sequence_run.argtypes = [vp, vp, i, i, i, i, vp, d, vp]
    lib.vo_ba_plan_create.argtypes = [i, i, i, i, vp, vp, C.POINTER(vp)]
    lib.vo_ba_plan_destroy.argtypes = [vp]
    lib.vo_ba_plan_destroy.restype = None
    lib.vo_ba_pass_device.argtypes = [vp] * 13
    lib.vo_ba_pass.argtypes = [vp] * 12
    lib.vo_ba_set_kernel_timing.argtypes = [vp, i]
    lib.vo_ba_get_kernel_timing.argtypes = [vp, vp, vp]
    lib.vo_debug_read.argtypes = [vp, i, i, i, vp, C.c_long]
    lib.vo_debug_read.restype = C.c_long
    lib.vo_debug_minimal_solver.argtypes = [vp, i, vp, i, vp]
    lib.vo_get_pose.argtypes = [vp, vp, vp, i, vp, vp, vp, vp]
    lib.vo_set_debug_stop.argtypes = [vp, i]
    lib.vo_set_kernel_timing.argtypes = [vp, i]
    lib.vo_get_kernel_timing.argtypes = [vp, vp, vp]
    _lib = lib
    return lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class VoError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libvo_mi355 error %d: %s" % (code, msg))
        self.code = code


class Context:
    """One device context (one HIP stream, one workspace).  Not thread-safe."""

    def __init__(self, max_w=1241, max_h=376, chunk_pairs=1, max_pairs=1, device=0, **orb):
        self.lib = load()
        self.params = OrbParams()
        self.lib.vo_default_orb_params(C.byref(self.params))
        for k, v in orb.items():
            setattr(self.params, k, v)
        h = C.c_void_p()
        rc = self.lib.vo_ctx_create(device, max_w, max_h, chunk_pairs, max_pairs, C.byref(self.params), C.byref(h))
        self.h = h
        if rc != VO_OK:
            msg = self.lib.vo_last_error(h).decode() if h else "allocation failed"
            if h:
                self.lib.vo_ctx_destroy(h)
            self.h = None
            raise VoError(rc, msg)
        kc = self.params.keypoint_capacity
        self.kp_cap = max(kc, self.params.nfeatures) if kc > 0 else self.params.nfeatures + 1024     # mirrors lane_init (vo_api.hip)
        self.device = device
        self.chunk_pairs, self.max_pairs = chunk_pairs, max(max_pairs, chunk_pairs)
        self._inflight = []

    def close(self):
        if getattr(self, "h", None):
            self.lib.vo_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != VO_OK:
            raise VoError(rc, self.lib.vo_last_error(self.h).decode())

    # ---- cv2 call replacements (host arrays)
    def orb_detect_compute(self, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        if img.ndim != 2:
            raise ValueError("expected a single-channel 8-bit image")
        h, w = img.shape
        kps = np.zeros(self.kp_cap, KP_DTYPE)
        desc = np.zeros((self.kp_cap, 32), np.uint8)
        n = C.c_int(0)
        self._check(self.lib.vo_orb_detect_compute(self.h, _ptr(img), w, h, img.strides[0], _ptr(kps), _ptr(desc), self.kp_cap, C.byref(n)))
        return kps[: n.value].copy(), desc[: n.value].copy()

    def match_knn2(self, q, t):
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        out = [np.zeros(max(len(q), 1), np.int32) for _ in range(4)]
        self._check(self.lib.vo_match_knn2(self.h, _ptr(q), len(q), _ptr(t), len(t), *[_ptr(o) for o in out]))
        return tuple(o[: len(q)] for o in out)

    def match_knn2_ratio(self, q, t, ratio=0.45):
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        cap = max(len(q), 1)
        out = [np.zeros(cap, np.int32) for _ in range(4)]
        n = C.c_int(0)
        self._check(self.lib.vo_match_knn2_ratio(self.h, _ptr(q), len(q), _ptr(t), len(t), float(ratio), *[_ptr(o) for o in out], cap, C.byref(n)))
        return tuple(o[: n.value].copy() for o in out)  # queryIdx, trainIdx, d1, d2

    def match_knn2_ratio_noalloc(self, q_ptr, nq, t_ptr, nt, out_ptr, cap, n_ptr, ratio=0.45, stream=None):
        """Device pointers in and out; enqueues on `stream` (None = default stream) and returns without synchronising.
        Allocates nothing (include/vo_mi355.h: vo_match_knn2_ratio_noalloc)."""
        self._check(self.lib.vo_match_knn2_ratio_noalloc(self.h, C.c_void_p(q_ptr), nq, C.c_void_p(t_ptr), nt, float(ratio), C.c_void_p(out_ptr), cap,
                                                         C.c_void_p(n_ptr), C.c_void_p(stream) if stream else None))

    def find_essential_ransac(self, q1, q2, K, prob=0.999, thr=1.0, max_iters=1000):
        q1 = np.ascontiguousarray(q1, np.float32).reshape(-1, 2)
        q2 = np.ascontiguousarray(q2, np.float32).reshape(-1, 2)
        K = np.ascontiguousarray(K, np.float64)
        n = len(q1)
        E = np.zeros((3, 3)); mask = np.zeros(max(n, 1), np.uint8); ninl = C.c_int(0)
        self._check(self.lib.vo_find_essential_ransac(self.h, _ptr(q1), _ptr(q2), n, _ptr(K), prob, thr, max_iters, _ptr(E), _ptr(mask), C.byref(ninl)))
        return E, mask[:n].copy(), ninl.value

    def get_pose(self, q1, q2, K):
        """findEssentialMat + recoverPose + form_transf in one call (libs/vo_utils.py:136-145) -> (T 4x4, n_inliers, n_good)"""
        q1 = np.ascontiguousarray(q1, np.float32).reshape(-1, 2)
        q2 = np.ascontiguousarray(q2, np.float32).reshape(-1, 2)
        K = np.ascontiguousarray(K, np.float64)
        T = np.zeros((4, 4)); ni = C.c_int(0); ng = C.c_int(0)
        self._check(self.lib.vo_get_pose(self.h, _ptr(q1), _ptr(q2), len(q1), _ptr(K), _ptr(T), C.byref(ni), C.byref(ng)))
        return T, ni.value, ng.value

    def recover_pose(self, E, q1, q2, K, dist_thresh=50.0):
        q1 = np.ascontiguousarray(q1, np.float32).reshape(-1, 2)
        q2 = np.ascontiguousarray(q2, np.float32).reshape(-1, 2)
        E = np.ascontiguousarray(E, np.float64); K = np.ascontiguousarray(K, np.float64)
        R = np.zeros((3, 3)); t = np.zeros(3); ng = C.c_int(0)
        self._check(self.lib.vo_recover_pose(self.h, _ptr(E), _ptr(q1), _ptr(q2), len(q1), _ptr(K), dist_thresh, _ptr(R), _ptr(t), C.byref(ng)))
        return R, t, ng.value

    def triangulate(self, P1, P2, x1, x2):
        P1 = np.ascontiguousarray(P1, np.float64); P2 = np.ascontiguousarray(P2, np.float64)
        x1 = np.ascontiguousarray(x1, np.float32).reshape(-1, 2); x2 = np.ascontiguousarray(x2, np.float32).reshape(-1, 2)
        n = len(x1)
        X4 = np.zeros((4, n), np.float32)
        self._check(self.lib.vo_triangulate(self.h, _ptr(P1), _ptr(P2), _ptr(x1), _ptr(x2), n, _ptr(X4)))
        return X4

    def get_matches(self, img_prev, img_cur, ratio=0.45):
        """get_matches of libs/vo_utils.py:83-110 in one call: (q1, q2) float32 (N,2)."""
        a = np.ascontiguousarray(img_prev, np.uint8); b = np.ascontiguousarray(img_cur, np.uint8)
        if a.ndim != 2 or a.shape != b.shape:
            raise ValueError("expected two single-channel 8-bit images of one size")
        h, w = a.shape
        q1 = np.zeros((self.kp_cap, 2), np.float32); q2 = np.zeros((self.kp_cap, 2), np.float32)
        n = C.c_int(0)
        self._check(self.lib.vo_get_matches(self.h, _ptr(a), _ptr(b), w, h, w, float(ratio), _ptr(q1), _ptr(q2), self.kp_cap, C.byref(n)))
        return q1[: n.value].copy(), q2[: n.value].copy()

    def pairs_triangulate(self, P1, P2, want_points2d=True):
        """All ratio-test matches of the first len(P1) pairs of the last pairs_run* / sequence_run* wave, triangulated in one
        launch with per-pair projection matrices P1[p], P2[p] (3x4 float64).  Returns (X float32 (N,3), q1 (N,2), q2 (N,2),
        offsets int64 (P+1)); q1, q2 are None unless want_points2d."""
        P1 = np.ascontiguousarray(P1, np.float64).reshape(-1, 12); P2 = np.ascontiguousarray(P2, np.float64).reshape(-1, 12)
        P = len(P1)
        if len(P2) != P:
            raise ValueError("P1 and P2 must hold one matrix per pair")
        cap = P * self.kp_cap
        X = np.zeros((cap, 3), np.float32)
        q1 = np.zeros((cap, 2), np.float32) if want_points2d else None
        q2 = np.zeros((cap, 2), np.float32) if want_points2d else None
        off = np.zeros(P + 1, np.int64)
        self._check(self.lib.vo_pairs_triangulate(self.h, P, _ptr(P1), _ptr(P2), _ptr(X), _ptr(q1) if want_points2d else None,
                                                  _ptr(q2) if want_points2d else None, cap, _ptr(off)))
        n = int(off[-1])
        return X[:n].copy(), (q1[:n].copy() if want_points2d else None), (q2[:n].copy() if want_points2d else None), off

    # ---- batched pairs
    def pairs_run(self, imgsA, imgsB, K, ratio=0.45):
        imgsA = np.ascontiguousarray(imgsA, np.uint8); imgsB = np.ascontiguousarray(imgsB, np.uint8)
        B, h, w = imgsA.shape
        K = np.ascontiguousarray(K, np.float64)
        out = np.zeros(B, PAIR_DTYPE)
        self._check(self.lib.vo_pairs_run(self.h, _ptr(imgsA), _ptr(imgsB), B, w, h, w, _ptr(K), float(ratio), _ptr(out)))
        return out

    def pairs_run_device(self, ptrA, ptrB, B, w, h, stride, K, ratio=0.45, stream=None):
        """ptrA/ptrB: device addresses (e.g. torch.Tensor.data_ptr()) of [B][h][stride] uint8."""
        K = np.ascontiguousarray(K, np.float64)
        out = np.zeros(B, PAIR_DTYPE)
        self._check(self.lib.vo_pairs_run_device(self.h, C.c_void_p(ptrA), C.c_void_p(ptrB), B, w, h, stride, _ptr(K), float(ratio), _ptr(out),
                                                 C.c_void_p(stream) if stream else None))
        return out

    def pairs_submit_device(self, ptrA, ptrB, B, w, h, stride, K, ratio=0.45, stream=None):
        """Asynchronous pairs_run_device (B <= max_pairs): returns once the work is enqueued; at most two in flight."""
        K = np.ascontiguousarray(K, np.float64)
        out = np.zeros(B, PAIR_DTYPE)
        self._check(self.lib.vo_pairs_submit_device(self.h, C.c_void_p(ptrA), C.c_void_p(ptrB), B, w, h, stride, _ptr(K), float(ratio), _ptr(out),
                                                    C.c_void_p(stream) if stream else None))
        self._inflight.append(out)

    def sequence_submit_device(self, ptr, F, w, h, stride, K, ratio=0.45, stream=None):
        """Asynchronous sequence_run_device (F - 1 <= max_pairs); pairs_wait() returns its records."""
        K = np.ascontiguousarray(K, np.float64)
        out = np.zeros(max(F - 1, 0), PAIR_DTYPE)
        self._check(self.lib.vo_sequence_submit_device(self.h, C.c_void_p(ptr), F, w, h, stride, _ptr(K), float(ratio), _ptr(out),
                                                       C.c_void_p(stream) if stream else None))
        self._inflight.append(out)

    def pairs_wait(self):
        """Records of the oldest submission in flight."""
        if not self._inflight:
            self._check(self.lib.vo_pairs_wait(self.h))       # raises: nothing in flight
        # vo_pairs_wait retires the oldest submission whether or not it succeeds: mirror that, so the queue here and the
        # library's n_waited never get out of step
        out = self._inflight.pop(0)
        self._check(self.lib.vo_pairs_wait(self.h))
        return out

    def sequence_run(self, imgs, K, ratio=0.45):
        """imgs: uint8 [F,h,w] consecutive frames (host).  Returns PAIR_DTYPE [F-1]: pair i = (frame i, frame i+1)."""
        imgs = np.ascontiguousarray(imgs, np.uint8)
        F, h, w = imgs.shape
        K = np.ascontiguousarray(K, np.float64)
        out = np.zeros(max(F - 1, 0), PAIR_DTYPE)
        self._check(self.lib.vo_sequence_run(self.h, _ptr(imgs), F, w, h, w, _ptr(K), float(ratio), _ptr(out)))
        return out

    def sequence_run_device(self, ptr, F, w, h, stride, K, ratio=0.45, stream=None):
        """ptr: device address of [F][h][stride] uint8 consecutive frames."""
        K = np.ascontiguousarray(K, np.float64)
        out = np.zeros(max(F - 1, 0), PAIR_DTYPE)
        self._check(self.lib.vo_sequence_run_device(self.h, C.c_void_p(ptr), F, w, h, stride, _ptr(K), float(ratio), _ptr(out),
                                                    C.c_void_p(stream) if stream else None))
        return out

    # ---- diagnostics
    def debug_solve_poly10(self, coeffs, wide=False):
        """the solver's cv::solvePoly on [n, 11] coefficient vectors (ascending powers) -> complex128 [n, 10]; wide: the
        16-lanes-per-case form that small batches use"""
        c = np.ascontiguousarray(coeffs, np.float64).reshape(-1, 11)
        out = np.zeros((len(c), 20))
        self._check(self.lib.vo_debug_minimal_solver(self.h, 2 if wide else 0, _ptr(c), len(c), _ptr(out)))
        return out.view(np.complex128)

    def debug_five_point(self, x1, x2, wide=False):
        """the five-point minimal solver on [n, 5, 2] normalised correspondences -> (counts int [n], E float64 [n, 10, 3, 3])"""
        x1 = np.ascontiguousarray(x1, np.float64).reshape(-1, 10); x2 = np.ascontiguousarray(x2, np.float64).reshape(-1, 10)
        inp = np.ascontiguousarray(np.concatenate([x1, x2], axis=1))
        out = np.zeros((len(inp), 91))
        self._check(self.lib.vo_debug_minimal_solver(self.h, 3 if wide else 1, _ptr(inp), len(inp), _ptr(out)))
        return out[:, 0].astype(np.int64), out[:, 1:].reshape(-1, 10, 3, 3).copy()

    def set_debug_stop(self, stage):
        self._check(self.lib.vo_set_debug_stop(self.h, stage))

    def set_kernel_timing(self, enable):
        self._check(self.lib.vo_set_kernel_timing(self.h, int(enable)))

    def kernel_timing(self):
        """{kernel name: (total ms, launches)} since timing was enabled."""
        ms = np.zeros(VO_N_KERNELS, np.float32); n = np.zeros(VO_N_KERNELS, np.int32)
        self._check(self.lib.vo_get_kernel_timing(self.h, _ptr(ms), _ptr(n)))
        return {k: (float(a), int(b)) for k, a, b in zip(KERNEL_NAMES, ms, n)}

    def debug_read(self, what, index, level=0, dtype=np.uint8):
        n = self.lib.vo_debug_read(self.h, what, index, level, None, 0)
        if n < 0:
            raise VoError(int(n), "vo_debug_read")
        buf = np.zeros(max(int(n), 1), np.uint8)
        if n:
            self.lib.vo_debug_read(self.h, what, index, level, _ptr(buf), int(n))
        return buf[: int(n)].view(dtype)
