"""Seeded synthetic frame pairs of BASELINE.json configs 2-4.

frame A = 4-octave value noise (cell sizes 4, 8, 16, 32 px; amplitudes 1, 1/2, 1/4, 1/8) plus
200 random axis-aligned rectangles (side 8-64 px, uniform gray), stretched to [0,255];
frame B = A inverse-warped (bilinear) by a KITTI-like two-view geometry: ground plane 1.65 m
below the camera for rows below the principal point, a fronto-parallel wall at 30 m above it,
camera motion 1 m forward with 0.5 degree yaw.  numpy.random.default_rng(seed) only.
"""
import numpy as np

K_KITTI02 = np.array([[718.856, 0.0, 607.1928], [0.0, 718.856, 185.2157], [0.0, 0.0, 1.0]])


def _interp_matrix(n_out, cell, n_grid):
    """Bilinear interpolation weights from a grid with spacing `cell` to n_out samples."""
    pos = (np.arange(n_out) + 0.5) / cell
    i0 = np.floor(pos).astype(np.int64)
    f = pos - i0
    W = np.zeros((n_out, n_grid), np.float32)
    W[np.arange(n_out), i0] = 1 - f
    W[np.arange(n_out), i0 + 1] = f
    return W


def frame_a(rng, w, h):
    img = np.zeros((h, w), np.float32)
    for cell, amp in ((4, 1.0), (8, 0.5), (16, 0.25), (32, 0.125)):
        gh, gw = h // cell + 3, w // cell + 3
        grid = rng.random((gh, gw), dtype=np.float32)
        img += amp * (_interp_matrix(h, cell, gh) @ grid @ _interp_matrix(w, cell, gw).T)
    img /= 1.875
    n_rect = 200
    xs = rng.integers(0, w, n_rect); ys = rng.integers(0, h, n_rect)
    ws = rng.integers(8, 65, n_rect); hs = rng.integers(8, 65, n_rect)
    gs = rng.random(n_rect, dtype=np.float32)
    for x, y, rw, rh, g in zip(xs, ys, ws, hs, gs):
        img[y:y + rh, x:x + rw] = g
    lo, hi = float(img.min()), float(img.max())
    return np.clip((img - lo) * (255.0 / (hi - lo)), 0, 255).astype(np.uint8)


def relative_pose(yaw_deg=0.5, forward=1.0):
    """x_B = R x_A + t for a camera that moves `forward` metres along +z and yaws."""
    a = np.deg2rad(yaw_deg)
    R = np.array([[np.cos(a), 0, -np.sin(a)], [0, 1, 0], [np.sin(a), 0, np.cos(a)]])
    C = np.array([0.0, 0.0, forward])
    return R, -R @ C


_WARP_CACHE = {}


def _warp_maps(w, h, K):
    """Seed-independent sampling positions of frame B in frame A (cached per geometry)."""
    key = (w, h, K.tobytes())
    if key not in _WARP_CACHE:
        R, t = relative_pose()
        u, v = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
        dx = (u - K[0, 2]) / K[0, 0]
        dy = (v - K[1, 2]) / K[1, 1]
        depth = np.where(dy > 1.65 / 30.0, 1.65 / np.maximum(dy, 1e-9), 30.0)   # ground plane below, wall above
        XB = np.stack([dx * depth, dy * depth, depth], -1)
        XA = (XB - t) @ R        # R^T (X_B - t)
        ua = K[0, 0] * XA[..., 0] / XA[..., 2] + K[0, 2]
        va = K[1, 1] * XA[..., 1] / XA[..., 2] + K[1, 2]
        ua = np.clip(ua, 0, w - 1.001); va = np.clip(va, 0, h - 1.001)
        x0 = np.floor(ua).astype(np.int64); y0 = np.floor(va).astype(np.int64)
        fx = (ua - x0).astype(np.float32); fy = (va - y0).astype(np.float32)
        i00 = (y0 * w + x0).ravel()
        _WARP_CACHE[key] = (i00, fx.ravel(), fy.ravel())
    return _WARP_CACHE[key]


def warp_b(img_a, K):
    h, w = img_a.shape
    i00, fx, fy = _warp_maps(w, h, np.ascontiguousarray(K, np.float64))
    a = img_a.astype(np.float32).ravel()
    top = a[i00] * (1 - fx) + a[i00 + 1] * fx
    bot = a[i00 + w] * (1 - fx) + a[i00 + w + 1] * fx
    return np.clip(np.rint(top * (1 - fy) + bot * fy), 0, 255).astype(np.uint8).reshape(h, w)


def config_K(w, h):
    """Intrinsics of the generator at frame size w x h: KITTI 02 scaled by w / 1241."""
    if w == 1241:
        return K_KITTI02
    K = K_KITTI02 * np.array([[w / 1241.0], [w / 1241.0], [1.0]])
    K[1, 2] = h / 2.0
    return K


def make_pair(seed, w=1241, h=376, K=None):
    """Returns (frame A, frame B, K, R, t) for one seed."""
    if K is None:
        K = config_K(w, h)
    rng = np.random.default_rng(seed)
    a = frame_a(rng, w, h)
    b = warp_b(a, K)
    R, t = relative_pose()
    return a, b, K, R, t


def _pair_ab(args):
    seed, w, h = args
    a, b, _, _, _ = make_pair(seed, w, h)
    return a, b


def make_batch(seeds, w=1241, h=376, workers=1):
    """uint8 arrays [B,h,w] x2 for the given seeds."""
    seeds = list(seeds)
    if workers > 1 and len(seeds) > 1:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(workers) as pool:
            out = pool.map(_pair_ab, [(s, w, h) for s in seeds], chunksize=max(1, len(seeds) // (workers * 4)))
    else:
        out = [_pair_ab((s, w, h)) for s in seeds]
    return np.stack([o[0] for o in out]), np.stack([o[1] for o in out])
