"""Bundle-adjustment residual / Jacobian pass on the GPU (BASELINE.json configs[4]).

The reference runs scipy `least_squares` on `VisualOdometry.bundle_adjustment_residuals`
(/root/reference/libs/vo_utils.py:218-245, called at /root/reference/vo.py:118) with a
finite-difference dense Jacobian over 6 + 3N parameters.  Here one launch pair evaluates, for P
poses x N points x M observations, the reprojection residuals (D = 0) with ANALYTIC Jacobians
and reduces them to the normal-equation blocks a Schur-complement solver consumes
(`vo_ba_pass*`, include/vo_mi355.h).  For an observation set sharded over ranks the pose blocks
are summed with one RCCL all_reduce (`allreduce_pose_blocks`); point blocks stay rank-local.
"""
import ctypes as C

import numpy as np

from . import _lib


def synth_problem(n_poses=1000, n_points=200000, obs_per_point=5, seed=0, K=None, noise=0.5):
    """BASELINE.json configs[4] / SURVEY.md 8(d) config 5: poses on a smooth trajectory, points
    uniform in a corridor around it, each point observed by `obs_per_point` consecutive poses,
    pixel noise sigma = `noise`, numpy default_rng(seed).  Observations are sorted by point.
    Returns dict(poses [P,6] (rvec,tvec: world->camera), points [N,3], obs_pose [M] i32,
    obs_point [M] i32, uv [M,2] f64, K [3,3])."""
    rng = np.random.default_rng(seed)
    if K is None:
        K = np.array([[718.856, 0, 607.1928], [0, 718.856, 185.2157], [0, 0, 1.0]])
    P, N = int(n_poses), int(n_points)
    s = np.arange(P, dtype=np.float64)
    # camera centres: forward motion with a slow sinusoidal drift; yaw follows the path
    C_w = np.stack([3.0 * np.sin(s / 80.0), 0.2 * np.sin(s / 37.0), 1.0 * s], 1)
    yaw = np.arctan(3.0 / 80.0 * np.cos(s / 80.0))
    poses = np.zeros((P, 6))
    for j in range(P):
        a = -yaw[j]                                                    # world -> camera: rotation about +y by -yaw
        Rcw = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        poses[j, 1] = a                                                # rvec = (0, a, 0)
        poses[j, 3:] = -Rcw @ C_w[j]
    first = rng.integers(0, P - obs_per_point + 1, N)
    mid = first + obs_per_point // 2
    depth = rng.uniform(8.0, 40.0, N)
    lateral = rng.uniform(-6.0, 6.0, N)
    height = rng.uniform(-2.0, 2.0, N)
    points = C_w[mid] + np.stack([lateral, height, depth], 1)
    obs_point = np.repeat(np.arange(N, dtype=np.int32), obs_per_point)
    obs_pose = (first[:, None] + np.arange(obs_per_point)[None, :]).astype(np.int32).ravel()
    # exact projections + noise
    rv = poses[obs_pose, :3]
    a = rv[:, 1]
    X = points[obs_point]
    ca, sa = np.cos(a), np.sin(a)
    Xc = np.stack([ca * X[:, 0] + sa * X[:, 2], X[:, 1], -sa * X[:, 0] + ca * X[:, 2]], 1) + poses[obs_pose, 3:]
    uv = np.stack([K[0, 0] * Xc[:, 0] / Xc[:, 2] + K[0, 2], K[1, 1] * Xc[:, 1] / Xc[:, 2] + K[1, 2]], 1)
    uv += rng.normal(0, noise, uv.shape)
    return {"poses": poses, "points": points, "obs_pose": obs_pose, "obs_point": obs_point, "uv": uv, "K": K}


class BAPlan:
    """Observation structure of one BA problem (static over solver iterations): uploads the
    (pose, point) index of every observation and the by-pose / by-point groupings once."""

    def __init__(self, obs_pose, obs_point, n_poses, n_points, device=0):
        self.lib = _lib.load()
        self.obs_pose = np.ascontiguousarray(obs_pose, np.int32)
        self.obs_point = np.ascontiguousarray(obs_point, np.int32)
        self.M = len(self.obs_pose)
        self.P, self.N = int(n_poses), int(n_points)
        h = C.c_void_p()
        rc = self.lib.vo_ba_plan_create(device, self.P, self.N, self.M, _lib._ptr(self.obs_pose), _lib._ptr(self.obs_point), C.byref(h))
        if rc != 0:
            raise _lib.VoError(rc, "vo_ba_plan_create failed (bad indices, no GPU, or out of memory)")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.vo_ba_plan_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, poses, points, uv, K):
        """Host arrays in, host arrays out: dict(r [M,2], norm [M], Hpp [P,6,6], bp [P,6],
        Hll [N,3,3], bl [N,3], Hpl [M,6,3]) float64."""
        poses = np.ascontiguousarray(poses, np.float64).reshape(self.P, 6)
        points = np.ascontiguousarray(points, np.float64).reshape(self.N, 3)
        uv = np.ascontiguousarray(uv, np.float64).reshape(self.M, 2)
        K = np.ascontiguousarray(K, np.float64).reshape(3, 3)
        out = {"r": np.zeros((self.M, 2)), "norm": np.zeros(self.M), "Hpp": np.zeros((self.P, 6, 6)), "bp": np.zeros((self.P, 6)),
               "Hll": np.zeros((self.N, 3, 3)), "bl": np.zeros((self.N, 3)), "Hpl": np.zeros((self.M, 6, 3))}
        rc = self.lib.vo_ba_pass(self.h, _lib._ptr(poses), _lib._ptr(points), _lib._ptr(uv), _lib._ptr(K), _lib._ptr(out["r"]),
                                 _lib._ptr(out["norm"]), _lib._ptr(out["Hpp"]), _lib._ptr(out["bp"]), _lib._ptr(out["Hll"]),
                                 _lib._ptr(out["bl"]), _lib._ptr(out["Hpl"]))
        if rc != 0:
            raise _lib.VoError(rc, "vo_ba_pass failed")
        return out

    def set_kernel_timing(self, on):
        """Diagnostic: bracket each launch of run_device with HIP events (the calls then wait for their kernels)."""
        if self.lib.vo_ba_set_kernel_timing(self.h, int(bool(on))) != 0:
            raise _lib.VoError(-1, "vo_ba_set_kernel_timing failed")

    def kernel_timing(self):
        """{'k_ba_rot': (ms_total, launches), 'k_ba_pass': (ms_total, launches)} since timing was switched on."""
        ms = (C.c_double * 2)(); n = (C.c_int * 2)()
        if self.lib.vo_ba_get_kernel_timing(self.h, ms, n) != 0:
            raise _lib.VoError(-1, "vo_ba_get_kernel_timing failed")
        return {"k_ba_rot": (ms[0], n[0]), "k_ba_pass": (ms[1], n[1])}

    def run_device(self, poses, points, uv, K, r, norm, Hpp, bp, Hll, bl, Hpl, stream=None):
        """Every argument except K is a device address (int), e.g. torch.Tensor.data_ptr(); any output may be 0 to skip it."""
        K = np.ascontiguousarray(K, np.float64).reshape(3, 3)
        vp = lambda a: C.c_void_p(int(a)) if a else None
        rc = self.lib.vo_ba_pass_device(self.h, vp(poses), vp(points), vp(uv), _lib._ptr(K), vp(r), vp(norm), vp(Hpp), vp(bp), vp(Hll), vp(bl),
                                        vp(Hpl), C.c_void_p(stream) if stream else None)
        if rc != 0:
            raise _lib.VoError(rc, "vo_ba_pass_device failed")


def schur_step(out, obs_pose, obs_point, lam=0.0, fixed_poses=()):
    """One damped Gauss-Newton (Levenberg-Marquardt) step from the blocks a pass returns: the caller side of
    /root/reference/vo.py:112-135, where the reference lets scipy `least_squares` differentiate numerically.
    Solves  [Hpp + lam D  Hpl; Hpl^T  Hll + lam D] [dp; dl] = -[bp; bl]  by the Schur complement on the points (each Hll block is
    3 x 3): S dp = -bp + sum Hpl Hll^-1 bl, then dl = Hll^-1 (-bl - Hpl^T dp).  Host numpy, dense reduced camera system
    (6P x 6P): meant for the reference's problem sizes (a handful of poses) and as a check that the blocks ARE the normal
    equations -- a large problem wants a sparse / device solver on the same blocks.  `fixed_poses`: indices held at zero
    update (gauge).  Returns (dp [P,6], dl [N,3]); poses and points are updated by adding them."""
    Hpp = np.array(out["Hpp"], np.float64); bp = np.asarray(out["bp"], np.float64)
    Hll = np.array(out["Hll"], np.float64); bl = np.asarray(out["bl"], np.float64); Hpl = np.asarray(out["Hpl"], np.float64)
    P, N = len(Hpp), len(Hll)
    if lam > 0:
        for j in range(P):
            Hpp[j] += lam * np.diag(np.maximum(np.diag(Hpp[j]), 1e-12))
        for i in range(N):
            Hll[i] += lam * np.diag(np.maximum(np.diag(Hll[i]), 1e-12))
    seen = np.zeros(N, bool); seen[np.asarray(obs_point)] = True
    Hinv = np.zeros_like(Hll)
    Hinv[seen] = np.linalg.inv(Hll[seen])
    S = np.zeros((6 * P, 6 * P)); g = -bp.reshape(-1).copy()
    for j in range(P):
        S[6 * j:6 * j + 6, 6 * j:6 * j + 6] = Hpp[j]
    by_point = {}
    for o, (j, i) in enumerate(zip(np.asarray(obs_pose), np.asarray(obs_point))):
        by_point.setdefault(int(i), []).append((int(j), o))
    for i, lst in by_point.items():
        for j, o in lst:
            WH = Hpl[o] @ Hinv[i]                                    # 6 x 3
            g[6 * j:6 * j + 6] += WH @ bl[i]
            for k, o2 in lst:
                S[6 * j:6 * j + 6, 6 * k:6 * k + 6] -= WH @ Hpl[o2].T
    free = np.ones(6 * P, bool)
    for j in fixed_poses:
        free[6 * j:6 * j + 6] = False
    dp = np.zeros(6 * P)
    dp[free] = np.linalg.solve(S[np.ix_(free, free)], g[free])
    dp = dp.reshape(P, 6)
    dl = np.zeros((N, 3))
    for i, lst in by_point.items():
        rhs = -bl[i].copy()
        for j, o in lst:
            rhs -= Hpl[o].T @ dp[j]
        dl[i] = Hinv[i] @ rhs
    return dp, dl


def allreduce_pose_blocks(Hpp, bp, force=False):
    """Sum the pose blocks of every rank (observations sharded by point): ONE all_reduce of the
    packed [P,42] float64 buffer (36 J^T J + 6 J^T r per pose; 336 KB at P = 1000) -- the only
    collective of the BA pass.  Tensors (any device) or numpy arrays; returns the same kind.
    Without an initialised process group the inputs are returned unchanged."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        return Hpp, bp
    is_np = isinstance(Hpp, np.ndarray)
    tH = torch.from_numpy(np.ascontiguousarray(Hpp)) if is_np else Hpp
    tb = torch.from_numpy(np.ascontiguousarray(bp)) if is_np else bp
    P = tH.shape[0]
    dev = tH.device
    if dist.get_backend() == "nccl" and dev.type != "cuda":
        dev = torch.device("cuda", torch.cuda.current_device())
    buf = torch.cat([tH.reshape(P, 36), tb.reshape(P, 6)], 1).to(dev).contiguous()
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    H2 = buf[:, :36].reshape(P, 6, 6)
    b2 = buf[:, 36:].reshape(P, 6)
    if is_np:
        return H2.cpu().numpy(), b2.cpu().numpy()
    return H2.to(tH.device), b2.to(tb.device)


def shard_observations(problem, rank, world):
    """Rank's share of a problem: a contiguous block of POINTS and all their observations
    (observations are sorted by point), poses replicated.  Point indices stay global."""
    from .parallel import shard_range
    N = len(problem["points"])
    lo, hi = shard_range(N, rank, world)
    sel = (problem["obs_point"] >= lo) & (problem["obs_point"] < hi)
    out = dict(problem)
    out["obs_pose"] = problem["obs_pose"][sel]
    out["obs_point"] = problem["obs_point"][sel]
    out["uv"] = problem["uv"][sel]
    return out
