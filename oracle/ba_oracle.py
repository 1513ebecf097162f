"""TEST INFRASTRUCTURE ONLY -- CPU oracle of the bundle-adjustment residual / Jacobian pass
(BASELINE.json configs[4]).  PARITY UNPINNED: the reference has no tests
for this path and cv2 is not installable here.

What it restates:
  * the residual of /root/reference/libs/vo_utils.py:184-216 (`calc_reprojection_error`):
    cv2.projectPoints(objp, rvec, tvec, K, D) with D = 0, i.e. Rodrigues(rvec), X_c = R X + t,
    u = fx X_c.x / X_c.z + cx, v = fy X_c.y / X_c.z + cy, then the 2-vector (u, v) - observed
    (the reference's scalar is its Euclidean norm, `norm` below);
  * the Jacobian that scipy's least_squares (vo.py:118) estimates by finite differences of
    `bundle_adjustment_residuals` (libs/vo_utils.py:218-245) -- here analytic, generalised from
    the reference's one camera to P poses, with the normal-equation blocks
        Hpp[j] = sum_obs Jp^T Jp (6x6), bp[j] = sum_obs Jp^T r (6),
        Hll[i] = sum_obs Jl^T Jl (3x3), bl[i] = sum_obs Jl^T r (3), Hpl[o] = Jp^T Jl (6x3).
Only tests/ and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np


def rodrigues_matrix(r):
    """cv2.Rodrigues(rvec) -> R (vector -> matrix branch)."""
    r = np.asarray(r, np.float64).reshape(3)
    th = np.sqrt(r @ r)
    if th < np.finfo(np.float64).eps:
        return np.eye(3)
    k = r / th
    c, s = np.cos(th), np.sin(th)
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return c * np.eye(3) + (1 - c) * np.outer(k, k) + s * Kx


def project(pose, X, K):
    R = rodrigues_matrix(pose[:3])
    Xc = R @ X + pose[3:6]
    return np.array([K[0, 0] * Xc[0] / Xc[2] + K[0, 2], K[1, 1] * Xc[1] / Xc[2] + K[1, 2]])


def obs_jacobians(pose, X, K):
    """(r2 without the observation, Jp 2x6, Jl 2x3) for one observation, analytic."""
    v = np.asarray(pose[:3], np.float64)
    R = rodrigues_matrix(v)
    Y = R @ X
    Xc = Y + pose[3:6]
    fx, fy = K[0, 0], K[1, 1]
    iz = 1.0 / Xc[2]
    x, y = Xc[0] * iz, Xc[1] * iz
    Jproj = np.array([[fx * iz, 0, -fx * x * iz], [0, fy * iz, -fy * y * iz]])
    th2 = v @ v
    dYdr = np.zeros((3, 3))
    if th2 < np.finfo(np.float64).eps ** 2:
        for i in range(3):
            e = np.zeros(3); e[i] = 1
            dYdr[:, i] = np.cross(e, Y)
    else:
        vxY = np.cross(v, Y)
        for i in range(3):
            e = np.zeros(3); e[i] = 1
            w = np.cross(v, e - R[:, i])
            dYdr[:, i] = (v[i] * vxY + np.cross(w, Y)) / th2
    Jp = np.hstack([Jproj @ dYdr, Jproj])
    Jl = Jproj @ R
    return np.array([fx * x + K[0, 2], fy * y + K[1, 2]]), Jp, Jl


def ba_pass(poses, points, obs_pose, obs_point, uv, K):
    """Returns dict(r [M,2], norm [M], Hpp [P,6,6], bp [P,6], Hll [N,3,3], bl [N,3], Hpl [M,6,3])."""
    poses = np.asarray(poses, np.float64).reshape(-1, 6)
    points = np.asarray(points, np.float64).reshape(-1, 3)
    uv = np.asarray(uv, np.float64).reshape(-1, 2)
    K = np.asarray(K, np.float64).reshape(3, 3)
    M = len(obs_pose)
    r = np.zeros((M, 2)); Hpl = np.zeros((M, 6, 3))
    Hpp = np.zeros((len(poses), 6, 6)); bp = np.zeros((len(poses), 6))
    Hll = np.zeros((len(points), 3, 3)); bl = np.zeros((len(points), 3))
    for o in range(M):
        j, i = int(obs_pose[o]), int(obs_point[o])
        p, Jp, Jl = obs_jacobians(poses[j], points[i], K)
        r[o] = p - uv[o]
        Hpp[j] += Jp.T @ Jp; bp[j] += Jp.T @ r[o]
        Hll[i] += Jl.T @ Jl; bl[i] += Jl.T @ r[o]
        Hpl[o] = Jp.T @ Jl
    return {"r": r, "norm": np.sqrt((r * r).sum(1)), "Hpp": Hpp, "bp": bp, "Hll": Hll, "bl": bl, "Hpl": Hpl}
