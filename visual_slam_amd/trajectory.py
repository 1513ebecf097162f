"""Pose chaining, trajectory and ground-truth evaluation.

Host-side mirror of what the per-frame loop of the reference does with the per-pair
transforms the hot path returns (/root/reference/vo.py:77-85, 145-154):

    transf   = nan_to_num(get_pose(q1, q2), neginf=0, posinf=0)      vo.py:77
    cur_pose = cur_pose @ inv(transf)                                 vo.py:80
    rvec, _  = cv2.Rodrigues(cur_pose[:3, :3]); tvec = cur_pose[:3, 3] vo.py:84-85
    estimated_path.append(tvec); gt_path.append(gt_poses[i][:3, 3])   vo.py:145-149
    diff = norm(gt_path - estimated_path, axis=1); mean(diff)         vo.py:152-154

The chaining is a prefix product of 4x4 float64 matrices: numpy on the host, with the same
calls as the reference (np.linalg.inv, @), so the trajectory is bit-identical to the
reference's for identical per-pair transforms.  For a sequence sharded over ranks,
`chain_poses` of every shard starts from identity and `join_shards` applies the carried
pose of the preceding shards (matrix products are associative up to rounding).
"""
import numpy as np


def sanitize(transf):
    """np.nan_to_num(transf, neginf=0, posinf=0) as at vo.py:77."""
    return np.nan_to_num(np.asarray(transf, np.float64), neginf=0, posinf=0)


def chain_poses(transfs, start=None):
    """cur_pose after every pair: list of float64 (4,4), one per transform (vo.py:80-81).

    transfs: iterable of (4,4) relative transforms (`VisualOdometry.get_pose` outputs or the
    `T` field of `Context.pairs_run` / `sequence_run` results).  A singular transform (a failed
    pair: all zeros after `sanitize`) raises numpy.linalg.LinAlgError exactly as the reference
    does.
    """
    cur = np.eye(4) if start is None else np.array(start, np.float64)
    poses = []
    for t in transfs:
        cur = cur @ np.linalg.inv(sanitize(t))
        poses.append(cur)
    return poses


def join_shards(shard_poses):
    """Trajectory of a sequence whose consecutive pairs were chained per shard from identity.

    shard_poses: list (one entry per rank, in rank order) of lists of (4,4) poses as returned
    by `chain_poses`.  Returns the single list the unsharded chain would give, up to rounding
    of the re-associated products."""
    out = []
    carry = np.eye(4)
    for poses in shard_poses:
        for p in poses:
            out.append(carry @ p)
        if poses:
            carry = carry @ poses[-1]
    return out


def rodrigues(R):
    """cv2.Rodrigues(R) for a 3x3 rotation matrix -> rvec (3,1) float64 (vo.py:84).

    Restates OpenCV 4.8 calib3d `cvRodrigues2` (matrix -> vector branch): R is first replaced
    by the nearest rotation U V^T of its SVD, then theta = acos((tr R - 1)/2) and the axis is
    taken from the skew part; near theta = pi the axis comes from the diagonal."""
    R = np.asarray(R, np.float64).reshape(3, 3)
    U, _, Vt = np.linalg.svd(R)
    R = U @ Vt
    rx, ry, rz = R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]
    s = np.sqrt((rx * rx + ry * ry + rz * rz) * 0.25)
    c = (R[0, 0] + R[1, 1] + R[2, 2] - 1) * 0.5
    c = 1.0 if c > 1.0 else (-1.0 if c < -1.0 else c)
    theta = np.arccos(c)
    if s < 1e-5:
        if c > 0:
            r = np.zeros(3)
        else:
            t = (R[0, 0] + 1) * 0.5
            rx = np.sqrt(max(t, 0.0))
            t = (R[1, 1] + 1) * 0.5
            ry = np.sqrt(max(t, 0.0)) * (-1.0 if R[0, 1] < 0 else 1.0)
            t = (R[2, 2] + 1) * 0.5
            rz = np.sqrt(max(t, 0.0)) * (-1.0 if R[0, 2] < 0 else 1.0)
            if abs(rx) < abs(ry) and abs(rx) < abs(rz) and (R[1, 2] > 0) != (ry * rz > 0):
                rz = -rz
            theta /= np.sqrt(rx * rx + ry * ry + rz * rz)
            r = np.array([rx, ry, rz]) * theta
    else:
        vth = 1.0 / (2.0 * s)
        vth *= theta
        r = np.array([rx, ry, rz]) * vth
    return r.reshape(3, 1)


def estimated_path(poses):
    """tvec of every chained pose (vo.py:85,145) as float64 (N,3)."""
    return np.array([p[:3, 3] for p in poses], np.float64).reshape(-1, 3)


def gt_path(gt_poses, first=1, count=None):
    """Ground-truth positions gt_poses[i][:3, 3] for frames first..first+count-1 (vo.py:148-149)."""
    sel = gt_poses[first:] if count is None else gt_poses[first:first + count]
    return np.array([g[:3, 3] for g in sel], np.float64).reshape(-1, 3)


def gt_error(est, gt):
    """mean over frames of ||gt - est|| (vo.py:152-154); est and gt are (N,3)."""
    est = np.asarray(est, np.float64)
    gt = np.asarray(gt, np.float64)
    diff = np.linalg.norm(gt - est, axis=1)
    return float(np.mean(diff)), diff


def relative_gt(gt_poses, i):
    """Ground-truth relative transform that get_pose estimates for the pair (i-1, i):
    inv(gt[i]) @ gt[i-1] (so that cur_pose @ inv(transf) follows gt)."""
    return np.linalg.inv(gt_poses[i]) @ gt_poses[i - 1]


def rotation_angle(Ra, Rb):
    """Angle (rad) of Ra^T Rb."""
    c = (np.trace(np.asarray(Ra).T @ np.asarray(Rb)) - 1) / 2
    return float(np.arccos(np.clip(c, -1, 1)))


def direction_angle(ta, tb):
    """Angle (rad) between two translation directions."""
    ta = np.asarray(ta, np.float64).ravel(); tb = np.asarray(tb, np.float64).ravel()
    c = ta @ tb / (np.linalg.norm(ta) * np.linalg.norm(tb))
    return float(np.arccos(np.clip(c, -1, 1)))


def projection_pairs(K, poses, start=None):
    """(P1, P2) float64 (N,3,4) that `VisualOdometry.triangulate(q1, q2, prev_pose, cur_pose)` forms for every frame of the
    loop (vo.py:92 with libs/vo_utils.py:153-157): P1 = K @ prev_pose[:3], P2 = K @ cur_pose[:3], where prev_pose is the
    chained pose before the pair (identity, or `start`, for the first) and cur_pose the one after it."""
    K = np.asarray(K, np.float64)
    prev = np.eye(4) if start is None else np.array(start, np.float64)
    P1, P2 = [], []
    for cur in poses:
        P1.append(K @ prev[:3, :])
        P2.append(K @ np.asarray(cur, np.float64)[:3, :])
        prev = cur
    return np.array(P1).reshape(-1, 3, 4), np.array(P2).reshape(-1, 3, 4)


def run_sequence(vo, max_frames=None, triangulate=False, wave=256, chunk=64):
    """The reference's main loop (vo.py:48-190) without display and BA, on the batched GPU path: the consecutive pairs of
    vo.image_paths go through `sequence_run` in waves of `wave` pairs; the poses are chained on the host exactly as
    vo.py:77-81 does; with triangulate=True every match of every pair is triangulated (vo.py:91-93) by ONE
    `pairs_triangulate` launch per wave instead of 2 N Python -> cv2 calls per frame.

    Returns a dict: transfs (N,4,4), poses (list of N (4,4)), estimated_path (N,3), gt_path ((N,3) or None), gt_error
    (mean, or None), results (the raw per-pair records) and, with triangulate, points (M,3) float32 = the reference's
    Q list, q1 / q2 (M,2), point_offsets (N+1): pair i owns rows point_offsets[i] .. point_offsets[i+1]-1."""
    from . import _lib, vo_utils
    paths = vo.image_paths if max_frames is None else vo.image_paths[:max_frames]
    imgs = np.stack([vo_utils._imread_gray(p) for p in paths])
    npairs = len(imgs) - 1
    wave = max(1, min(wave, npairs))
    ctx = _lib.Context(max_w=imgs.shape[2], max_h=imgs.shape[1], chunk_pairs=max(1, min(chunk, wave)), max_pairs=wave, nfeatures=3000)
    res_all, poses, pts, q1s, q2s, offs = [], [], [], [], [], [0]
    try:
        cur = np.eye(4)
        for w0 in range(0, npairs, wave):
            w1 = min(w0 + wave, npairs)
            res = ctx.sequence_run(imgs[w0:w1 + 1], vo.K)
            bad = np.nonzero(res["status"] != 0)[0]
            if len(bad):
                raise RuntimeError("pair %d failed with status %d" % (w0 + int(bad[0]), int(res["status"][bad[0]])))
            wposes = chain_poses(res["T"].reshape(-1, 4, 4), start=cur)
            if triangulate:
                P1, P2 = projection_pairs(vo.K, wposes, start=cur)
                X, q1, q2, off = ctx.pairs_triangulate(P1, P2)
                pts.append(X); q1s.append(q1); q2s.append(q2)
                offs.extend((off[1:] + offs[-1]).tolist())
            cur = wposes[-1]
            poses.extend(wposes)
            res_all.append(res)
    finally:
        ctx.close()
    res = np.concatenate(res_all)
    transfs = res["T"].reshape(-1, 4, 4)
    est = estimated_path(poses)
    out = {"transfs": transfs, "poses": poses, "estimated_path": est, "gt_path": None, "gt_error": None, "results": res}
    if triangulate:
        out.update(points=np.concatenate(pts), q1=np.concatenate(q1s), q2=np.concatenate(q2s), point_offsets=np.array(offs, np.int64))
    if vo.gt_poses:
        g = gt_path(vo.gt_poses, 1, len(poses))
        out["gt_path"] = g
        out["gt_error"] = gt_error(est[:len(g)], g)[0]
    return out


def reprojection_errors(out, K, device=0):
    """The reference's per-frame debug figure (vo.py:107-109) for a whole sequence in ONE launch:
    `calc_reprojection_error(K, D=0, rvec_pose, tvec_pose, Q_local, q2)` of every triangulated point of every pair, with
    (rvec_pose, tvec_pose) = (Rodrigues(cur_pose[:3,:3]), cur_pose[:3,3]) as vo.py:84-85 forms them (the chained pose is used
    as is, like the reference does).  `out` is what `run_sequence(..., triangulate=True)` returns.  Evaluated by the BA
    residual pass (`vo_ba_pass`): one observation per point, pose index = its pair.
    Returns (per-point error float64 (M,), per-pair mean error float64 (N,), NaN for a pair without points)."""
    from . import ba
    poses, off = out["poses"], np.asarray(out["point_offsets"], np.int64)
    n = len(poses)
    M = int(off[-1])
    per_pair = np.full(n, np.nan)
    if M == 0:
        return np.zeros(0), per_pair
    pv = np.zeros((n, 6))
    for i, p in enumerate(poses):
        pv[i, :3] = rodrigues(p[:3, :3]).ravel()
        pv[i, 3:] = p[:3, 3]
    counts = np.diff(off)
    obs_pose = np.repeat(np.arange(n, dtype=np.int32), counts)
    plan = ba.BAPlan(obs_pose, np.arange(M, dtype=np.int32), n, M, device=device)
    try:
        err = plan.run(pv, np.asarray(out["points"], np.float64), np.asarray(out["q2"], np.float64), K)["norm"]
    finally:
        plan.close()
    has = counts > 0
    per_pair[has] = np.add.reduceat(err, off[:-1][has]) / counts[has]
    return err, per_pair
