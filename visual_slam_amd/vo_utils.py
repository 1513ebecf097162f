"""Drop-in for /root/reference/libs/vo_utils.py: class VisualOdometry with the same
constructor, attributes, method names, argument meaning and return types, whose cv2 calls
are replaced by the MI355X library (libvo_mi355.so, include/vo_mi355.h).

Differences that are deliberate (SURVEY.md "three facts", BASELINE.json north_star):
  * the matcher is exact brute-force Hamming (BFMatcher(NORM_HAMMING)), not the reference's
    RNG-dependent FLANN-LSH (libs/vo_utils.py:31-33,98);
  * recoverPose ignores the RANSAC mask exactly like the reference's call does
    (libs/vo_utils.py:143 passes `mask` in the slot of the output R);
  * `triangulate` additionally accepts (N,2) point arrays and then returns (N,3).
There is no CPU fallback: constructing the object needs a GPU and the built library.
"""
import os

import numpy as np

from . import _lib


def _imread_gray(path):
    """cv2.imread(path, cv2.IMREAD_GRAYSCALE) for 8-bit gray files."""
    from PIL import Image

    im = Image.open(path)
    if im.mode != "L":
        im = im.convert("L")
    return np.asarray(im)


class VisualOdometry():
    def __init__(self, data_dir, debug):
        # Params (libs/vo_utils.py:9-13)
        self.K, self.P = self.load_calib(os.path.join(data_dir, 'calib.txt'))
        self.D = np.zeros(5)
        self.gt_poses = self.load_poses(os.path.join(data_dir, 'poses.txt'))
        self.image_paths = self.load_image_paths(data_dir)
        self.debug = debug
        # cv2.ORB_create(3000) + matcher objects (libs/vo_utils.py:23-33) -> one device context,
        # created lazily at the first frame size seen
        self._ctx = None
        self._ctx_shape = None

    def _context(self, h, w, keypoint_capacity=0):
        if self._ctx is None or self._ctx_shape != (h, w) or keypoint_capacity > self._ctx.kp_cap:
            if self._ctx is not None:
                self._ctx.close()
            self._ctx = _lib.Context(max_w=w, max_h=h, chunk_pairs=1, max_pairs=1, nfeatures=3000, keypoint_capacity=keypoint_capacity)
            self._ctx_shape = (h, w)
        return self._ctx

    @staticmethod
    def load_calib(filepath):
        """Load calibration of camera (libs/vo_utils.py:36-46)"""
        with open(filepath, 'r') as f:
            for line in f:
                if line.startswith('P0:'):  # only read left camera parameters
                    params = np.array(line.strip().split(': ')[1].split(), dtype=np.float64)
                    P = np.reshape(params, (3, 4))
                    K = P[0:3, 0:3]
        return K, P

    @staticmethod
    def load_poses(filepath):
        """Load GT poses (libs/vo_utils.py:49-63)"""
        if not os.path.exists(filepath):
            return None
        poses = []
        with open(filepath, 'r') as f:
            for line in f.readlines():
                T = np.array(line.split(), dtype=np.float64)
                T = T.reshape(3, 4)
                T = np.vstack((T, [0, 0, 0, 1]))
                poses.append(T)
        return poses

    @staticmethod
    def load_image_paths(filepath):
        """Load image paths (libs/vo_utils.py:66-71)"""
        return [os.path.join(filepath, file) for file in sorted(os.listdir(filepath))
                if file.endswith('.png') or file.endswith('.jpg')]

    @staticmethod
    def form_transf(R, t):
        """Form transformation matrix from R and t (libs/vo_utils.py:74-81)"""
        T = np.eye(4, dtype=np.float64)
        T[:3, :3] = R
        T[:3, 3] = t
        return T

    def get_matches(self, i, detector, frame, prev_frame, ratio=0.45):
        """Detect and compute matching keypoints and descriptors from the i-1'th and i'th img
        (libs/vo_utils.py:83-110).  Returns q1, q2 float32 (N,2)."""
        if detector != "ORB":
            # the SIFT branch of the reference is dead code (vo.py:74 hard-codes "ORB")
            raise NotImplementedError("only the ORB branch of get_matches is on the hot path")
        if self.debug:
            print(detector)
        if frame is not None:
            img1, img2 = np.asarray(prev_frame), np.asarray(frame)
        else:
            img1 = _imread_gray(self.image_paths[i - 1])
            img2 = _imread_gray(self.image_paths[i])
        ctx = self._context(*img1.shape)
        # detectAndCompute x2 + knnMatch + ratio loop + the two list comprehensions (:94-108) are ONE library call:
        # both frames are extracted in one launch wave and matched on the device, only q1 / q2 come back
        while True:
            try:
                q1, q2 = ctx.get_matches(img1, img2, ratio)
                break
            except _lib.VoError as e:
                if e.code == _lib.VO_ERR_TOO_FEW:
                    # knnMatch returns fewer than 2 neighbours -> `for m,n in matches` raises in the reference
                    raise ValueError("not enough values to unpack (expected 2 neighbours per match)") from e
                if e.code == _lib.VO_ERR_CAPACITY and ctx.kp_cap < 65535:
                    # retainBest keeps every tie at the cut, so cv2 returns more than nfeatures keypoints for such a frame
                    # (libs/vo_utils.py:94-95 takes whatever comes): same here, with a workspace that holds them
                    ctx = self._context(*img1.shape, keypoint_capacity=min(65535, 2 * ctx.kp_cap))
                    continue
                raise
        if len(q1) == 0:
            return np.float32([]), np.float32([])     # np.float32([]) has shape (0,), as in the reference
        return q1, q2

    def get_pose(self, q1, q2):
        """Calculate transformation from keypoints (libs/vo_utils.py:136-145)"""
        q1 = np.asarray(q1, np.float32).reshape(-1, 2)
        q2 = np.asarray(q2, np.float32).reshape(-1, 2)
        ctx = self._ctx if self._ctx is not None else self._context(376, 1241)
        T, _, _ = ctx.get_pose(q1, q2, self.K)     # one library call: RANSAC + recoverPose + form_transf
        return T

    def triangulate(self, pts1, pts2, P1, P2):
        """Triangulate 3d points from corresponding 2d matching points (libs/vo_utils.py:148-164).
        pts (2,) -> returns (3,) float32 like the reference; pts (N,2) -> (N,3)."""
        P1 = self.K @ np.asarray(P1, np.float64)[:3, :]
        P2 = self.K @ np.asarray(P2, np.float64)[:3, :]
        a = np.asarray(pts1, np.float32)
        single = a.ndim == 1
        ctx = self._ctx if self._ctx is not None else self._context(376, 1241)
        points_4D = ctx.triangulate(P1, P2, a.reshape(-1, 2), np.asarray(pts2, np.float32).reshape(-1, 2))
        points_3D = points_4D / points_4D[3]
        points_3D = points_3D[:3, :].T
        return points_3D.flatten() if single else points_3D

    def calc_reprojection_error(self, K, D, rvec, tvec, objp, imgpoints):
        """Reprojection error as 2D Euclidean distance (libs/vo_utils.py:184-216).  Host numpy:
        this runs only under debug/ba in the reference."""
        rvec = np.asarray(rvec, np.float64).reshape(3)
        tvec = np.asarray(tvec, np.float64).reshape(3)
        th = np.linalg.norm(rvec)
        if th < 1e-300:
            R = np.eye(3)
        else:
            k = rvec / th
            Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
            R = np.cos(th) * np.eye(3) + (1 - np.cos(th)) * np.outer(k, k) + np.sin(th) * Kx
        X = np.asarray(objp, np.float64).reshape(-1, 3) @ R.T + tvec
        x = X[:, 0] / X[:, 2]
        y = X[:, 1] / X[:, 2]
        D = np.concatenate([np.asarray(D, np.float64).ravel(), np.zeros(5)])[:5]
        k1, k2, p1, p2, k3 = D
        r2 = x * x + y * y
        radial = 1 + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2
        xd = x * radial + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        yd = y * radial + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        K = np.asarray(K, np.float64)
        proj_pts = np.stack([K[0, 0] * xd + K[0, 2], K[1, 1] * yd + K[1, 2]], 1)
        img_pts = np.asarray(imgpoints).squeeze()
        err_sq = np.power(proj_pts - img_pts, 2)
        err_sq = np.sum(err_sq, axis=1)
        return np.sqrt(err_sq).ravel()

    def bundle_adjustment_residuals(self, params, K, D, imgpoints):
        """Residual function for bundle adjustment between consecutive frames
        (libs/vo_utils.py:218-245)."""
        rvec, tvec = params[:3], params[3:6]
        points3d = params[6:].reshape((-1, 3))
        return self.calc_reprojection_error(K, D, rvec, tvec, points3d, imgpoints)
