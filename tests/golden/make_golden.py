#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (oracle = restatement of OpenCV 4.8.0, not
cv2 itself: PARITY UNPINNED).  Inputs are the KITTI frames committed
beside this script and one seeded synthetic pair.  Re-run: python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from PIL import Image  # noqa: E402

from oracle import oracle as O  # noqa: E402
from visual_slam_amd import synth  # noqa: E402


def pair_record(a, b, K):
    k1, f1 = O.orb_detect_compute(a)
    k2, f2 = O.orb_detect_compute(b)
    i1, d1, i2, d2 = O.match_knn2(f1, f2)
    keep = O.ratio_filter(d1, d2, 0.45)
    q1 = np.stack([k1["x"][keep], k1["y"][keep]], 1).astype(np.float32)
    q2 = np.stack([k2["x"][i1[keep]], k2["y"][i1[keep]]], 1).astype(np.float32)
    E, mask, ninl, iters = O.find_essential_ransac(q1, q2, K)
    R, t, ngood = O.recover_pose(E, q1, q2, K)
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = t
    cur = np.linalg.inv(T)
    X4 = O.triangulate_f32(K @ np.eye(4)[:3], K @ cur[:3], q1, q2)
    return dict(kps1=k1, desc1=f1, kps2=k2, desc2=f2, qidx=keep, tidx=i1[keep], d1=d1[keep], d2=d2[keep],
                E=E, mask=mask, R=R, t=t, ngood=np.int32(ngood), iters=np.int32(iters), X4=X4, K=K)


def main():
    K02 = np.array([[718.856, 0, 607.1928], [0, 718.856, 185.2157], [0, 0, 1.0]])
    a = np.asarray(Image.open(os.path.join(HERE, "kitti02", "000000.png")))
    b = np.asarray(Image.open(os.path.join(HERE, "kitti02", "000001.png")))
    np.savez_compressed(os.path.join(HERE, "kitti02_pair_0_1.npz"), **pair_record(a, b, K02))
    a, b, K, _, _ = synth.make_pair(0)
    np.savez_compressed(os.path.join(HERE, "synth_seed0.npz"), **pair_record(a, b, K))


if __name__ == "__main__":
    main()
