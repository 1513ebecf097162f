"""Where a five-point solve spends its cycles: tools/exp_build.sh clk "-DFP_CLOCKS=1" vo_pose.hip, then
python tools/exp_solver_clocks.py build/exp/libvo_clk.so   (the solver stamps the cycle counter at its phase boundaries; 16 cases =
one RANSAC round of a single pair, in the one-lane and the 16-lanes-per-hypothesis form)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visual_slam_amd import _lib

_lib.SO_PATH = os.path.abspath(sys.argv[1])
ctx = _lib.Context(max_w=640, max_h=480, chunk_pairs=1, max_pairs=1)
rng = np.random.default_rng(0)
X = rng.standard_normal((16, 3)); X[:, 2] += 5
x1 = X[:, :2] / X[:, 2:]; Y = X + np.array([0.3, 0.02, 0.1]); x2 = Y[:, :2] / Y[:, 2:]
idx = [rng.choice(16, 5, replace=False) for _ in range(16)]
a = np.stack([x1[i] for i in idx]); b = np.stack([x2[i] for i in idx]) + 0.001 * rng.standard_normal((16, 5, 2))
for wide in (False, True):
    for _ in range(3):
        cnt, E = ctx.debug_five_point(a, b, wide=wide)
    tk = E.reshape(16, 90)[:, 80:87]
    d = np.median(np.diff(tk, axis=1), axis=0).astype(int)
    print("%-9s models %.1f; cycles [null space, polynomials, elimination, b + det, roots, back-substitution] = %s, total %d"
          % ("16 lanes" if wide else "one lane", cnt.mean(), d.tolist(), int(np.median(tk[:, 6] - tk[:, 0]))))
