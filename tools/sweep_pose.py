"""One-off sweep (not part of the test suite): findEssentialMat + recoverPose + triangulatePoints + matcher on random
inputs, GPU vs the CPU oracle.  Point sets of 5..3000 correspondences, 0..85 % outliers, noise 0..1.5 px, also planar
scenes and pure rotations; descriptor sets of random sizes.  python tools/sweep_pose.py [count] [seed]"""
import os, sys
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rodrigues(r):
    th = np.linalg.norm(r)
    if th < 1e-12:
        return np.eye(3)
    k = r / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.cos(th) * np.eye(3) + (1 - np.cos(th)) * np.outer(k, k) + np.sin(th) * Kx


def scene(rng, K):
    n = int(rng.choice([5, 6, 7, 8, 12, 30, 100, 300, 700, 1500, 3000]))
    kind = rng.choice(["general", "planar", "rotation", "forward"])
    X = np.stack([rng.uniform(-8, 8, n), rng.uniform(-3, 3, n), rng.uniform(6, 40, n)], 1)
    if kind == "planar":
        X[:, 2] = 20 + 0.1 * X[:, 0]
    R = rodrigues(rng.normal(0, 0.05, 3))
    t = rng.normal(0, 1, 3); t /= np.linalg.norm(t)
    if kind == "rotation":
        t = t * 1e-9
    if kind == "forward":
        t = np.array([0.02, -0.01, -1.0]); t /= np.linalg.norm(t)
    x1 = (K @ X.T).T; x1 = x1[:, :2] / x1[:, 2:]
    X2 = (R @ X.T).T + t
    x2 = (K @ X2.T).T; x2 = x2[:, :2] / x2[:, 2:]
    noise = float(rng.choice([0, 0.1, 0.5, 1.5]))
    if noise:
        x1 = x1 + rng.normal(0, noise, x1.shape); x2 = x2 + rng.normal(0, noise, x2.shape)
    no = int(n * float(rng.choice([0, 0.1, 0.3, 0.6, 0.85]))) if n > 8 else 0
    if no:
        x2[:no] = rng.uniform(0, 1200, (no, 2))
    return x1.astype(np.float32), x2.astype(np.float32), R, t, kind, n, no, noise


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
    from oracle import oracle as O
    from visual_slam_amd import _lib, synth
    O.build()
    K = synth.K_KITTI02
    ctx = _lib.Context(max_w=1241, max_h=376, chunk_pairs=1, max_pairs=1)
    cases = [scene(rng, K) for _ in range(count)]

    def ref(c):
        try:
            E, m, ni, it = O.find_essential_ransac(c[0], c[1], K)
        except RuntimeError:
            return None
        R, t, g = O.recover_pose(E, c[0], c[1], K)
        P1 = K @ np.eye(4)[:3]; P2 = K @ np.c_[R, t]
        return E, m, ni, it, R, t, g, O.triangulate_f32(P1, P2, c[0], c[1])
    with ThreadPoolExecutor(16) as ex:
        refs = list(ex.map(ref, cases))
    bad = 0; rounds = 0
    for c, r in zip(cases, refs):
        tag = (c[4], c[5], c[6], c[7])
        try:
            E, m, ni = ctx.find_essential_ransac(c[0], c[1], K)
        except _lib.VoError as e:
            if r is not None:
                bad += 1; print("MISMATCH", tag, "gpu error", e.code, "oracle ok")
            continue
        if r is None:
            bad += 1; print("MISMATCH", tag, "oracle found no model, gpu did"); continue
        Eo, mo, no_, it, Ro, to, go, Xo = r
        rounds += it > 16
        Rg, tg, gg = ctx.recover_pose(E, c[0], c[1], K)
        P1 = K @ np.eye(4)[:3]; P2 = K @ np.c_[Ro, to]
        Xg = ctx.triangulate(P1, P2, c[0], c[1])
        ok = ni == no_ and np.array_equal(m, mo) and np.abs(E - Eo).max() < 1e-9 and gg == go and np.abs(Rg - Ro).max() < 1e-9 and \
            np.abs(np.asarray(tg).ravel() - np.asarray(to).ravel()).max() < 1e-9 and np.abs(np.abs(Xg) - np.abs(Xo)).max() < 1e-5
        if not ok:
            bad += 1
            print("MISMATCH", tag, "inliers", ni, no_, "good", gg, go, "dE", np.abs(E - Eo).max(), "iters", it, flush=True)
    # matcher: random descriptor sets, clustered so that ties and near-ties are frequent
    mbad = 0
    for k in range(40):
        nq, nt = int(rng.integers(0, 4024)), int(rng.integers(2, 4024))
        base = rng.integers(0, 256, (16, 32), dtype=np.uint8)
        def mk(n):
            d = base[rng.integers(0, 16, n)].copy()
            flips = rng.integers(0, 256, (n, 32), dtype=np.uint8) & rng.integers(0, 256, (n, 32), dtype=np.uint8) & rng.integers(0, 256, (n, 32), dtype=np.uint8)
            return d ^ (flips * (rng.random((n, 1)) < 0.7)).astype(np.uint8)
        q, t = mk(nq), mk(nt)
        g = ctx.match_knn2(q, t); o = O.match_knn2(q, t) if nq else tuple(np.zeros(0, np.int32) for _ in range(4))
        if not all(np.array_equal(a, b) for a, b in zip(g, o)):
            mbad += 1; print("MATCH MISMATCH", nq, nt, flush=True)
    ctx.close()
    print("RESULT: %d pose cases (%d needed more than the first RANSAC round), %d mismatches; 40 matcher cases, %d mismatches" % (count, rounds, bad, mbad))
    return 1 if bad or mbad else 0


if __name__ == "__main__":
    sys.exit(main())
