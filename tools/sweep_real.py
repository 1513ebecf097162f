"""One-off sweep (not part of the test suite; the frames are not committed): consecutive pairs of real KITTI frames
placed under build/kitti_tmp/<sequence>/ through vo_sequence_run vs the CPU oracle (status, keypoint / match / inlier
counts exact, pose < 1e-6 rad), plus detectAndCompute bit for bit on every 4th frame.  python tools/sweep_real.py"""
import os, sys
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def ang(Ra, Rb):
    return float(np.arccos(np.clip((np.trace(Ra.T @ Rb) - 1) / 2, -1, 1)))


def dang(a, b):
    return float(np.arccos(np.clip(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)), -1, 1)))


def main():
    from PIL import Image
    from oracle import oracle as O
    from visual_slam_amd import _lib, synth
    O.build()
    base = os.path.join(ROOT, "build", "kitti_tmp")
    bad = 0; total = 0; dtotal = 0
    for seq in sorted(os.listdir(base)):
        files = sorted(f for f in os.listdir(os.path.join(base, seq)) if f.endswith(".png"))
        imgs = np.stack([np.asarray(Image.open(os.path.join(base, seq, f)).convert("L")) for f in files])
        n, h, w = imgs.shape
        K = synth.config_K(w, h)
        ctx = _lib.Context(max_w=w, max_h=h, chunk_pairs=16, max_pairs=n - 1)
        res = ctx.sequence_run(imgs, K)
        with ThreadPoolExecutor(16) as ex:
            ref = list(ex.map(lambda i: O.pair_run(imgs[i], imgs[i + 1], K), range(n - 1)))
            dref = list(ex.map(lambda i: O.orb_detect_compute(imgs[i]), range(0, n, 4)))
        for i, (rc, T, counts) in enumerate(ref):
            r = res[i]; total += 1
            ok = rc == r["status"]
            if ok and rc == 0:
                ok = [r["n_kp1"], r["n_kp2"], r["n_match"], r["n_inlier"]] == counts.tolist() and \
                     ang(T[:3, :3], r["T"][:3, :3]) < 1e-6 and dang(T[:3, 3], r["T"][:3, 3]) < 1e-6
            if not ok:
                bad += 1
                print("MISMATCH seq", seq, "pair", i, "oracle", rc, counts.tolist(), "gpu", int(r["status"]), [int(r[k]) for k in ("n_kp1", "n_kp2", "n_match", "n_inlier")], flush=True)
        for j, (ko, do) in zip(range(0, n, 4), dref):
            kg, dg = ctx.orb_detect_compute(imgs[j]); dtotal += 1
            if not (len(kg) == len(ko) and all(np.array_equal(kg[f], ko[f]) for f in ko.dtype.names) and np.array_equal(dg, do)):
                bad += 1; print("DETECT MISMATCH seq", seq, "frame", j, flush=True)
        ctx.close()
        print("seq %s: %d frames %dx%d, matches per pair %d..%d, mismatches so far %d" % (seq, n, w, h, res["n_match"].min(), res["n_match"].max(), bad), flush=True)
    print("RESULT: %d real pairs + %d real frames, %d mismatches" % (total, dtotal, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
