"""One-off wide parity sweep (not part of the test suite): N seeded synthetic pairs through vo_pairs_run vs the CPU
oracle, pair by pair -- status and the keypoint / match / inlier counts exactly, pose within 1e-6 rad.
python tools/sweep_parity.py [first_seed] [count] [threads]   (the oracle runs on host threads; ctypes releases the GIL)"""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def ang(Ra, Rb):
    return float(np.arccos(np.clip((np.trace(Ra.T @ Rb) - 1) / 2, -1, 1)))


def dang(a, b):
    return float(np.arccos(np.clip(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)), -1, 1)))


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    threads = int(sys.argv[3]) if len(sys.argv) > 3 else 14
    from oracle import oracle as O
    from visual_slam_amd import _lib, synth
    O.build()
    K = synth.K_KITTI02
    bad = 0
    t0 = time.time()
    for s0 in range(first, first + count, 128):
        seeds = list(range(s0, min(s0 + 128, first + count)))
        a, b = synth.make_batch(seeds, workers=1)
        c = _lib.Context(max_w=a.shape[2], max_h=a.shape[1], chunk_pairs=32, max_pairs=len(seeds))
        try:
            res = c.pairs_run(a, b, K)
        finally:
            c.close()
        with ThreadPoolExecutor(threads) as ex:
            ref = list(ex.map(lambda i: O.pair_run(a[i], b[i], K), range(len(seeds))))
        for i, (rc, T, counts) in enumerate(ref):
            r = res[i]
            ok = rc == r["status"]
            if ok and rc == 0:
                ok = [r["n_kp1"], r["n_kp2"], r["n_match"], r["n_inlier"]] == counts.tolist() and \
                     ang(T[:3, :3], r["T"][:3, :3]) < 1e-6 and dang(T[:3, 3], r["T"][:3, 3]) < 1e-6
            if not ok:
                bad += 1
                print("MISMATCH seed", seeds[i], "oracle", rc, counts.tolist(), "gpu", int(r["status"]), [int(r[k]) for k in ("n_kp1", "n_kp2", "n_match", "n_inlier")], flush=True)
        print("seeds %d..%d done, %d mismatches so far, %.0f s" % (seeds[0], seeds[-1], bad, time.time() - t0), flush=True)
    print("RESULT: %d pairs, %d mismatches" % (count, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
