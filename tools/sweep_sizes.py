"""One-off sweep (not part of the test suite): detectAndCompute on random frame sizes, GPU vs the CPU oracle, keypoint
records and descriptors bit for bit.  python tools/sweep_sizes.py [count] [seed]"""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
    from oracle import oracle as O
    from visual_slam_amd import _lib, synth
    O.build()
    ctx = _lib.Context(max_w=1400, max_h=640, chunk_pairs=1, max_pairs=1)
    sizes = [(int(rng.integers(64, 1400)), int(rng.integers(64, 640))) for _ in range(count)]
    sizes[:6] = [(64, 64), (65, 67), (1399, 639), (1400, 64), (64, 640), (333, 111)]
    imgs = []
    for k, (w, h) in enumerate(sizes):
        big = synth.frame_a(np.random.default_rng(1000 + k), max(w, 256), max(h, 256))
        imgs.append(np.ascontiguousarray(big[:h, :w]))
    # real imagery: the bundled KITTI frames flipped, negated, contrast-scaled and cropped at random
    from PIL import Image
    gold = os.path.join(ROOT, "tests", "golden")
    for d, n in (("kitti02", 4), ("kitti09", 2)):
        for i in range(n):
            g = np.asarray(Image.open(os.path.join(gold, d, "%06d.png" % i)))
            for v in (g, g[:, ::-1], g[::-1], 255 - g, (g * 0.6 + 40).astype(np.uint8), np.clip(g.astype(np.int32) * 2 - 100, 0, 255).astype(np.uint8)):
                v = np.ascontiguousarray(v)
                imgs.append(v); sizes.append((v.shape[1], v.shape[0]))
                x0, y0 = int(rng.integers(0, 600)), int(rng.integers(0, 150))
                cw, ch = int(rng.integers(64, v.shape[1] - x0)), int(rng.integers(64, v.shape[0] - y0))
                imgs.append(np.ascontiguousarray(v[y0:y0 + ch, x0:x0 + cw])); sizes.append((cw, ch))
    count = len(imgs)
    with ThreadPoolExecutor(16) as ex:
        ref = list(ex.map(lambda im: O.orb_detect_compute(im), imgs))
    bad = 0
    for (w, h), im, (ko, do) in zip(sizes, imgs, ref):
        try:
            kg, dg = ctx.orb_detect_compute(im)
        except Exception as e:           # sizes whose upper levels are too small are rejected by both sides or by neither
            print("size", w, h, "gpu error:", e, "oracle n", len(ko)); bad += 1; continue
        ok = len(kg) == len(ko) and all(np.array_equal(kg[f], ko[f]) for f in ko.dtype.names) and np.array_equal(dg, do)
        if not ok:
            bad += 1
            print("MISMATCH size", w, h, "n gpu", len(kg), "n oracle", len(ko), flush=True)
    ctx.close()
    print("RESULT: %d sizes, %d mismatches, keypoints per frame %d..%d" % (count, bad, min(len(r[0]) for r in ref), max(len(r[0]) for r in ref)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
