"""Per-kernel launch times of library variants: python tools/exp_time.py build/exp/libvo_A.so [more .so ...]
Each variant runs in a child process (a ctypes library cannot be unloaded); 32 synthetic pairs per launch."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(so):
    import numpy as np, torch
    from visual_slam_amd import _lib, synth
    if so != "-":
        _lib.SO_PATH = os.path.abspath(so)
    lib = _lib.load()
    n = int(os.environ.get("EXP_N", "32"))
    if os.environ.get("EXP_KITTI"):      # the bundled KITTI frames instead of the synthetic generator
        from visual_slam_amd import loader
        fr = [loader.decode_gray(p) for p in loader.load_image_paths(os.path.join(ROOT, "tests", "golden", "kitti02"))]
        K = synth.K_KITTI02
        A = np.stack([fr[i % 3] for i in range(n)]); B = np.stack([fr[i % 3 + 1] for i in range(n)])
    else:
        pairs = [synth.make_pair(seed=s) for s in range(8)]
        K = pairs[0][2]
        A = np.stack([pairs[i % 8][0] for i in range(n)]); B = np.stack([pairs[i % 8][1] for i in range(n)])
    dA, dB = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()
    ctx = _lib.Context(max_w=A.shape[2], max_h=A.shape[1], chunk_pairs=n, max_pairs=n, nfeatures=3000)
    h, w = A.shape[1:]
    run = lambda: ctx.pairs_run_device(dA.data_ptr(), dB.data_ptr(), n, w, h, w, K)
    run()
    ctx.set_kernel_timing(True)
    for _ in range(5):
        r = run()
    print(json.dumps({k: round(ms / max(nl, 1) * 1000, 1) for k, (ms, nl) in ctx.kernel_timing().items() if nl}))
    print("checksum", int(r["n_match"].sum()), int(r["n_inlier"].sum()), float(np.abs(r["T"]).sum()))


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        for so in sys.argv[1:]:
            print("==", so, flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", so], check=False, timeout=300)
