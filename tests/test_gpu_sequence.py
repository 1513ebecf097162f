"""Sequence entry point (feature reuse across consecutive pairs) and the
trajectory on the bundled KITTI frames (N1), through the C ABI on the GPU."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _records_equal(a, b):
    return all(np.array_equal(a[f], b[f]) for f in a.dtype.names)


@pytest.mark.parametrize("chunk", [1, 2, 8])
def test_sequence_equals_independent_pairs(kitti02, K02, chunk):
    from visual_slam_amd import _lib
    imgs = np.stack(kitti02)
    F = len(imgs)
    ctx = _lib.Context(max_w=imgs.shape[2], max_h=imgs.shape[1], chunk_pairs=chunk, max_pairs=F - 1)
    seq = ctx.sequence_run(imgs, K02)
    ref = ctx.pairs_run(imgs[:-1], imgs[1:], K02)
    ctx.close()
    assert len(seq) == F - 1 and (seq["status"] == 0).all()
    assert _records_equal(seq, ref)


def test_sequence_device_pointer_and_small_pose_batches(kitti02, K02):
    import torch
    from visual_slam_amd import _lib
    imgs = np.stack(kitti02)
    F, h, w = imgs.shape
    d = torch.from_numpy(imgs).cuda()
    ctx = _lib.Context(max_w=w, max_h=h, chunk_pairs=2, max_pairs=2)   # pose stage in batches of 2 pairs
    seq = ctx.sequence_run_device(d.data_ptr(), F, w, h, w, K02)
    ref = ctx.pairs_run(imgs[:-1], imgs[1:], K02)
    ctx.close()
    assert _records_equal(seq, ref)


def test_trajectory_on_kitti02_follows_ground_truth():
    from visual_slam_amd import trajectory as tj
    from visual_slam_amd.vo_utils import VisualOdometry
    vo = VisualOdometry(os.path.join(GOLDEN, "kitti02"), False)
    out = tj.run_sequence(vo)
    n = len(out["poses"])
    assert n == len(vo.image_paths) - 1
    for i in range(1, n + 1):
        Tgt = tj.relative_gt(vo.gt_poses, i)
        T = out["transfs"][i - 1]
        assert tj.rotation_angle(T[:3, :3], Tgt[:3, :3]) < 1e-2          # ~1e-2 rad plausibility gate
        assert tj.direction_angle(T[:3, 3], Tgt[:3, 3]) < 1e-1   # minimal-sample E, no refit: noisy direction
    # monocular: unit-length steps, so the path error is bounded by the true step lengths
    assert out["gt_error"] is not None and np.isfinite(out["gt_error"])
    est_dir = out["estimated_path"][-1] / np.linalg.norm(out["estimated_path"][-1])
    gt_dir = out["gt_path"][-1] / np.linalg.norm(out["gt_path"][-1])
    assert float(est_dir @ gt_dir) > 0.99


def test_loader_feeds_sequence_run(kitti02, K02):
    """Frames decoded by the loader, uploaded on its side stream, run chunk by chunk == one sequence_run."""
    import torch
    from visual_slam_amd import _lib, loader
    paths = loader.load_image_paths(os.path.join(GOLDEN, "kitti02"))
    imgs = np.stack(kitti02)
    F, h, w = imgs.shape
    ctx = _lib.Context(max_w=w, max_h=h, chunk_pairs=2, max_pairs=F - 1)
    ref = ctx.sequence_run(imgs, K02)
    got = []
    for d in loader.FrameLoader(paths, chunk=3, overlap=1, workers=2, device=0):
        assert d.is_cuda and d.dtype == torch.uint8
        got.append(ctx.sequence_run_device(d.data_ptr(), d.shape[0], w, h, w, K02))
        torch.cuda.synchronize()
    ctx.close()
    got = np.concatenate(got)
    assert _records_equal(got, ref)


def test_device_images_from_unsynchronised_torch_op(kitti02, K02):
    """Stream-ordering contract (include/vo_mi355.h): with stream=NULL the library's private stream waits for the
    device's default stream, so images still being produced by torch ops when the call is made are read complete."""
    import torch
    from visual_slam_amd import _lib
    imgs = np.stack(kitti02)
    F, h, w = imgs.shape
    ctx = _lib.Context(max_w=w, max_h=h, chunk_pairs=2, max_pairs=F - 1)
    ref = ctx.sequence_run(imgs, K02)
    src = torch.from_numpy(imgs).cuda()
    big = torch.zeros((256, h, w), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for _ in range(3):
        d = torch.zeros_like(src)
        # a long queue of default-stream work ahead of the op that finally produces the frames: no synchronisation
        for _ in range(40):
            big = big + 1
        d.copy_(src.flip(0).flip(0) + (big[:F] - big[:F]))
        got = ctx.sequence_run_device(d.data_ptr(), F, w, h, w, K02)
        assert _records_equal(got, ref)
    ctx.close()


def test_submit_wait_equals_synchronous_calls(kitti02, K02):
    """vo_pairs_submit_device / vo_pairs_wait (two submissions in flight, host-free RANSAC schedule, pinned result staging)
    give the records of vo_pairs_run_device, for different batches in flight at once and across several lanes."""
    import torch
    from visual_slam_amd import _lib, synth
    A, B = synth.make_batch(range(900, 912), workers=1)
    A = np.concatenate([A, np.stack(kitti02[:3])]); B = np.concatenate([B, np.stack(kitti02[1:])])
    flat = np.full_like(A[0], 99)
    A[5] = flat; B[5] = flat                                                # a failing pair inside a batch
    dA, dB = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()
    n, h, w = A.shape
    K = synth.K_KITTI02
    ctx = _lib.Context(max_w=w, max_h=h, chunk_pairs=2, max_pairs=8)       # 4 lanes
    fs = h * w
    spans = [(0, 8), (8, 15), (3, 9), (14, 15), (0, 15 - 7)]
    want = [ctx.pairs_run_device(dA.data_ptr() + a * fs, dB.data_ptr() + a * fs, b - a, w, h, w, K) for a, b in spans]
    got = []
    for k, (a, b) in enumerate(spans):
        ctx.pairs_submit_device(dA.data_ptr() + a * fs, dB.data_ptr() + a * fs, b - a, w, h, w, K)
        if k >= 1:
            got.append(ctx.pairs_wait())
    got.append(ctx.pairs_wait())
    for x, y in zip(want, got):
        assert len(x) == len(y) and _records_equal(x, y)
    assert want[0]["status"][5] != 0 and (np.delete(want[0]["status"], 5) == 0).all()
    # with nothing in flight any more, the pair arrays of the submission waited for last (pairs 0..7) serve vo_pairs_triangulate
    P = np.tile((K @ np.eye(4)[:3])[None], (8, 1, 1))
    X, q1, q2, off = ctx.pairs_triangulate(P, P)
    assert np.array_equal(np.diff(off), np.where(got[-1]["status"] == 0, got[-1]["n_match"], 0)) and len(X) == off[-1]
    with pytest.raises(_lib.VoError):
        ctx.pairs_wait()                                                     # nothing in flight
    # the sequence form: frames 0..8 of A taken as consecutive frames, two submissions in flight
    seq_want = [ctx.sequence_run_device(dA.data_ptr() + a * fs, b - a, w, h, w, K) for a, b in ((0, 9), (4, 7))]
    ctx.sequence_submit_device(dA.data_ptr(), 9, w, h, w, K)
    ctx.sequence_submit_device(dA.data_ptr() + 4 * fs, 3, w, h, w, K)
    assert _records_equal(ctx.pairs_wait(), seq_want[0]) and _records_equal(ctx.pairs_wait(), seq_want[1])
    ctx.pairs_submit_device(dA.data_ptr(), dB.data_ptr(), 4, w, h, w, K)
    with pytest.raises(_lib.VoError):
        ctx.pairs_run_device(dA.data_ptr(), dB.data_ptr(), 4, w, h, w, K)   # synchronous calls are refused while one is in flight
    with pytest.raises(_lib.VoError):
        # refused BEFORE the geometry tables are rebuilt for another frame size (the lanes in flight read them): ADVICE r2
        ctx.pairs_run_device(dA.data_ptr(), dB.data_ptr(), 4, w - 64, h - 32, w, K)
    with pytest.raises(_lib.VoError):
        ctx.sequence_run_device(dA.data_ptr(), 4, w - 64, h - 32, w, K)
    with pytest.raises(_lib.VoError):
        ctx.pairs_run(A[:2, :h - 32, :w - 64], B[:2, :h - 32, :w - 64], K)
    with pytest.raises(_lib.VoError):
        ctx.sequence_run(A[:3, :h - 32, :w - 64], K)
    with pytest.raises(_lib.VoError):
        ctx.pairs_triangulate(P, P)                                          # and so is everything else that shares its buffers
    with pytest.raises(_lib.VoError):
        ctx.pairs_submit_device(dA.data_ptr(), dB.data_ptr(), 9, w, h, w, K)   # more than max_pairs
    assert _records_equal(ctx.pairs_wait(), want[0][:4])                      # the refused calls left the submission intact
    # the per-pair status vo_pairs_triangulate relies on is only written by the batched pose waves: after a single-pair
    # entry point it is stale, and the call is refused instead of emitting points by a leftover status
    ctx.pairs_triangulate(P[:4], P[:4])
    ctx.get_matches(A[0], B[0])
    with pytest.raises(_lib.VoError):
        ctx.pairs_triangulate(P[:1], P[:1])
    ctx.close()


def test_sequence_reprojection_errors_match_the_dropin():
    """vo.py:107-109 for a whole sequence (batched triangulation + the BA residual pass) == the drop-in's per-frame
    calc_reprojection_error on the same points."""
    from visual_slam_amd import trajectory as tj
    from visual_slam_amd.vo_utils import VisualOdometry
    vo = VisualOdometry(os.path.join(GOLDEN, "kitti02"), False)
    out = tj.run_sequence(vo, triangulate=True)
    err, per_pair = tj.reprojection_errors(out, vo.K)
    off = out["point_offsets"]
    assert len(err) == off[-1] and len(per_pair) == 3
    for i, pose in enumerate(out["poses"]):
        a, b = int(off[i]), int(off[i + 1])
        ref = vo.calc_reprojection_error(vo.K, vo.D, tj.rodrigues(pose[:3, :3]), pose[:3, 3], out["points"][a:b], out["q2"][a:b])
        assert np.allclose(err[a:b], ref, rtol=1e-9, atol=1e-9)
        assert abs(per_pair[i] - ref.mean()) <= 1e-9 * max(1.0, ref.mean())


def test_plain_c_consumer_matches_the_python_binding(kitti02, K02, tmp_path):
    """examples/vo_pairs_demo.c (gcc, C99, only include/vo_mi355.h) runs the bundled KITTI frames through vo_sequence_run and
    prints what the ctypes binding returns, number for number."""
    import subprocess
    from visual_slam_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "vo_pairs_demo")
    if not os.path.exists(exe):
        pytest.skip("examples/vo_pairs_demo is not built (make -C examples / __graft_entry__.build())")
    imgs = np.stack(kitti02)
    raw = tmp_path / "frames.raw"
    imgs.tofile(raw)
    F, h, w = imgs.shape
    out = subprocess.run([exe, str(raw), str(w), str(h), str(F), repr(float(K02[0, 0])), repr(float(K02[1, 1])), repr(float(K02[0, 2])), repr(float(K02[1, 2])), "2"],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    c = _lib.Context(max_w=w, max_h=h, chunk_pairs=2, max_pairs=F - 1)
    ref = c.sequence_run(imgs, K02)
    c.close()
    lines = out.stdout.strip().splitlines()
    assert len(lines) == F
    cur = np.eye(4)
    for i, line in enumerate(lines[:-1]):
        t = line.split()
        assert [int(t[3]), int(t[5]), int(t[6]), int(t[8]), int(t[10]), int(t[12])] == \
            [ref[i]["status"], ref[i]["n_kp1"], ref[i]["n_kp2"], ref[i]["n_match"], ref[i]["n_inlier"], ref[i]["n_good"]]
        assert np.array_equal(np.array([float(v) for v in t[14:26]]), ref[i]["T"].reshape(-1)[:12])      # %.17g round-trips doubles
        cur = cur @ np.linalg.inv(ref[i]["T"])
    pos = np.array([float(v) for v in lines[-1].split()[1:4]])
    assert np.allclose(pos, cur[:3, 3], atol=1e-12)
