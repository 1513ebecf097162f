#!/usr/bin/env python3
"""bench.py -- frame-pairs/sec (detect + match + pose) at 1241x376 / 3000 ORB features.

One "step" = one pass of the hot path (vo_pairs_run_device: ORB detect x2, Hamming 2-NN +
ratio, five-point RANSAC, recoverPose) over BASELINE.json configs[2]: a batch of 4096 synthetic
frame pairs (the configs[1] generator, seed = global pair index 0..4095), already resident in
HBM when the timed region starts.  Pairs are independent: the batch is sharded contiguously
over the ranks (visual_slam_amd.parallel.shard_range) with no data-path collective, so the
total work is fixed ("strong" scaling); the only collective is one all_gather of the
152-byte per-pair records per step.  `--pairs-per-gpu B` switches to a weak-scaling
diagnostic (B pairs on every rank).

Contract: python bench.py --gpus N --steps K --warmup W   (N>1 under torch.distributed.run)
prints ONE JSON line on rank 0.  Beside the headline it carries, measured in the same run on rank 0:
`roofline` (dominant byte-streaming kernel, HIP events on the launch stream), `cpu_baseline`
(the CPU oracle on this box's host cores: 1 thread and all cores, per stage, median of 5),
`h2d` and `png_decode` (reported separately: they are not part of `value`), `latency`
(configs[1]: one pair, B=1, and the drop-in class's get_matches -> get_pose), `kitti`
(configs[0]: the bundled KITTI frames, tiled).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0
# matcher: one Hamming distance = a 256-long dot product on the matrix cores = 512 ops.  Since round 3 the operands are fp4
# (v_mfma_scale_f32_32x32x64_f8f6f4): dense FP4 MFMA peak = 4 x bf16 = 256 CU x 4 SIMD x 4096 ops/clk x 2.4 GHz (MI355X_MICROARCH.md,
# Matrix cores: ~10 PF); the i8 kernel of rounds 1-2 was priced against the i8 peak, half of that -- kept beside it for comparison
MFMA_FP4_PEAK_TOPS = 256 * 4 * 4096 * 2.4e9 / 1e12
MFMA_I8_PEAK_TOPS = 256 * 4 * 2048 * 2.4e9 / 1e12
# VALU issue (MI355X_MICROARCH.md: a wave64 VALU instruction issues over 2 cycles, 32 lanes per cycle and SIMD):
# 256 CU x 4 SIMD x 2.4 GHz / 2 = 1.229 T wave64-instructions/s = 78.6 T lane-ops/s at full rate
VALU_PEAK_WAVE_INSTS = 256 * 4 * 2.4e9 / 2
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r3_pmc_hbm_traffic.json")
VALU_FILE = os.path.join(ROOT, "profiles", "r3_pmc_valu.json")
BA_TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r3_ba_pmc_hbm_traffic.json")
KITTI_DIR = os.path.join(ROOT, "tests", "golden", "kitti02")


def level_sizes(w, h, nlevels=8, scale_factor=1.2):
    """(w_l, h_l) of every pyramid level with the rounding of vo_build_geometry / OpenCV:
    scale_l = (float)pow((double)1.2f, l); size = cvRound(size0 * (1.f / scale_l)) in float32."""
    sf = float(np.float32(scale_factor))
    out = []
    for l in range(nlevels):
        inv = np.float32(1.0) / np.float32(sf ** l)
        out.append((int(np.rint(np.float32(w) * inv)), int(np.rint(np.float32(h) * inv))))
    return out


def algo_bytes_per_frame(w, h, nfeat, nlevels=8):
    """Algorithmic HBM bytes per frame of every byte-streaming / gather kernel."""
    px = [a * b for a, b in level_sizes(w, h, nlevels)]
    P = sum(px)
    return {
        "k_pyramid": P,                      # read level 0 + write levels 1..7 (all 7 launches together)
        "k_fast_tiles": P,                   # read every pyramid pixel once
        "k_blur": 2 * P,                     # read + write every pyramid pixel
        "k_rbrief": nfeat * 512 + nfeat * 32,
        "k_orient": nfeat * 749,
        "k_harris": 2 * nfeat * 81,
    }, P


def _pmc_file(path, w, h):
    """A committed PMC summary, or None when it is missing or was taken at another frame size."""
    try:
        with open(path) as f:
            t = json.load(f)
        if (t["meta"].get("w", 1241), t["meta"].get("h", 376)) != (w, h):
            return None
        return t
    except Exception:
        return None


def measured_traffic(kernel, frames_per_launch, w, h):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs of this
    script), scaled to this run's frames per launch.  None when the file is missing or of another frame size: bench.py
    itself never runs the profiler."""
    t = _pmc_file(TRAFFIC_FILE, w, h)
    if t is None or kernel not in t["kernels"]:
        return None
    k = t["kernels"][kernel]
    # gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies 128-B requests at 64 B, so it is doubled;
    # k_blur confirms it for these dword-wide reads (raw 93.7 MB fetched for 184.8 MB that must be read).
    kb = 2.0 * k["FETCH_SIZE_KB_avg_per_launch"] + k["WRITE_SIZE_KB_avg_per_launch"]
    return int(kb * 1024 * frames_per_launch / t["meta"]["frames_per_launch"])


def valu_roofline(kernel, frames_per_launch, avg_launch_s, P, w, h):
    """VALU-issue view of a kernel from the committed SQ_INSTS_VALU pass (wave64 instructions per launch)."""
    t = _pmc_file(VALU_FILE, w, h)
    if t is None or kernel not in t["kernels"]:
        return None
    insts = t["kernels"][kernel]["SQ_INSTS_VALU_avg_per_launch"] * frames_per_launch / t["meta"]["frames_per_launch"]
    achieved = insts / avg_launch_s
    return {"wave_insts_per_launch": int(insts), "insts_per_px": round(insts * 64 / (P * frames_per_launch), 2),
            "achieved": round(achieved / 1e9, 2), "peak": round(VALU_PEAK_WAVE_INSTS / 1e9, 1), "unit": "G wave64-instructions/s",
            "frac": round(achieved / VALU_PEAK_WAVE_INSTS, 4), "issue_frac_at_measured_clock": t["kernels"][kernel].get("valu_issue_frac"),
            "note": "peak = every SIMD issuing one full-rate wave64 VALU instruction per 2 cycles (nominal 2.4 GHz); issue_frac_at_measured_clock = SQ_INSTS_VALU / (16 x SQ_BUSY_CYCLES) from the same PMC pass. Most of the kernel's instructions are half-rate (v_lerp_u8, v_perm, v_min3/v_max3, v_cmp, v_mbcnt, VOP3 forms: 4.4-4.9 cycles against 2.7 for a full-rate one, tools/ubench3.hip), and its LDS pipe is as busy as its VALU (profiles/r3_c_pmc_lds.csv): DESIGN.md 3.3"}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_threads():
    """Threads this job may use: the box's CPU share (OMP_NUM_THREADS is set to it on the GPU box), not the machine's CPUs."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    env = os.environ.get("OMP_NUM_THREADS", "")
    if env.isdigit() and int(env) > 0:
        n = min(n, int(env))
    return max(1, n)


def cpu_baseline(imgsA, imgsB, K, nfeat, repeats=5):
    """BASELINE.md section 2, implementation B: the CPU oracle (a restatement of the cv2 path, oracle/) timed on this
    host on a bounded sample of the same pairs: one thread with a per-stage split, and all cores with one pair per
    thread at a time (ctypes releases the GIL, the oracle is re-entrant); medians of `repeats` repeats after a warm-up."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    O.build()
    par = O.default_params(nfeatures=nfeat)
    n_avail = len(imgsA)

    def staged(i):
        t0 = time.perf_counter()
        k1, f1 = O.orb_detect_compute(imgsA[i], par)
        k2, f2 = O.orb_detect_compute(imgsB[i], par)
        t1 = time.perf_counter()
        i1, d1, i2, d2 = O.match_knn2(f1, f2)
        keep = O.ratio_filter(d1, d2, 0.45)
        q1 = np.stack([k1["x"][keep], k1["y"][keep]], 1)
        q2 = np.stack([k2["x"][i1[keep]], k2["y"][i1[keep]]], 1)
        t2 = time.perf_counter()
        E, _, _, _ = O.find_essential_ransac(q1, q2, K)
        O.recover_pose(E, q1, q2, K)
        t3 = time.perf_counter()
        return t1 - t0, t2 - t1, t3 - t2

    staged(0)                                                  # warm-up
    n1 = min(3, n_avail)
    reps = []
    for r in range(repeats):
        ts = np.array([staged((r * n1 + j) % n_avail) for j in range(n1)])
        reps.append(ts.mean(0))
    reps = np.array(reps)                                      # [repeats][3] seconds per pair
    per_pair = np.median(reps.sum(1))
    one = {"value": round(1.0 / per_pair, 3), "unit": "frame-pairs/sec", "threads": 1,
           "ms_per_pair": round(per_pair * 1e3, 2),
           "per_stage_ms": {"detect_x2": round(float(np.median(reps[:, 0])) * 1e3, 2), "match_ratio": round(float(np.median(reps[:, 1])) * 1e3, 2),
                            "essential_ransac_recover_pose": round(float(np.median(reps[:, 2])) * 1e3, 2)},
           "repeats": repeats, "pairs_per_repeat": n1}

    T = host_threads()
    per_thread = 2
    pool = ThreadPoolExecutor(T)

    def one_pair(i):
        return O.pair_run(imgsA[i % n_avail], imgsB[i % n_avail], K, 0.45, par)[0]

    list(pool.map(one_pair, range(T)))                         # warm-up
    rates = []
    for r in range(repeats):
        t0 = time.perf_counter()
        list(pool.map(one_pair, range(r * T * per_thread, (r + 1) * T * per_thread)))
        rates.append(T * per_thread / (time.perf_counter() - t0))
    pool.shutdown()
    allc = {"value": round(float(np.median(rates)), 3), "unit": "frame-pairs/sec", "threads": T, "repeats": repeats, "pairs_per_repeat": T * per_thread,
            "parallelism": "one pair per thread at a time (thread pool over pairs; the C++ oracle runs outside the GIL)"}

    out = {"value": allc["value"], "unit": "frame-pairs/sec", "cores": T, "kind": "port",
           "sample": "%d + %d pair evaluations of rank 0's batch (warm-up excluded): C++ oracle = restated cv2 ORB + BFMatcher(NORM_HAMMING) + ratio + "
                     "findEssentialMat + recoverPose, g++ -O3; value = all-core median of %d repeats" % (repeats * n1, repeats * T * per_thread, repeats),
           "os_cpu_count": os.cpu_count(), "cpu_model": cpu_model(), "single_thread": one, "all_cores": allc}
    try:      # SURVEY.md 8(d) "live cv2": never installed or fetched; timed beside the port only where a box already has it
        import cv2
    except ImportError:
        out["cv2"] = None
        return out

    def cv_pair(i, orb, bf):
        k1, f1 = orb.detectAndCompute(imgsA[i % n_avail], None)
        k2, f2 = orb.detectAndCompute(imgsB[i % n_avail], None)
        good = [a for a, b in bf.knnMatch(f1, f2, k=2) if a.distance < 0.45 * b.distance]
        q1 = np.float32([k1[g.queryIdx].pt for g in good]); q2 = np.float32([k2[g.trainIdx].pt for g in good])
        E, mask = cv2.findEssentialMat(q1, q2, K, cv2.RANSAC, 0.999, 1.0, None)
        cv2.recoverPose(E, q1, q2, K, mask)

    res = {"version": cv2.__version__}
    for label, nthreads in (("setNumThreads_1", 1), ("default_threads", -1)):
        cv2.setNumThreads(nthreads)
        orb, bf = cv2.ORB_create(nfeat), cv2.BFMatcher(cv2.NORM_HAMMING)
        cv_pair(0, orb, bf)
        rr = []
        for r in range(repeats):
            t0 = time.perf_counter()
            for j in range(n1):
                cv_pair(r * n1 + j, orb, bf)
            rr.append(n1 / (time.perf_counter() - t0))
        res[label] = {"value": round(float(np.median(rr)), 3), "unit": "frame-pairs/sec", "repeats": repeats}
    out["cv2"] = res
    return out


def png_decode_leg(repeats=5):
    """PNG decode (cv2.imread's job, libs/vo_utils.py:94-95) of the bundled KITTI frames: ms per frame on one thread and
    the FrameLoader's frames/s with 1 / 4 / 16 decoder threads (host only, numpy output).  Reported separately from `value`."""
    from visual_slam_amd import loader
    paths = loader.load_image_paths(KITTI_DIR)
    if not paths:
        return None
    for p in paths:
        loader.decode_gray(p)
    ts = []
    for _ in range(repeats):
        t0 = time.perf_counter()
        for p in paths:
            loader.decode_gray(p)
        ts.append((time.perf_counter() - t0) / len(paths))
    out = {"ms_per_frame_1thread": round(float(np.median(ts)) * 1e3, 3), "file_kb": round(float(np.mean([os.path.getsize(p) for p in paths])) / 1024, 1),
           "frames": "tests/golden/kitti02 (4 KITTI odometry PNGs, 1241x376, 8-bit gray), Pillow", "loader_frames_per_sec": {}}
    tiled = paths * 64                                          # 256 frames
    for nt in (1, 4, 16):
        t0 = time.perf_counter()
        n = 0
        for chunk in loader.FrameLoader(tiled, chunk=64, overlap=0, workers=nt, device=None):
            n += len(chunk)
        out["loader_frames_per_sec"]["%d_threads" % nt] = round(n / (time.perf_counter() - t0), 1)
    # decoder PROCESSES (spawned; shared-memory ring): Pillow re-takes the GIL ~30 times per PNG, threads stop scaling at 3-4
    np_ = host_threads()
    tiled = paths * 256                                         # 1024 frames: amortises the worker start-up
    ld = iter(loader.FrameLoader(tiled, chunk=64, overlap=0, processes=np_, device=None))
    n = len(next(ld))                                           # first chunk: includes spawning the workers
    t0 = time.perf_counter()
    m = 0
    for chunk in ld:
        m += len(chunk)
    out["loader_frames_per_sec"]["%d_processes" % np_] = round(m / (time.perf_counter() - t0), 1)
    out["note"] = ("decode caps a PNG-fed pipeline far below the GPU rate (~50 k pairs/s = 100 k frames/s for independent pairs, ~85 k frames/s "
                   "in sequence mode): the loader is the bottleneck by that ratio unless frames arrive decoded")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=4096, help="GLOBAL batch per step, sharded over the ranks (configs[2]: 4096)")
    ap.add_argument("--pairs-per-gpu", type=int, default=0, help="diagnostic: weak scaling, this many pairs on EVERY rank (overrides --pairs)")
    ap.add_argument("--distinct", type=int, default=0,
                    help="diagnostic: generate only this many distinct seeded pairs per rank and tile them (0 = every pair its own seed, as configs[2] says)")
    ap.add_argument("--chunk", type=int, default=32, help="pairs per detect/match launch wave")
    ap.add_argument("--pose-batch", type=int, default=0,
                    help="pairs per pose-stage wave (vo_ctx_create max_pairs; 0 = the context's whole share).  Smaller waves overlap the pose stage "
                         "of one wave with detect + match of the next inside a context: measured 2-7 %% SLOWER than one wave at the end (DESIGN.md 3.2)")
    ap.add_argument("--workers", type=int, default=0, help="host processes for input generation (0 = auto; use 1 under rocprofv3)")
    ap.add_argument("--streams", type=int, default=1,
                    help="contexts per GPU, each driven by one host thread.  One is enough since round 2: a context spreads its chunks over "
                         "up to 4 internal lanes (workspace + stream each) by itself; more contexts are a diagnostic")
    ap.add_argument("--sync-calls", action="store_true", help="diagnostic: one synchronous vo_pairs_run_device call per pass instead of submit / wait")
    ap.add_argument("--skip-mask", type=int, default=0,
                    help="experiment builds only (-DVO_EXPERIMENT_SKIP): after the warm-up, leave out the kernels of this VO_K_* bit mask; "
                         "the results are meaningless, the time tells what the kernel costs end to end")
    ap.add_argument("--join-steps", action="store_true", help="diagnostic: all contexts finish a step before any starts the next one")
    ap.add_argument("--ba", action="store_true",
                    help="extra diagnostic: time the BA residual/Jacobian pass of configs[4] (1000 poses x 200k points x 5 obs, points "
                         "sharded over ranks, pose blocks all-reduced over RCCL)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-extras", action="store_true", help="skip the h2d / png_decode / latency / kitti legs (profiling runs)")
    ap.add_argument("--lib", default=None, help="diagnostic: load this build of libvo_mi355.so instead of the in-tree one (A/B runs of kernel variants)")
    ap.add_argument("--sequence", action="store_true",
                    help="extra diagnostic: also time vo_sequence_run_device (each frame extracted once), one sequence per context")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo to rehearse several ranks on one GPU)")
    ap.add_argument("--config4", action="store_true",
                    help="diagnostic: run configs[3] instead (3840x2160 frames, 20000 ORB features; use e.g. --pairs 64 --chunk 16 --streams 2 --steps 3)")
    ap.add_argument("--dump-records", default=None, help="rank 0 writes the last step's gathered per-pair records (numpy .npy, PAIR_DTYPE) to this path")
    ap.add_argument("--config4-pairs", type=int, default=16, help="pairs of the small configs[3] leg that the default N = 1 run carries (0 = leave it out)")
    ap.add_argument("--kitti-pairs", type=int, default=1024, help="pairs in the configs[0] leg (the bundled KITTI frames, tiled)")
    args = ap.parse_args()
    W_IMG, H_IMG, NFEAT = (3840, 2160, 20000) if args.config4 else (1241, 376, 3000)
    ALGO, P_TOTAL = algo_bytes_per_frame(W_IMG, H_IMG, NFEAT)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    # this rank's contiguous block of the global batch; the seed of a pair is its GLOBAL index (configs[2]: seeds 0..4095)
    from visual_slam_amd import parallel, synth
    if args.pairs_per_gpu > 0:
        G = args.pairs_per_gpu * world
        lo, hi = rank * args.pairs_per_gpu, (rank + 1) * args.pairs_per_gpu
        scaling = "weak"
    else:
        G = args.pairs
        lo, hi = parallel.shard_range(G, rank, world)
        scaling = "strong"
    B = hi - lo
    if B < 1:
        raise SystemExit("fewer pairs than ranks")

    # synthetic input first: the generator forks worker processes, which must happen before this process touches the GPU or RCCL
    workers = args.workers if args.workers > 0 else max(1, min(12, host_threads() // max(min(world, 8), 1)))
    nd = B if args.distinct <= 0 else min(B, args.distinct)
    t_gen = time.perf_counter()
    imgsA, imgsB = synth.make_batch(range(lo, lo + nd), W_IMG, H_IMG, workers=workers)
    t_gen = time.perf_counter() - t_gen
    if nd < B:  # diagnostic: tile the distinct pairs up to the shard size (same work per pair, shorter set-up)
        idx = np.arange(B) % nd
        imgsA, imgsB = imgsA[idx], imgsB[idx]
    # side legs at N = 1 (none of them part of `value`): h2d / png_decode / latency / kitti, and since round 3 also
    # sequence mode, configs[3] (a small 4K leg) and configs[4] (the BA pass, which runs at every N)
    side_legs = rank == 0 and world == 1 and not args.no_extras and not args.config4
    c4A = c4B = None
    if side_legs and args.config4_pairs > 0:
        c4A, c4B = synth.make_batch(range(args.config4_pairs), 3840, 2160, workers=workers)

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU fallback")
    if args.backend != "nccl":
        local_rank %= torch.cuda.device_count()      # rehearsal: several ranks share the visible GPUs
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend)

    from visual_slam_amd import _lib
    if args.lib:
        _lib.SO_PATH = os.path.abspath(args.lib)

    K = synth.config_K(W_IMG, H_IMG)
    dA = torch.from_numpy(imgsA).cuda()
    dB = torch.from_numpy(imgsB).cuda()
    torch.cuda.synchronize()
    S = max(1, min(args.streams, B))
    bounds = [parallel.shard_range(B, i, S) for i in range(S)]
    ctxs = [_lib.Context(max_w=W_IMG, max_h=H_IMG, chunk_pairs=min(args.chunk, b1 - b0), max_pairs=(b1 - b0) if args.pose_batch <= 0 else max(1, min(args.pose_batch, b1 - b0)), device=local_rank, nfeatures=NFEAT) for b0, b1 in bounds]
    ctx = ctxs[0]
    fsz = W_IMG * H_IMG
    pool = None
    if S > 1:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(S)

    def run_part(i):
        b0, b1 = bounds[i]
        return ctxs[i].pairs_run_device(dA.data_ptr() + b0 * fsz, dB.data_ptr() + b0 * fsz, b1 - b0, W_IMG, H_IMG, W_IMG, K)

    def ctx_passes(i, nsteps):
        """Context i makes `nsteps` passes over its share.  Each pass is one asynchronous submission (vo_pairs_submit_device)
        with at most two in flight: the pose stage of pass k (about 3 ms, latency bound) runs while detect + match of pass
        k+1 are already executing, instead of ending every pass with an idle GPU -- what matters once a rank's share of a
        step is small (strong scaling at 8 GPUs: 512 pairs per step).  All passes complete inside the caller's timed
        region (the last wait).  Returns the records pass by pass."""
        b0, b1 = bounds[i]
        if args.sync_calls or args.pose_batch > 0 or nsteps < 1:     # a submission is one pose wave
            return [run_part(i) for _ in range(nsteps)]
        out = []
        for k in range(nsteps):
            ctxs[i].pairs_submit_device(dA.data_ptr() + b0 * fsz, dB.data_ptr() + b0 * fsz, b1 - b0, W_IMG, H_IMG, W_IMG, K)
            if k >= 1:
                out.append(ctxs[i].pairs_wait())
        out.append(ctxs[i].pairs_wait())
        return out

    def run_steps(nsteps):
        # this rank's shard of the job, split over S contexts/streams (ctypes calls release the GIL); then (N > 1) the only
        # collective of the path, once per step: an RCCL all_gather of the 152-byte per-pair result records
        parts = list(pool.map(lambda i: ctx_passes(i, nsteps), range(S))) if pool else [ctx_passes(0, nsteps)]
        last = None
        for k in range(nsteps):
            local = np.concatenate([parts[i][k] for i in range(S)])
            last = parallel.gather_pair_results(local, G) if world > 1 else local
        return last

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    res = run_steps(args.warmup) if args.warmup > 0 else None
    barrier()
    if args.skip_mask:
        os.environ["VO_SKIP_MASK"] = str(args.skip_mask)
    t0 = time.perf_counter()
    if args.join_steps:
        for _ in range(args.steps):
            res = run_steps(1)
    else:
        res = run_steps(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    dt = parallel.max_over_ranks(dt)
    ok = int((res["status"] == 0).sum())

    # per-kernel durations: HIP events around every launch on the launch stream (diagnostic pass,
    # outside the timed region)
    roofline = None
    kernels = {}
    if rank == 0:
        ctx.set_kernel_timing(True)
        r0 = run_part(0)
        kt = ctx.kernel_timing()
        ctx.set_kernel_timing(False)
        nb0 = bounds[0][1] - bounds[0][0]
        frames_per_launch = 2 * min(args.chunk, nb0)
        for name, (ms, n) in kt.items():
            if n:
                kernels[name] = {"ms_total": round(ms, 4), "launches": n, "avg_ms": round(ms / n, 5)}
        # dominant HBM-bound kernel = the one of the byte-streaming kernels with the largest total time
        hbm = {k: v for k, v in kernels.items() if k in ("k_pyramid", "k_fast_tiles", "k_blur")}
        dom = max(hbm, key=lambda k: hbm[k]["ms_total"])
        launches_per_pass = 7 if dom == "k_pyramid" else 1
        bytes_per_launch = ALGO[dom] * frames_per_launch / launches_per_pass
        avg_s = hbm[dom]["avg_ms"] * 1e-3
        achieved = bytes_per_launch / avg_s / 1e9
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": measured_traffic(dom, frames_per_launch, W_IMG, H_IMG),
                    "bytes_per_launch": int(bytes_per_launch), "frames_per_launch": frames_per_launch, "avg_launch_ms": hbm[dom]["avg_ms"],
                    "note": "k_fast_tiles streams every pyramid pixel once but is VALU-issue bound (min/max arc tests): see `valu` and DESIGN.md 3.2"}
        if dom == "k_fast_tiles":
            roofline["valu"] = valu_roofline(dom, frames_per_launch, avg_s, P_TOTAL, W_IMG, H_IMG)
        roofline["per_kernel_GBps"] = {k: round(ALGO[k] * frames_per_launch / (7 if k == "k_pyramid" else 1) / (kernels[k]["avg_ms"] * 1e-3) / 1e9, 1)
                                       for k in ALGO if k in kernels}
        if "k_match" in kernels:
            nk = r0["n_kp1"].astype(np.float64) * r0["n_kp2"].astype(np.float64)
            ops = float(nk.sum()) * 512.0
            a = ops / (kernels["k_match"]["ms_total"] * 1e-3) / 1e12
            roofline["matcher"] = {"bound": "mfma", "kernel": "k_match", "achieved": round(a, 1), "peak": round(MFMA_FP4_PEAK_TOPS, 1),
                                   "unit": "TOP/s (fp4 operands, f32 accumulate; exact)", "frac": round(a / MFMA_FP4_PEAK_TOPS, 4),
                                   "frac_of_i8_peak": round(a / MFMA_I8_PEAK_TOPS, 4), "avg_launch_ms": kernels["k_match"]["avg_ms"],
                                   # the same work in the north-star's terms: 8 xor + 8 popcount lane-ops per distance;
                                   # tools/ubench measures 36.1 T lane-ops/s for the xor+v_bcnt pair on this chip (v_bcnt is half rate)
                                   "equiv_xor_popcount_Tlaneops": round(float(nk.sum()) * 16.0 / (kernels["k_match"]["ms_total"] * 1e-3) / 1e12, 2),
                                   "valu_xor_popcount_ceiling_Tlaneops": 36.1}

    extras = side_legs                                                                 # the side legs are measured at N = 1 only
    h2d = latency = kitti = png = None
    if extras:
        # ---- H2D copy, reported separately (never part of `value`): pinned host memory -> HBM, and the library's host-image
        # entry point (vo_pairs_run: pageable memory, synchronous staging copies + the same kernels)
        nh = min(B, 1024)
        hA = torch.from_numpy(imgsA[:nh]).pin_memory(); hB = torch.from_numpy(imgsB[:nh]).pin_memory()
        tA = torch.empty_like(dA[:nh]); tB = torch.empty_like(dB[:nh])
        for _ in range(2):
            tA.copy_(hA, non_blocking=True); tB.copy_(hB, non_blocking=True)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t1 = time.perf_counter()
            tA.copy_(hA, non_blocking=True); tB.copy_(hB, non_blocking=True)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t1)
        th = float(np.median(ts))
        c1 = _lib.Context(max_w=W_IMG, max_h=H_IMG, chunk_pairs=min(args.chunk, nh), max_pairs=nh, device=local_rank, nfeatures=NFEAT)
        def med3(fn):
            fn()
            ts = []
            for _ in range(3):
                t1 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t1)
            return float(np.median(ts))
        t_host = med3(lambda: c1.pairs_run(imgsA[:nh], imgsB[:nh], K))
        pA, pB = hA.numpy(), hB.numpy()                      # the same call on page-locked arrays: the copies are asynchronous
        t_pin = med3(lambda: c1.pairs_run(pA, pB, K))
        t_dev = med3(lambda: c1.pairs_run_device(dA.data_ptr(), dB.data_ptr(), nh, W_IMG, H_IMG, W_IMG, K))
        h2d = {"ms_per_pair": round(th / nh * 1e3, 4), "bytes_per_pair": 2 * fsz, "GBps": round(2 * fsz * nh / th / 1e9, 2), "pinned": True, "pairs": nh,
               "one_context_pairs_per_sec": {"images_in_hbm": round(nh / t_dev, 1), "host_images_pcie_inclusive": round(nh / t_host, 1),
                                             "pinned_host_images_pcie_inclusive": round(nh / t_pin, 1)},
               "note": "value excludes this copy; *_pcie_inclusive = vo_pairs_run on one context (host arrays; every chunk's copy is enqueued on its lane's "
                       "stream beside the other lanes' kernels): pageable arrays are bounded by the host thread the runtime stages them with, page-locked ones by the GPU"}

        # ---- configs[1]: ONE synthetic pair (seed 0) on one MI355X: latency of the batched entry point at B = 1 and of the
        # drop-in class's call sequence get_matches -> get_pose (vo.py:74-77), images in host memory as the reference has them
        from visual_slam_amd.vo_utils import VisualOdometry
        a0, b0 = imgsA[0], imgsB[0]
        ts = []
        for k in range(25):
            t1 = time.perf_counter()
            c1.pairs_run_device(dA.data_ptr(), dB.data_ptr(), 1, W_IMG, H_IMG, W_IMG, K)
            ts.append(time.perf_counter() - t1)
        vo = VisualOdometry.__new__(VisualOdometry)           # no data_dir: the frames are passed in (frame is not None branch)
        vo.K, vo.D, vo.debug, vo._ctx, vo._ctx_shape = K, np.zeros(5), False, None, None
        tg, tp, nm = [], [], 0
        for k in range(12):
            t1 = time.perf_counter()
            q1, q2 = vo.get_matches(1, "ORB", b0, a0)
            t2 = time.perf_counter()
            vo.get_pose(q1, q2)
            t3 = time.perf_counter()
            tg.append(t2 - t1); tp.append(t3 - t2); nm = len(q1)
        latency = {"workload": "configs[1]: one synthetic 1241x376 pair (seed 0), 3000 ORB features, single MI355X",
                   "pairs_run_device_B1_ms": round(float(np.median(ts[5:])) * 1e3, 3),
                   "dropin_get_matches_ms": round(float(np.median(tg[2:])) * 1e3, 3), "dropin_get_pose_ms": round(float(np.median(tp[2:])) * 1e3, 3),
                   "dropin_total_ms": round(float(np.median(np.array(tg[2:]) + np.array(tp[2:]))) * 1e3, 3), "matches": nm,
                   "note": "medians after warm-up; the drop-in path uploads both frames per call (vo_get_matches) and reads only q1 / q2 back; get_pose uploads them again"}
        c1.close()

        # ---- configs[0]: the bundled KITTI frames (tests/golden/kitti02, the reference's assets/test_imgs/02/000000-000003.png),
        # consecutive pairs (0,1) (1,2) (2,3) tiled to --kitti-pairs, same contexts and call pattern as the headline
        try:
            from visual_slam_amd import loader
            kp = loader.load_image_paths(KITTI_DIR)
            kimgs = np.stack([loader.decode_gray(p) for p in kp])
            if kimgs.shape[1:] == (H_IMG, W_IMG) and len(kimgs) >= 2:
                KB = min(args.kitti_pairs, B)
                ia = np.arange(KB) % (len(kimgs) - 1)
                kA = torch.from_numpy(kimgs[ia]).cuda(); kB_ = torch.from_numpy(kimgs[ia + 1]).cuda()
                torch.cuda.synchronize()
                kb = [parallel.shard_range(KB, i, S) for i in range(S)]
                Kk = VisualOdometry.load_calib(os.path.join(KITTI_DIR, "calib.txt"))[0]

                def kitti_part(i, nsteps):
                    b0_, b1_ = kb[i]
                    r = None
                    for _ in range(nsteps):
                        r = ctxs[i].pairs_run_device(kA.data_ptr() + b0_ * fsz, kB_.data_ptr() + b0_ * fsz, b1_ - b0_, W_IMG, H_IMG, W_IMG, Kk)
                    return r

                def kitti_run(nsteps):
                    parts = list(pool.map(lambda i: kitti_part(i, nsteps), range(S))) if pool else [kitti_part(0, nsteps)]
                    return np.concatenate(parts)

                kitti_run(1)
                torch.cuda.synchronize()
                ksteps = max(3, min(args.steps, 10))
                t1 = time.perf_counter()
                kr = kitti_run(ksteps)
                torch.cuda.synchronize()
                kdt = time.perf_counter() - t1
                ctx.set_kernel_timing(True)
                kitti_part(0, 1)
                kkt = ctx.kernel_timing()
                ctx.set_kernel_timing(False)
                kf = 2 * min(args.chunk, kb[0][1] - kb[0][0])
                kitti = {"workload": "configs[0]: bundled KITTI seq 02 frames 000000-000003 (1241x376), consecutive pairs tiled to %d, K from calib.txt" % KB,
                         "frame_pairs_per_sec": round(KB * ksteps / kdt, 1), "steps": ksteps, "pairs_ok": int((kr["status"] == 0).sum()), "pairs_total": int(len(kr)),
                         "mean_matches": round(float(kr["n_match"].mean()), 1), "mean_inliers": round(float(kr["n_inlier"].mean()), 1),
                         "k_fast_tiles_avg_ms": round(kkt["k_fast_tiles"][0] / max(kkt["k_fast_tiles"][1], 1), 5),
                         "k_fast_tiles_GBps": round(ALGO["k_fast_tiles"] * kf / (kkt["k_fast_tiles"][0] / max(kkt["k_fast_tiles"][1], 1) * 1e-3) / 1e9, 1),
                         "ransac_stage_ms": round(kkt["ransac(stage)"][0], 3), "data": "real (6 fixture frames ship with the repo; the reference's other 773 do not travel)"}
                del kA, kB_
        except OSError:
            kitti = None
        png = png_decode_leg()
        if png is not None:
            # end to end: PNG files -> decoder processes -> pinned ring -> HBM (side stream) -> vo_sequence_run_device, chunk by chunk
            try:
                kp4 = loader.load_image_paths(KITTI_DIR)
                bounce = (kp4 + kp4[-2:0:-1]) * 172                       # 0 1 2 3 2 1 0 1 ...: every consecutive pair is a real one
                c2 = _lib.Context(max_w=W_IMG, max_h=H_IMG, chunk_pairs=32, max_pairs=128, device=local_rank, nfeatures=NFEAT)
                Kk = VisualOdometry.load_calib(os.path.join(KITTI_DIR, "calib.txt"))[0]
                it = iter(loader.FrameLoader(bounce, chunk=129, overlap=1, processes=host_threads(), device=local_rank))
                first = next(it)
                c2.sequence_run_device(first.data_ptr(), first.shape[0], W_IMG, H_IMG, W_IMG, Kk)
                t1 = time.perf_counter()
                npairs = nok = 0
                for fr in it:
                    r = c2.sequence_run_device(fr.data_ptr(), fr.shape[0], W_IMG, H_IMG, W_IMG, Kk)
                    npairs += len(r); nok += int((r["status"] == 0).sum())
                png["end_to_end_png_to_pose"] = {"frame_pairs_per_sec": round(npairs / (time.perf_counter() - t1), 1), "pairs": npairs, "pairs_ok": nok,
                                                 "decoder_processes": host_threads(), "note": "files -> FrameLoader(processes) -> pinned ring -> HBM -> vo_sequence_run_device; decode bound"}
                # the same pipeline fed from the decoded-once format (loader.pack_sequence -> one memory-mapped uint8 [F,h,w] file)
                import tempfile
                with tempfile.TemporaryDirectory() as td:
                    arr = np.load(loader.pack_sequence(bounce, os.path.join(td, "seq.npy")), mmap_mode="r")
                    c2.close()
                    c2 = _lib.Context(max_w=W_IMG, max_h=H_IMG, chunk_pairs=32, max_pairs=512, device=local_rank, nfeatures=NFEAT)
                    arr = np.concatenate([arr] * 4)          # 4128 frames in memory (file cache stand-in), waves of 512 pairs
                    it = iter(loader.FrameLoader(arr, chunk=513, overlap=1, workers=min(8, host_threads()), device=local_rank))
                    first = next(it)
                    c2.sequence_run_device(first.data_ptr(), first.shape[0], W_IMG, H_IMG, W_IMG, Kk)
                    t1 = time.perf_counter()
                    npairs = nok = 0
                    held = []                                  # frames of the submissions in flight stay referenced until their wait
                    for fr in it:
                        c2.sequence_submit_device(fr.data_ptr(), fr.shape[0], W_IMG, H_IMG, W_IMG, Kk)
                        held.append(fr)
                        if len(held) == 2:
                            r = c2.pairs_wait(); held.pop(0)
                            npairs += len(r); nok += int((r["status"] == 0).sum())
                    while held:
                        r = c2.pairs_wait(); held.pop(0)
                        npairs += len(r); nok += int((r["status"] == 0).sum())
                    png["end_to_end_packed_to_pose"] = {"frame_pairs_per_sec": round(npairs / (time.perf_counter() - t1), 1), "pairs": npairs, "pairs_ok": nok,
                                                        "note": "decoded frames (loader.pack_sequence format) -> pinned ring (8 copy threads) -> HBM -> vo_sequence_submit_device / vo_pairs_wait in waves of 512 pairs, two in flight, one host thread"}
                    del arr, it
                c2.close()
            except OSError:
                pass

    seq = None
    if rank == 0 and (args.sequence or extras):
        # same kernels on S sequences, one per context: A_k, B_k, A_k, B_k, ... of ONE pair k each, so that every consecutive
        # pair is a pair of related views (forward or reverse motion) and none dies at the ratio test; each frame is
        # extracted once per chunk
        sb = [parallel.shard_range(B, i, S) for i in range(S)]
        seqs = []
        for i, (b0, b1) in enumerate(sb):
            Fi = b1 - b0 + 1
            k = i % nd
            par = (torch.arange(Fi, device=dA.device) % 2 == 0)[:, None, None]
            seqs.append(torch.where(par, dA[k][None], dB[k][None]).contiguous())
        torch.cuda.synchronize()

        def seq_part(i, nsteps):
            r = None
            if args.sync_calls or args.pose_batch > 0:
                for _ in range(nsteps):
                    r = ctxs[i].sequence_run_device(seqs[i].data_ptr(), seqs[i].shape[0], W_IMG, H_IMG, W_IMG, K)
                return r
            for k in range(nsteps):                       # submit / wait, two passes in flight
                ctxs[i].sequence_submit_device(seqs[i].data_ptr(), seqs[i].shape[0], W_IMG, H_IMG, W_IMG, K)
                if k >= 1:
                    r = ctxs[i].pairs_wait()
            return ctxs[i].pairs_wait()

        def seq_run(nsteps):
            parts = list(pool.map(lambda i: seq_part(i, nsteps), range(S))) if pool else [seq_part(0, nsteps)]
            return np.concatenate(parts)

        seq_run(1)
        torch.cuda.synchronize()
        ssteps = max(3, min(args.steps, 10))
        t1 = time.perf_counter()
        r2 = seq_run(ssteps)
        torch.cuda.synchronize()
        seq = {"frame_pairs_per_sec": round(len(r2) * ssteps / (time.perf_counter() - t1), 2), "pairs_ok": int((r2["status"] == 0).sum()),
               "pairs_total": int(len(r2)), "note": "consecutive-pair mode (SURVEY 8(f) N3): one sequence per context, each frame extracted once per chunk"}
        del seqs

    config4 = None
    if extras and c4A is not None:
        # ---- configs[3]: 3840x2160 frames, 20000 ORB features, K scaled: a small leg (16 pairs by default, chunks of 8) so that the
        # driver's record carries the O(N^2) matcher case; `--config4 --pairs 64 --chunk 16` is the full-size diagnostic
        W4, H4, NF4 = 3840, 2160, 20000
        A4, P4 = algo_bytes_per_frame(W4, H4, NF4)
        K4 = synth.config_K(W4, H4)
        n4 = len(c4A)
        d4A = torch.from_numpy(c4A).cuda(); d4B = torch.from_numpy(c4B).cuda()
        c4 = _lib.Context(max_w=W4, max_h=H4, chunk_pairs=min(8, n4), max_pairs=n4, device=local_rank, nfeatures=NF4)
        c4.pairs_run_device(d4A.data_ptr(), d4B.data_ptr(), n4, W4, H4, W4, K4)
        torch.cuda.synchronize()
        steps4 = 3
        t1 = time.perf_counter()
        for k in range(steps4):
            c4.pairs_submit_device(d4A.data_ptr(), d4B.data_ptr(), n4, W4, H4, W4, K4)
            if k >= 1:
                r4 = c4.pairs_wait()
        r4 = c4.pairs_wait()
        torch.cuda.synchronize()
        dt4 = time.perf_counter() - t1
        c4.set_kernel_timing(True)
        r4t = c4.pairs_run_device(d4A.data_ptr(), d4B.data_ptr(), n4, W4, H4, W4, K4)
        kt4 = c4.kernel_timing()
        c4.set_kernel_timing(False)
        c4.close()
        fpl4 = 2 * min(8, n4)
        k4 = {name: {"ms_total": round(ms, 4), "launches": n, "avg_ms": round(ms / n, 5)} for name, (ms, n) in kt4.items() if n}
        config4 = {"workload": "configs[3]: %d synthetic 3840x2160 pairs (seeds 0..%d), 20000 ORB features, K scaled x3.094, chunks of %d pairs" % (n4, n4 - 1, min(8, n4)),
                   "frame_pairs_per_sec": round(n4 * steps4 / dt4, 2), "steps": steps4, "pairs_ok": int((r4["status"] == 0).sum()), "pairs_total": int(len(r4)),
                   "mean_keypoints": round(float(r4["n_kp1"].mean()), 1), "mean_matches": round(float(r4["n_match"].mean()), 1), "mean_inliers": round(float(r4["n_inlier"].mean()), 1),
                   "per_kernel_GBps": {k: round(A4[k] * fpl4 / (7 if k == "k_pyramid" else 1) / (k4[k]["avg_ms"] * 1e-3) / 1e9, 1) for k in A4 if k in k4},
                   "kernels": k4}
        if "k_match" in k4:
            ops4 = float((r4t["n_kp1"].astype(np.float64) * r4t["n_kp2"].astype(np.float64)).sum()) * 512.0
            a4 = ops4 / (k4["k_match"]["ms_total"] * 1e-3) / 1e12
            config4["matcher"] = {"bound": "mfma", "achieved": round(a4, 1), "peak": round(MFMA_FP4_PEAK_TOPS, 1), "unit": "TOP/s (fp4 operands, f32 accumulate; exact)",
                                  "frac": round(a4 / MFMA_FP4_PEAK_TOPS, 4), "frac_of_i8_peak": round(a4 / MFMA_I8_PEAK_TOPS, 4)}
        del d4A, d4B

    ba_out = None
    if args.ba or (not args.no_extras and not args.config4):
        # ---- configs[4]: the BA residual / Jacobian pass; points (and their observations) sharded over the ranks, the pose
        # blocks all-reduced (RCCL at N > 1).  Runs at every N; never part of `value`.
        from visual_slam_amd import ba as BA
        pr = BA.synth_problem(1000, 200000, 5, seed=0)
        sh = BA.shard_observations(pr, rank, world)
        plan = BA.BAPlan(sh["obs_pose"], sh["obs_point"], 1000, 200000, device=local_rank)
        M = len(sh["obs_pose"])
        dev = torch.device("cuda", local_rank)
        t_p = torch.from_numpy(sh["poses"]).to(dev); t_x = torch.from_numpy(sh["points"]).to(dev); t_uv = torch.from_numpy(sh["uv"]).to(dev)
        o_r = torch.empty((M, 2), dtype=torch.float64, device=dev); o_n = torch.empty(M, dtype=torch.float64, device=dev)
        o_Hpp = torch.empty((1000, 6, 6), dtype=torch.float64, device=dev); o_bp = torch.empty((1000, 6), dtype=torch.float64, device=dev)
        o_Hll = torch.empty((200000, 3, 3), dtype=torch.float64, device=dev); o_bl = torch.empty((200000, 3), dtype=torch.float64, device=dev)
        o_Hpl = torch.empty((M, 6, 3), dtype=torch.float64, device=dev)

        def ba_step():
            plan.run_device(t_p.data_ptr(), t_x.data_ptr(), t_uv.data_ptr(), pr["K"], o_r.data_ptr(), o_n.data_ptr(), o_Hpp.data_ptr(),
                            o_bp.data_ptr(), o_Hll.data_ptr(), o_bl.data_ptr(), o_Hpl.data_ptr())
            return BA.allreduce_pose_blocks(o_Hpp, o_bp)

        for _ in range(3):
            ba_step()
        barrier()
        t1 = time.perf_counter()
        for _ in range(20):
            ba_step()
        barrier()
        ba_dt = parallel.max_over_ranks((time.perf_counter() - t1) / 20)
        # algorithmic HBM bytes of one pass over this rank's M observations: read uv 16 + the two indices 8 per
        # observation and every point once (24 B); write r 16 + norm 8 + Hpl 144 per observation, Hll 72 + bl 24 per point,
        # Hpp + bp 336 per pose (the replicated pose table is 48 KB: left out)
        npts = int(len(np.unique(sh["obs_point"])))
        ba_bytes = M * (16 + 8 + 16 + 8 + 144) + npts * (24 + 72 + 24) + 1000 * 336
        ba_out = {"workload": "configs[4]: 1000 poses x 200000 points x 5 observations, fp64, residuals + Hpp/bp/Hll/bl/Hpl", "ms_per_pass": round(ba_dt * 1e3, 4),
                  "observations_per_sec": round(1e6 / ba_dt, 1), "rms_residual_px": round(float(torch.sqrt((o_n ** 2).mean())), 4),
                  "collective": "all_reduce of [1000,42] f64 pose blocks" if world > 1 else "none (1 rank)"}
        if rank == 0:
            # the kernels one at a time, HIP events on the stream they are launched on (a torch side stream handed to the library)
            side = torch.cuda.Stream(device=dev)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            kb = {}
            with torch.cuda.stream(side):
                for _ in range(2):
                    plan.run_device(t_p.data_ptr(), t_x.data_ptr(), t_uv.data_ptr(), pr["K"], o_r.data_ptr(), o_n.data_ptr(), o_Hpp.data_ptr(),
                                    o_bp.data_ptr(), o_Hll.data_ptr(), o_bl.data_ptr(), o_Hpl.data_ptr(), stream=side.cuda_stream)
                side.synchronize()
                batches = []
                for _b in range(7):        # 7 batches of 20 passes back to back; the median batch (a pass is two launches: a slow host moment shows as device time)
                    e0.record(side)
                    for _ in range(20):
                        plan.run_device(t_p.data_ptr(), t_x.data_ptr(), t_uv.data_ptr(), pr["K"], o_r.data_ptr(), o_n.data_ptr(), o_Hpp.data_ptr(),
                                        o_bp.data_ptr(), o_Hll.data_ptr(), o_bl.data_ptr(), o_Hpl.data_ptr(), stream=side.cuda_stream)
                    e1.record(side)
                    side.synchronize()
                    batches.append(e0.elapsed_time(e1) / 20)
                kb["whole_pass"] = float(np.median(batches))
                kb["batches"] = [round(b, 5) for b in batches]
            # the pass kernel alone, HIP events around its launch on the launch stream (as `roofline` does for the detect kernels):
            # a pass is two dependent launches, and the gap between them -- 1 to 6 us per boundary from box to box -- is not kernel time
            plan.set_kernel_timing(True)
            with torch.cuda.stream(side):
                for _ in range(20):
                    plan.run_device(t_p.data_ptr(), t_x.data_ptr(), t_uv.data_ptr(), pr["K"], o_r.data_ptr(), o_n.data_ptr(), o_Hpp.data_ptr(),
                                    o_bp.data_ptr(), o_Hll.data_ptr(), o_bl.data_ptr(), o_Hpl.data_ptr(), stream=side.cuda_stream)
            kt_ba = plan.kernel_timing()
            plan.set_kernel_timing(False)
            kb["k_ba_pass"] = kt_ba["k_ba_pass"][0] / max(kt_ba["k_ba_pass"][1], 1)
            kb["k_ba_rot"] = kt_ba["k_ba_rot"][0] / max(kt_ba["k_ba_rot"][1], 1)
            a_ba = ba_bytes / (kb["k_ba_pass"] * 1e-3) / 1e9
            ba_traffic = None
            try:        # the committed FETCH_SIZE / WRITE_SIZE passes describe the whole configs[4] problem on one rank
                with open(BA_TRAFFIC_FILE) as f:
                    bt = json.load(f)
                if world == 1 and bt["meta"]["observations"] == M:
                    ba_traffic = int(sum(2.0 * k_["FETCH_SIZE_KB_avg_per_launch"] + k_["WRITE_SIZE_KB_avg_per_launch"] for k_ in bt["kernels"].values()) * 1024)
            except Exception:
                pass
            ba_out["roofline"] = {"bound": "hbm", "achieved": round(a_ba, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(a_ba / HBM_PEAK_GBS, 4), "traffic": ba_traffic,
                                  "bytes_per_pass": int(ba_bytes), "kernel": "k_ba_pass", "avg_launch_ms": round(kb["k_ba_pass"], 5), "k_ba_rot_avg_launch_ms": round(kb["k_ba_rot"], 5),
                                  "device_ms_per_pass": round(kb["whole_pass"], 5), "device_ms_per_pass_batches": kb["batches"],
                                  "frac_of_whole_pass": round(ba_bytes / (kb["whole_pass"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                  "note": "achieved / frac: the algorithmic bytes of a pass over the average duration of k_ba_pass, HIP events around its launch (20 launches); device_ms_per_pass: median of 7 batches of 20 passes back to back, which adds k_ba_rot and the two launch boundaries (k_ba_rot + k_ba_pass per pass; kernel durations: profiles/r3_*_ba_kernel_stats.csv); ms_per_pass above is host wall clock per synchronous call (+ the all-reduce at N > 1)"}
        plan.close()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:      # on rank 0 at N = 1 only
        cpu = cpu_baseline(imgsA, imgsB, K, NFEAT)

    if rank == 0 and args.dump_records:
        np.save(args.dump_records, res)
    if rank == 0:
        value = G * args.steps / dt
        out = {
            "metric": "frame-pairs/sec (detect+match+pose) at 1241x376/3k ORB",
            "value": round(value, 2), "unit": "frame-pairs/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "u8/int32 detect+match, f64 pose", "data": "synthetic",
            "config": {"workload": ("configs[2]: batch of 4096 synthetic 1241x376 pairs (configs[1] generator, seed = pair index), 3000 ORB features, 8 levels" if not args.config4 else
                                    "configs[3] (diagnostic): synthetic 3840x2160 pairs, 20000 ORB features, 8 levels"),
                       "global_pairs": G, "pairs_on_rank0": B, "distinct_pairs_on_rank0": nd, "chunk_pairs": min(args.chunk, B), "pose_batch_pairs": B if args.pose_batch <= 0 else min(args.pose_batch, B), "lanes_per_context": int(os.environ.get("VO_LANES", "6")), "streams_per_gpu": S,
                       "contexts_join_every_step": bool(args.join_steps), "calls": "synchronous" if (args.sync_calls or args.pose_batch > 0) else "submit/wait, 2 in flight", "parallelism": "pairs sharded contiguously over ranks, no data-path collective; all_gather of 152-byte records per step",
                       "pairs_ok": ok, "pairs_total": int(len(res)), "timed_region_s": round(dt, 3), "input_generation_s": round(t_gen, 1)},
            "roofline": roofline, "cpu_baseline": cpu, "h2d": h2d, "png_decode": png, "latency": latency, "kitti": kitti,
            "sequence_mode": seq, "config4": config4, "ba_pass": ba_out, "kernels": kernels,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
