#!/usr/bin/env python3
"""bench.py -- frame-pairs/sec (detect + match + pose) at 1241x376 / 3000 ORB features.

One "step" = one pass of the hot path (vo_pairs_run_device: ORB detect x2, Hamming 2-NN +
ratio, five-point RANSAC, recoverPose) over a batch of B synthetic frame pairs per GPU
(BASELINE.json configs[2]: a batch of configs[1]-style pairs, seeds rank*B .. rank*B+B-1),
already resident in HBM when the timed region starts.  Pairs are independent, so ranks
shard them with no data-path collective ("weak" scaling: B per GPU is fixed).

Contract: python bench.py --gpus N --steps K --warmup W   (N>1 under torch.distributed.run)
prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W_IMG, H_IMG, NFEAT = 1241, 376, 3000
# algorithmic HBM bytes per frame; per-launch figures are these x frames per launch
PYR_PX = [1241 * 376, 1034 * 313, 862 * 261, 718 * 218, 598 * 181, 499 * 151, 416 * 126, 346 * 105]
P_TOTAL = sum(PYR_PX)  # 1,444,097
ALGO_BYTES_PER_FRAME = {
    "k_pyramid": P_TOTAL,                      # read level 0 + write levels 1..7 (all 7 launches together)
    "k_fast_tiles": P_TOTAL,                   # read every pyramid pixel once
    "k_blur": 2 * P_TOTAL,                     # read + write every pyramid pixel
    "k_rbrief": NFEAT * 512 + NFEAT * 32,
    "k_orient": NFEAT * 749,
    "k_harris": 2 * NFEAT * 81,
}
HBM_PEAK_GBS = 8000.0
# matcher: one Hamming distance = a 256-long i8 dot product on the matrix cores = 512 ops;
# dense i8 MFMA peak = 2 x bf16 = 256 CU x 4 SIMD x 2048 ops/clk x 2.4 GHz (MI355X_MICROARCH.md, Matrix cores)
MFMA_I8_PEAK_TOPS = 256 * 4 * 2048 * 2.4e9 / 1e12
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r1_f_pmc_hbm_traffic.json")


def measured_traffic(kernel, frames_per_launch):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE + WRITE_SIZE, KB, separate
    runs of this script; profiles/r1_f_pmc_hbm_traffic.json), scaled to this run's frames per launch.
    None when the file or the kernel is missing: bench.py itself never runs the profiler."""
    try:
        with open(TRAFFIC_FILE) as f:
            t = json.load(f)
        k = t["kernels"][kernel]
        # gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies 128-B requests at 64 B, so it is doubled;
        # k_blur confirms it for these dword-wide reads (raw 93.7 MB fetched for 184.8 MB that must be read).
        kb = 2.0 * k["FETCH_SIZE_KB_avg_per_launch"] + k["WRITE_SIZE_KB_avg_per_launch"]
        return int(kb * 1024 * frames_per_launch / t["meta"]["frames_per_launch"])
    except Exception:
        return None


def cpu_baseline(imgsA, imgsB, K, budget_s=12.0):
    """The CPU oracle (a port of the cv2 path, oracle/) timed on this host, one core,
    on a bounded sample of the same pairs."""
    from oracle import oracle as O
    O.build()
    n = 0
    t0 = time.perf_counter()
    while n < len(imgsA) and (time.perf_counter() - t0 < budget_s or n < 2):
        rc, _, _ = O.pair_run(imgsA[n], imgsB[n], K)
        n += 1
    dt = time.perf_counter() - t0
    out = {"value": n / dt, "unit": "frame-pairs/sec", "cores": 1, "kind": "port",
           "sample": "first %d pairs of rank 0's batch, scalar C++ oracle (restated cv2 ORB+BFMatcher+findEssentialMat+recoverPose), 1 thread" % n}
    try:      # SURVEY.md 8(d) "live cv2": never installed or fetched; timed beside the port only where a box already has it
        import cv2
    except ImportError:
        out["cv2"] = None
        return out
    cv2.setNumThreads(1)
    orb, bf = cv2.ORB_create(NFEAT), cv2.BFMatcher(cv2.NORM_HAMMING)
    m, t0 = 0, time.perf_counter()
    while m < len(imgsA) and (time.perf_counter() - t0 < budget_s or m < 2):
        k1, f1 = orb.detectAndCompute(imgsA[m], None)
        k2, f2 = orb.detectAndCompute(imgsB[m], None)
        good = [a for a, b in bf.knnMatch(f1, f2, k=2) if a.distance < 0.45 * b.distance]
        q1 = np.float32([k1[g.queryIdx].pt for g in good]); q2 = np.float32([k2[g.trainIdx].pt for g in good])
        E, mask = cv2.findEssentialMat(q1, q2, K, cv2.RANSAC, 0.999, 1.0, None)
        cv2.recoverPose(E, q1, q2, K, mask)
        m += 1
    out["cv2"] = {"value": m / (time.perf_counter() - t0), "unit": "frame-pairs/sec", "cores": 1, "version": cv2.__version__,
                  "sample": "first %d pairs, cv2 ORB + BFMatcher + findEssentialMat + recoverPose, setNumThreads(1)" % m}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=2048, help="frame pairs per GPU per step")
    ap.add_argument("--distinct", type=int, default=256, help="distinct seeded pairs generated per GPU (tiled up to --pairs)")
    ap.add_argument("--chunk", type=int, default=32, help="pairs per detect/match launch wave")
    ap.add_argument("--workers", type=int, default=0, help="host processes for input generation (0 = auto; use 1 under rocprofv3)")
    ap.add_argument("--streams", type=int, default=4,
                    help="contexts (one HIP stream and workspace each) driven by one host thread each; the batch is split between them")
    ap.add_argument("--join-steps", action="store_true", help="diagnostic: all contexts finish a step before any starts the next one")
    ap.add_argument("--ba", action="store_true",
                    help="extra diagnostic: time the BA residual/Jacobian pass of configs[4] (1000 poses x 200k points x 5 obs, points "
                         "sharded over ranks, pose blocks all-reduced over RCCL)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--lib", default=None, help="diagnostic: load this build of libvo_mi355.so instead of the in-tree one (A/B runs of kernel variants)")
    ap.add_argument("--sequence", action="store_true",
                    help="extra diagnostic: also time vo_sequence_run_device on the A frames taken as one sequence (each frame extracted once)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo to rehearse several ranks on one GPU)")
    ap.add_argument("--config4", action="store_true",
                    help="diagnostic: run configs[3] instead (3840x2160 frames, 20000 ORB features; use e.g. --pairs 64 --distinct 4 --chunk 16 --streams 2)")
    args = ap.parse_args()
    global W_IMG, H_IMG, NFEAT
    if args.config4:
        W_IMG, H_IMG, NFEAT = 3840, 2160, 20000

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    # synthetic input first: the generator forks worker processes, which must happen before this process touches the GPU or RCCL
    from visual_slam_amd import synth
    B = args.pairs
    workers = args.workers if args.workers > 0 else max(1, min(12, (os.cpu_count() or 1) // max(world, 1)))
    nd = min(B, args.distinct)
    imgsA, imgsB = synth.make_batch(range(rank * nd, rank * nd + nd), W_IMG, H_IMG, workers=workers)
    if nd < B:  # tile the distinct pairs up to the batch size (same work per pair, bounded host set-up time)
        reps = (B + nd - 1) // nd
        imgsA = np.concatenate([imgsA] * reps)[:B]
        imgsB = np.concatenate([imgsB] * reps)[:B]

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

    from visual_slam_amd import _lib, parallel, synth
    if args.lib:
        _lib.SO_PATH = os.path.abspath(args.lib)

    K = synth.config_K(W_IMG, H_IMG)
    dA = torch.from_numpy(imgsA).cuda()
    dB = torch.from_numpy(imgsB).cuda()
    S = max(1, min(args.streams, B))
    bounds = [parallel.shard_range(B, i, S) for i in range(S)]
    ctxs = [_lib.Context(max_w=W_IMG, max_h=H_IMG, chunk_pairs=min(args.chunk, hi - lo), max_pairs=hi - lo, device=local_rank, nfeatures=NFEAT) for lo, hi in bounds]
    ctx = ctxs[0]
    fsz = W_IMG * H_IMG
    pool = None
    if S > 1:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(S)

    def run_part(i):
        lo, hi = bounds[i]
        return ctxs[i].pairs_run_device(dA.data_ptr() + lo * fsz, dB.data_ptr() + lo * fsz, hi - lo, W_IMG, H_IMG, W_IMG, K)

    def ctx_passes(i, nsteps):
        """Context i makes `nsteps` passes over its share as back-to-back calls: its host thread does not wait for the other
        contexts between passes, so one context's latency-bound pose stage (end of a call) overlaps the detect/match chunks
        of the others instead of all contexts reaching it together (measured +7 % over joining after every step; shifting the
        contexts half a pass out of phase on purpose measured no better).  Returns the records pass by pass."""
        lo, hi = bounds[i]
        return [ctxs[i].pairs_run_device(dA.data_ptr() + lo * fsz, dB.data_ptr() + lo * fsz, hi - lo, W_IMG, H_IMG, W_IMG, K) for _ in range(nsteps)]

    def run_steps(nsteps):
        # this rank's shard of the job, split over S contexts/streams (ctypes calls release the GIL); then (N > 1) the only
        # collective of the path, once per step: an RCCL all_gather of the 152-byte per-pair result records
        parts = list(pool.map(lambda i: ctx_passes(i, nsteps), range(S))) if pool else [ctx_passes(0, nsteps)]
        last = None
        for k in range(nsteps):
            local = np.concatenate([parts[i][k] for i in range(S)])
            last = parallel.gather_pair_results(local, world * B) if world > 1 else local
        return last

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    res = run_steps(args.warmup)
    barrier()
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
        run_part(0)
        kt = ctx.kernel_timing()
        ctx.set_kernel_timing(False)
        frames_per_launch = 2 * min(args.chunk, bounds[0][1] - bounds[0][0])
        for name, (ms, n) in kt.items():
            if n:
                kernels[name] = {"ms_total": round(ms, 4), "launches": n, "avg_ms": round(ms / n, 5)}
        # dominant HBM-bound kernel = the one of the byte-streaming kernels with the largest total time
        hbm = {k: v for k, v in kernels.items() if k in ("k_pyramid", "k_fast_tiles", "k_blur")}
        dom = max(hbm, key=lambda k: hbm[k]["ms_total"])
        launches_per_pass = 7 if dom == "k_pyramid" else 1
        bytes_per_launch = ALGO_BYTES_PER_FRAME[dom] * frames_per_launch / launches_per_pass
        avg_s = hbm[dom]["avg_ms"] * 1e-3
        achieved = bytes_per_launch / avg_s / 1e9
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": measured_traffic(dom, frames_per_launch),
                    "bytes_per_launch": int(bytes_per_launch), "avg_launch_ms": hbm[dom]["avg_ms"],
                    "note": "k_fast_tiles streams every pyramid pixel once but is VALU-issue bound (min/max arc tests), see DESIGN.md"}
        if "k_match" in kernels:
            nb0 = bounds[0][1] - bounds[0][0]
            nk = res["n_kp1"][:nb0].astype(np.float64) * res["n_kp2"][:nb0].astype(np.float64)
            ops = float(nk.sum()) * 512.0
            a = ops / (kernels["k_match"]["ms_total"] * 1e-3) / 1e12
            roofline["matcher"] = {"bound": "mfma", "kernel": "k_match", "achieved": round(a, 1), "peak": round(MFMA_I8_PEAK_TOPS, 1),
                                   "unit": "TOP/s (i8)", "frac": round(a / MFMA_I8_PEAK_TOPS, 4), "avg_launch_ms": kernels["k_match"]["avg_ms"],
                                   # the same work in the north-star's terms: 8 xor + 8 popcount lane-ops per distance;
                                   # tools/ubench measures 36.1 T lane-ops/s for the xor+v_bcnt pair on this chip (v_bcnt is half rate)
                                   "equiv_xor_popcount_Tlaneops": round(float(nk.sum()) * 16.0 / (kernels["k_match"]["ms_total"] * 1e-3) / 1e12, 2),
                                   "valu_xor_popcount_ceiling_Tlaneops": 36.1}

    seq = None
    if rank == 0 and args.sequence:
        # same kernels on one B-frame sequence A_k,B_k,A_k,B_k,... (blocks of 16 frames per distinct pair), so that
        # consecutive frames are related views; every frame is extracted once per chunk
        f = np.arange(B)
        k = (f // 16) % nd
        dS = torch.where(torch.from_numpy((f % 2 == 0)).cuda()[:, None, None], dA[torch.from_numpy(k).cuda()], dB[torch.from_numpy(k).cuda()]).contiguous()
        # the B-1 pairs are split over the S contexts like the pair batches; neighbouring blocks share one frame
        sb = [parallel.shard_range(B - 1, i, S) for i in range(S)]

        def seq_part(i, nsteps):
            lo, hi = sb[i]
            r = None
            for _ in range(nsteps):
                r = ctxs[i].sequence_run_device(dS.data_ptr() + lo * fsz, hi - lo + 1, W_IMG, H_IMG, W_IMG, K)
            return r

        def seq_run(nsteps):
            parts = list(pool.map(lambda i: seq_part(i, nsteps), range(S))) if pool else [seq_part(0, nsteps)]
            return np.concatenate(parts)

        seq_run(1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        r2 = seq_run(args.steps)
        torch.cuda.synchronize()
        seq = {"frame_pairs_per_sec": round((B - 1) * args.steps / (time.perf_counter() - t1), 2), "pairs_ok": int((r2["status"] == 0).sum()),
               "pairs_total": B - 1, "note": "consecutive-pair mode (SURVEY 8(f) N3): B frames, B-1 pairs, each frame extracted once per chunk"}

    ba_out = None
    if args.ba:
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
        ba_out = {"workload": "configs[4]: 1000 poses x 200000 points x 5 observations, fp64, residuals + Hpp/bp/Hll/bl/Hpl", "ms_per_pass": round(ba_dt * 1e3, 4),
                  "observations_per_sec": round(1e6 / ba_dt, 1), "rms_residual_px": round(float(torch.sqrt((o_n ** 2).mean())), 4),
                  "collective": "all_reduce of [1000,42] f64 pose blocks" if world > 1 else "none (1 rank)"}
        plan.close()

    cpu = None
    if rank == 0 and not args.no_cpu:
        cpu = cpu_baseline(imgsA, imgsB, K)

    if rank == 0:
        value = world * B * args.steps / dt
        out = {
            "metric": "frame-pairs/sec (detect+match+pose) at 1241x376/3k ORB",
            "value": round(value, 2), "unit": "frame-pairs/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/int32 detect+match, f64 pose", "data": "synthetic",
            "config": {"workload": ("configs[2]: batch of synthetic 1241x376 pairs (configs[1] generator, seeds 0..), 3000 ORB features, 8 levels" if not args.config4 else
                                    "configs[3] (diagnostic): synthetic 3840x2160 pairs, 20000 ORB features, 8 levels"),
                       "pairs_per_gpu": B, "distinct_pairs_per_gpu": nd, "global_pairs": world * B, "chunk_pairs": min(args.chunk, B), "streams_per_gpu": S, "contexts_join_every_step": bool(args.join_steps), "parallelism": "pairs sharded over ranks, no collective",
                       "pairs_ok": ok, "pairs_total": int(len(res))},
            "roofline": roofline, "cpu_baseline": cpu, "sequence_mode": seq, "ba_pass": ba_out, "kernels": kernels,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
