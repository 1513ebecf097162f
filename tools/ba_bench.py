"""BA residual / Jacobian pass of BASELINE.json configs[4] (1000 poses x 200000 points x 5 observations) run N times on
device buffers: the program rocprofv3 wraps for the k_ba_* kernel statistics and PMC passes (tools/profile_r3.sh).
python tools/ba_bench.py [passes]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    mode = sys.argv[2] if len(sys.argv) > 2 else "all"          # all | obs | pose | point: which outputs are asked for
    import torch
    from visual_slam_amd import ba as BA
    pr = BA.synth_problem(1000, 200000, 5, seed=0)
    plan = BA.BAPlan(pr["obs_pose"], pr["obs_point"], 1000, 200000, device=0)
    M = len(pr["obs_pose"])
    dev = torch.device("cuda", 0)
    t_p = torch.from_numpy(pr["poses"]).to(dev); t_x = torch.from_numpy(pr["points"]).to(dev); t_uv = torch.from_numpy(pr["uv"]).to(dev)
    o_r = torch.empty((M, 2), dtype=torch.float64, device=dev); o_n = torch.empty(M, dtype=torch.float64, device=dev)
    o_Hpp = torch.empty((1000, 6, 6), dtype=torch.float64, device=dev); o_bp = torch.empty((1000, 6), dtype=torch.float64, device=dev)
    o_Hll = torch.empty((200000, 3, 3), dtype=torch.float64, device=dev); o_bl = torch.empty((200000, 3), dtype=torch.float64, device=dev)
    o_Hpl = torch.empty((M, 6, 3), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    outs = {"all": (o_r.data_ptr(), o_n.data_ptr(), o_Hpp.data_ptr(), o_bp.data_ptr(), o_Hll.data_ptr(), o_bl.data_ptr(), o_Hpl.data_ptr()),
            "obs": (o_r.data_ptr(), o_n.data_ptr(), 0, 0, 0, 0, o_Hpl.data_ptr()), "pose": (0, 0, o_Hpp.data_ptr(), o_bp.data_ptr(), 0, 0, 0),
            "point": (0, 0, 0, 0, o_Hll.data_ptr(), o_bl.data_ptr(), 0)}[mode]
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        plan.run_device(t_p.data_ptr(), t_x.data_ptr(), t_uv.data_ptr(), pr["K"], *outs)
        ts.append(time.perf_counter() - t0)
    print("ba pass: median %.4f ms (host wall clock, synchronous call), rms residual %.4f px" % (float(np.median(ts)) * 1e3, float(torch.sqrt((o_n ** 2).mean()))))
    plan.close()


if __name__ == "__main__":
    main()
