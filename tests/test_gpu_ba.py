"""BA residual / Jacobian pass through the C ABI vs the numpy oracle (fp64).
Tolerances: per-observation quantities 1e-12 relative (same formulas, libm vs device cos/sin),
accumulated blocks 1e-10 relative to the block's largest entry (summation order differs)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _close(a, b, rel):
    scale = max(1.0, float(np.abs(b).max()))
    return float(np.abs(a - b).max()) <= rel * scale


def test_ba_pass_matches_oracle():
    from oracle import ba_oracle as BO
    from visual_slam_amd import ba
    pr = ba.synth_problem(n_poses=40, n_points=700, seed=2)
    plan = ba.BAPlan(pr["obs_pose"], pr["obs_point"], 40, 700)
    out = plan.run(pr["poses"], pr["points"], pr["uv"], pr["K"])
    ref = BO.ba_pass(pr["poses"], pr["points"], pr["obs_pose"], pr["obs_point"], pr["uv"], pr["K"])
    plan.close()
    for k in ("r", "norm", "Hpl"):
        assert _close(out[k], ref[k], 1e-12), k
    for k in ("Hpp", "bp", "Hll", "bl"):
        assert _close(out[k], ref[k], 1e-10), k
    assert np.allclose(out["Hpp"], np.transpose(out["Hpp"], (0, 2, 1)), rtol=0, atol=1e-9 * np.abs(out["Hpp"]).max())


def test_ba_edge_cases():
    from oracle import ba_oracle as BO
    from visual_slam_amd import _lib, ba
    # a pose with zero rotation, a pose without observations, unsorted observations
    rng = np.random.default_rng(5)
    poses = np.zeros((3, 6)); poses[1, :3] = [0.02, -0.3, 0.01]; poses[1, 3:] = [0.1, 0, -0.5]; poses[2, :3] = [1e-20, 0, 0]
    pts = np.stack([rng.uniform(-2, 2, 9), rng.uniform(-1, 1, 9), rng.uniform(5, 20, 9)], 1)
    obs_pose = np.array([1, 0, 1, 0, 0, 1, 0], np.int32); obs_point = np.array([8, 0, 3, 3, 5, 0, 8], np.int32)
    K = np.array([[700.0, 0, 600], [0, 710.0, 180], [0, 0, 1]])
    uv = np.array([BO.project(poses[j], pts[i], K) for j, i in zip(obs_pose, obs_point)]) + rng.normal(0, 1, (7, 2))
    plan = ba.BAPlan(obs_pose, obs_point, 3, 9)
    out = plan.run(poses, pts, uv, K)
    plan.close()
    ref = BO.ba_pass(poses, pts, obs_pose, obs_point, uv, K)
    for k in ref:
        assert _close(out[k], ref[k], 1e-11), k
    assert not out["Hpp"][2].any() and not out["Hll"][1].any()
    with pytest.raises(_lib.VoError):
        ba.BAPlan(np.array([0, 5], np.int32), np.array([0, 0], np.int32), 3, 9)   # pose index out of range


def test_ba_config5_properties():
    """Full-size structure check on a reduced config 5 (100 poses x 20k points x 5): linearity of the
    residual in uv and block sums equal to sums of the per-observation products."""
    import torch
    from visual_slam_amd import ba
    pr = ba.synth_problem(n_poses=100, n_points=20000, seed=0)
    plan = ba.BAPlan(pr["obs_pose"], pr["obs_point"], 100, 20000)
    out = plan.run(pr["poses"], pr["points"], pr["uv"], pr["K"])
    out2 = plan.run(pr["poses"], pr["points"], pr["uv"] + 1.0, pr["K"])
    plan.close()
    assert np.allclose(out2["r"], out["r"] - 1.0, rtol=0, atol=1e-9)
    # noise sigma 0.5 px: rms residual ~ 0.5 * sqrt(2)
    rms = float(np.sqrt((out["norm"] ** 2).mean()))
    assert 0.6 < rms < 0.8
    # sum over a point's observations of Hpl relates to nothing else, but Hll must be PSD and Hpp PSD
    w = np.linalg.eigvalsh(out["Hpp"])
    assert (w > -1e-6 * np.abs(w).max()).all()
    w = np.linalg.eigvalsh(out["Hll"])
    assert (w > -1e-6 * np.abs(w).max()).all()


def test_ba_point_sums_fused_and_separate_agree():
    """Observations sorted by point take the fused path (point blocks summed inside the observation role, blocks cut at point
    boundaries; points nobody observes are zeroed); the same problem with its observations shuffled, or with one point seen
    more than 256 times, takes the separate point role.  Both must give the oracle's blocks, and for the same terms the
    same bits (the sums run in observation order either way)."""
    from oracle import ba_oracle as BO
    from visual_slam_amd import ba
    rng = np.random.default_rng(11)
    P, N = 12, 400
    poses = np.zeros((P, 6)); poses[:, :3] = rng.normal(0, 0.05, (P, 3)); poses[:, 3:] = rng.normal(0, 0.3, (P, 3))
    pts = np.stack([rng.uniform(-3, 3, N), rng.uniform(-2, 2, N), rng.uniform(6, 30, N)], 1)
    counts = rng.integers(0, 9, N); counts[7] = 0; counts[N - 1] = 0; counts[100] = 300          # unobserved points; one point > 256 later
    K = np.array([[700.0, 0, 600], [0, 710.0, 180], [0, 0, 1]])

    def problem(cnt):
        op = np.repeat(np.arange(N, dtype=np.int32), cnt)
        oj = rng.integers(0, P, len(op)).astype(np.int32)
        uv = np.array([BO.project(poses[j], pts[i], K) for j, i in zip(oj, op)]) + rng.normal(0, 1, (len(op), 2))
        return oj, op, uv

    small = counts.copy(); small[100] = 5
    for cnt in (small, counts):                      # fused; then not fusable (a point with 300 observations)
        oj, op, uv = problem(cnt)
        plan = ba.BAPlan(oj, op, P, N)
        out = plan.run(poses, pts, uv, K)
        plan.close()
        ref = BO.ba_pass(poses, pts, oj, op, uv, K)
        for k in ("r", "norm", "Hpl"):
            assert _close(out[k], ref[k], 1e-12), k
        for k in ("Hpp", "bp", "Hll", "bl"):
            assert _close(out[k], ref[k], 1e-10), k
        assert not out["Hll"][7].any() and not out["bl"][N - 1].any()
        # shuffled observations: the separate point role; a point's terms are then summed in another order
        perm = rng.permutation(len(op))
        plan = ba.BAPlan(oj[perm], op[perm], P, N)
        out2 = plan.run(poses, pts, uv[perm], K)
        plan.close()
        assert np.array_equal(out2["r"], out["r"][perm]) and np.array_equal(out2["Hpl"], out["Hpl"][perm])
        assert _close(out2["Hll"], out["Hll"], 1e-12) and _close(out2["bl"], out["bl"], 1e-12) and _close(out2["Hpp"], out["Hpp"], 1e-12)


def test_levenberg_marquardt_on_the_blocks_converges():
    """The blocks ARE the normal equations of the reprojection problem (/root/reference/vo.py:112-135 hands the same
    residual to scipy least_squares): a few damped Gauss-Newton steps through the Schur complement (ba.schur_step, host numpy
    on the GPU pass's output) bring perturbed poses and points back to the pixel-noise floor, and the cost never rises."""
    from visual_slam_amd import ba
    P, N, noise = 6, 240, 0.3
    pr = ba.synth_problem(n_poses=P, n_points=N, obs_per_point=4, seed=11, noise=noise)
    rng = np.random.default_rng(3)
    poses = pr["poses"].copy(); points = pr["points"].copy()
    poses[1:, :3] += rng.normal(0, 0.004, (P - 1, 3)); poses[1:, 3:] += rng.normal(0, 0.05, (P - 1, 3))
    points += rng.normal(0, 0.15, points.shape)
    plan = ba.BAPlan(pr["obs_pose"], pr["obs_point"], P, N)
    try:
        cost = lambda o: float((o["r"] ** 2).sum())
        out = plan.run(poses, points, pr["uv"], pr["K"])
        c0 = cost(out); lam = 1e-3; costs = [c0]
        for _ in range(12):
            dp, dl = ba.schur_step(out, pr["obs_pose"], pr["obs_point"], lam, fixed_poses=(0,))
            trial = plan.run(poses + dp, points + dl, pr["uv"], pr["K"])
            if cost(trial) < costs[-1]:
                poses, points, out = poses + dp, points + dl, trial
                costs.append(cost(trial)); lam = max(lam / 5, 1e-9)
            else:
                lam *= 10
    finally:
        plan.close()
    M = len(pr["uv"])
    rms0, rms = np.sqrt(c0 / (2 * M)), np.sqrt(costs[-1] / (2 * M))
    assert rms0 > 5 * noise                       # the start was far from the optimum
    assert rms < 1.15 * noise, (rms0, rms, costs) # the end is at the noise floor (fitted parameters absorb a little of it)
    assert all(b < a for a, b in zip(costs, costs[1:]))
