"""vo_match_knn2_ratio_noalloc (the entry point that allocates nothing) against vo_match_knn2_ratio,
through the C ABI, bit exact, on two calls in a row, the second with a smaller input."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _descs(rng, n, base=None, flips=24):
    """random 256-bit descriptors; with `base`, noisy copies of its rows (so that the ratio test keeps some)"""
    if base is None:
        return rng.integers(0, 256, (n, 32), dtype=np.uint8)
    src = base[rng.integers(0, len(base), n)]
    bits = np.unpackbits(src, axis=1)
    for r in range(n):
        k = rng.integers(0, flips + 1)
        bits[r, rng.choice(256, k, replace=False)] ^= 1
    return np.packbits(bits, axis=1)


def test_noalloc_matches_existing_entry_point():
    import torch
    from visual_slam_amd import _lib

    rng = np.random.default_rng(7)
    ctx = _lib.Context(max_w=256, max_h=256, chunk_pairs=1, max_pairs=1, nfeatures=3072)   # kp_cap 4096
    cases = [(3000, 2900), (700, 1234)]          # the second call is smaller: stale scratch contents must not leak
    for nq, nt in cases:
        t = _descs(rng, nt)
        q = _descs(rng, nq, base=t)
        want = np.stack(ctx.match_knn2_ratio(q, t, 0.8), 1)               # existing entry point (host arrays)
        dq, dt = torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda()
        cap = nq
        out = torch.full((cap, 4), -7, dtype=torch.int32, device="cuda")
        n = torch.full((1,), -1, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        ctx.match_knn2_ratio_noalloc(dq.data_ptr(), nq, dt.data_ptr(), nt, out.data_ptr(), cap, n.data_ptr(), 0.8)
        torch.cuda.synchronize()
        got_n = int(n.item())
        assert got_n == len(want) and got_n > 50, (got_n, len(want))
        got = out.cpu().numpy()
        assert np.array_equal(got[:got_n], want)
        assert (got[got_n:] == -7).all()
    # a capacity smaller than the number of matches: the count is still the total, only `cap` rows are written
    out = torch.full((16, 4), -7, dtype=torch.int32, device="cuda")
    ctx.match_knn2_ratio_noalloc(dq.data_ptr(), nq, dt.data_ptr(), nt, out.data_ptr(), 16, n.data_ptr(), 0.8)
    torch.cuda.synchronize()
    assert int(n.item()) == len(want) and np.array_equal(out.cpu().numpy(), want[:16])
    # sizes beyond the maximum are rejected with a clear error
    with pytest.raises(_lib.VoError) as e:
        ctx.match_knn2_ratio_noalloc(dq.data_ptr(), 4097, dt.data_ptr(), nt, out.data_ptr(), 16, n.data_ptr())
    assert e.value.code == _lib.VO_ERR_CAPACITY and "4096" in str(e.value)
    ctx.close()
