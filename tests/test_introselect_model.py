"""The closed-form (data-parallel) replay of libstdc++ nth_element/partition must leave
exactly the permutation libstdc++ leaves (oracle.retain_best calls the real library)."""
import numpy as np
import pytest

from introselect_model import retain_best_model


@pytest.mark.parametrize("seed", range(6))
def test_model_equals_libstdcxx(oracle, seed):
    rng = np.random.default_rng(seed)
    for _ in range(40):
        n = int(rng.integers(2, 3000))
        k = int(rng.integers(1, n + 2))
        kind = rng.integers(0, 4)
        if kind == 0:      # FAST-like: few distinct integer scores, many ties
            keys = rng.integers(20, 40, n).astype(np.float32)
        elif kind == 1:    # Harris-like floats
            keys = rng.random(n).astype(np.float32)
        elif kind == 2:    # sorted / reverse sorted
            keys = np.sort(rng.random(n).astype(np.float32))
            if rng.integers(0, 2):
                keys = keys[::-1].copy()
        else:              # all equal
            keys = np.full(n, 7, np.float32)
        assert np.array_equal(retain_best_model(keys, k), oracle.retain_best(keys, k)), (n, k, kind)


def test_model_heap_fallback_equals_libstdcxx(oracle):
    """Force the depth-limit fallback (heap select) with a median-of-3 killer sequence."""
    def killer(n):  # Musser's adversary for median-of-3 quicksort, ascending comparator
        k = n // 2
        a = np.zeros(n)
        for i in range(1, k + 1):
            if i % 2 == 1:
                a[i - 1] = i
                a[i] = k + i
            a[k + i - 1] = 2 * i
        return a
    hit = 0
    for n in (64, 200, 1024, 4096):
        for sign in (1.0, -1.0):
            keys = (sign * killer(n)).astype(np.float32)
            for k in (n // 2, n // 3, 5, n - 2):
                assert np.array_equal(retain_best_model(keys, k), oracle.retain_best(keys, k)), (n, sign, k)
    # and exercise the fallback code itself regardless of whether the adversary reached it
    rng = np.random.default_rng(0)
    for _ in range(30):
        n = int(rng.integers(8, 400))
        k = int(rng.integers(1, n))
        keys = rng.integers(0, 50, n).astype(np.float32)
        got = retain_best_model(keys, k, force_depth=int(rng.integers(0, 3)))
        thr = np.sort(keys)[::-1][k - 1]
        assert set(got.tolist()) == set(np.nonzero(keys >= thr)[0].tolist())
