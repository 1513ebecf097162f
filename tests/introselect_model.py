"""Data-parallel model of libstdc++'s std::nth_element + std::partition as used by
KeyPointsFilter::retainBest.  The HIP kernel `select_topk` is a
transcription of this model: every Hoare partition is resolved in closed form from the
ordered lists of left and right 'stoppers' instead of the two-pointer walk.

comp(a, b) = a > b (descending by response)."""
import numpy as np


def _lg(n):
    return int(n).bit_length() - 1


def _median_to_first(key, idx, result, a, b, c):
    def comp(i, j):
        return key[i] > key[j]

    def swap(i, j):
        key[i], key[j] = key[j], key[i]
        idx[i], idx[j] = idx[j], idx[i]

    if comp(a, b):
        if comp(b, c):
            swap(result, b)
        elif comp(a, c):
            swap(result, c)
        else:
            swap(result, a)
    elif comp(a, c):
        swap(result, a)
    elif comp(b, c):
        swap(result, c)
    else:
        swap(result, b)


def parallel_hoare(key, idx, lo, hi, pivot):
    """__unguarded_partition(lo, hi, pivot) in closed form.  Returns the cut."""
    seg = key[lo:hi]
    L = lo + np.nonzero(~(seg > pivot))[0]            # ascending positions with a <= p
    R = (lo + np.nonzero(~(pivot > seg))[0])[::-1]    # descending positions with a >= p
    m = min(len(L), len(R))
    K = int(np.count_nonzero(L[:m] < R[:m]))
    assert np.all(L[:K] < R[:K])
    if K:
        a, b = L[:K], R[:K]
        key[a], key[b] = key[b].copy(), key[a].copy()
        idx[a], idx[b] = idx[b].copy(), idx[a].copy()
    if K == 0:
        return int(L[0])
    cut = int(R[K - 1])
    if K < len(L):
        cut = min(cut, int(L[K]))
    return cut


def _heap_select_fallback(key, idx, first, nth, last):
    # sequential replay of std::__heap_select(first, nth+1, last) + iter_swap(first, nth)
    def comp(i_val, j_val):
        return i_val > j_val

    def adjust_heap(first, hole, length, vk, vi):
        top = hole
        child = hole
        while child < (length - 1) // 2:
            child = 2 * (child + 1)
            if comp(key[first + child], key[first + child - 1]):
                child -= 1
            key[first + hole] = key[first + child]; idx[first + hole] = idx[first + child]
            hole = child
        if (length & 1) == 0 and child == (length - 2) // 2:
            child = 2 * (child + 1)
            key[first + hole] = key[first + child - 1]; idx[first + hole] = idx[first + child - 1]
            hole = child - 1
        parent = (hole - 1) // 2
        while hole > top and comp(key[first + parent], vk):
            key[first + hole] = key[first + parent]; idx[first + hole] = idx[first + parent]
            hole = parent
            parent = (hole - 1) // 2
        key[first + hole] = vk; idx[first + hole] = vi

    middle = nth + 1
    length = middle - first
    if length >= 2:
        parent = (length - 2) // 2
        while True:
            adjust_heap(first, parent, length, key[first + parent].copy(), idx[first + parent].copy())
            if parent == 0:
                break
            parent -= 1
    for i in range(middle, last):
        if comp(key[i], key[first]):
            vk, vi = key[i].copy(), idx[i].copy()
            key[i] = key[first]; idx[i] = idx[first]
            adjust_heap(first, 0, length, vk, vi)
    key[first], key[nth] = key[nth].copy(), key[first].copy()
    idx[first], idx[nth] = idx[nth].copy(), idx[first].copy()


def retain_best_model(keys, n_points, force_depth=None):
    key = np.asarray(keys, np.float32).copy()
    n = len(key)
    idx = np.arange(n, dtype=np.int32)
    if not (n_points >= 0 and n > n_points):
        return idx
    if n_points == 0:
        return idx[:0]
    first, last, nth = 0, n, n_points - 1
    depth = _lg(n) * 2 if force_depth is None else force_depth
    done = False
    while last - first > 3:
        if depth == 0:
            _heap_select_fallback(key, idx, first, nth, last)
            done = True
            break
        depth -= 1
        mid = first + (last - first) // 2
        _median_to_first(key, idx, first, first + 1, mid, last - 1)
        cut = parallel_hoare(key, idx, first + 1, last, key[first])
        if cut <= nth:
            first = cut
        else:
            last = cut
    if not done:
        # insertion sort of the <= 3 remaining elements (stable, descending)
        for i in range(first + 1, last):
            j = i
            while j > first and key[j] > key[j - 1]:
                key[j], key[j - 1] = key[j - 1].copy(), key[j].copy()
                idx[j], idx[j - 1] = idx[j - 1].copy(), idx[j].copy()
                j -= 1
    amb = key[n_points - 1]
    # std::partition(begin + n_points, end, response >= amb), closed form
    seg = key[n_points:]
    F = n_points + np.nonzero(~(seg >= amb))[0]
    T = (n_points + np.nonzero(seg >= amb)[0])[::-1]
    m = min(len(F), len(T))
    K = int(np.count_nonzero(F[:m] < T[:m]))
    if K:
        a, b = F[:K], T[:K]
        key[a], key[b] = key[b].copy(), key[a].copy()
        idx[a], idx[b] = idx[b].copy(), idx[a].copy()
    return idx[: n_points + len(T)]
