// vo_match.hip -- brute-force Hamming 2-NN + Lowe ratio test on MI355X (gfx950).
//
// Stands behind knnMatch(features1, features2, k=2) and the ratio loop at
// /root/reference/libs/vo_utils.py:98-108, with the north-star's exact
// BFMatcher(NORM_HAMMING) semantics: per query the two smallest
// (distance, trainIdx) in lexicographic order.
//
// k_match: one lane owns one query descriptor (8 dwords in VGPRs); the train descriptors
// are wave-uniform, so they stream through scalar loads (SGPR operands of v_xor).  Per
// (query, train): 8 v_xor_b32 + 8 accumulating v_bcnt_u32_b32 + 3 integer ops on the
// packed key (distance << 16 | trainIdx), whose unsigned order IS the lexicographic
// (distance, trainIdx) order: best = min(key, best); second = med3(key, best, second).
// The train set is split into MATCH_SPLIT segments to fill the chip; k_ratio merges the
// segments, applies the ratio test in Python-double arithmetic and compacts the survivors
// in ascending query order.  Integer ALU bound: 19 lane-ops per distance.
#include "vo_internal.h"

#define MATCH_SPLIT 8
#define MATCH_QPB 512   // queries per 256-thread block: two per lane, so every scalar train load feeds 2x19 VALU ops

__device__ __forceinline__ void match_update(const uint32_t (&a)[8], const uint32_t* __restrict__ b, uint32_t j, uint32_t& k1, uint32_t& k2) {
    uint32_t d = __popc(a[0] ^ b[0]);
    d += __popc(a[1] ^ b[1]);
    d += __popc(a[2] ^ b[2]);
    d += __popc(a[3] ^ b[3]);
    d += __popc(a[4] ^ b[4]);
    d += __popc(a[5] ^ b[5]);
    d += __popc(a[6] ^ b[6]);
    d += __popc(a[7] ^ b[7]);
    const uint32_t key = (d << 16) + j;
    const uint32_t lo = min(key, k1), hi = max(key, k1);
    k2 = min(k2, hi);
    k1 = lo;
}

__global__ __launch_bounds__(256) void k_match(const uint8_t* __restrict__ desc, const int* __restrict__ nkp, int kp_cap, int nA,
                                               uint2* __restrict__ seg_best) {
    const int p = blockIdx.z, seg = blockIdx.y;
    const int qf = p, tf = nA + p;
    const int nq = nkp[qf], nt = nkp[tf];
    if (blockIdx.x * MATCH_QPB >= nq) return;
    const int qa = blockIdx.x * MATCH_QPB + threadIdx.x, qb = qa + 256;
    uint32_t a[8], c[8];
    {
        const uint32_t* qp = reinterpret_cast<const uint32_t*>(desc + ((long)qf * kp_cap + min(qa, nq - 1)) * 32);
        uint4 lo = *reinterpret_cast<const uint4*>(qp), hi = *reinterpret_cast<const uint4*>(qp + 4);
        a[0] = lo.x; a[1] = lo.y; a[2] = lo.z; a[3] = lo.w; a[4] = hi.x; a[5] = hi.y; a[6] = hi.z; a[7] = hi.w;
        qp = reinterpret_cast<const uint32_t*>(desc + ((long)qf * kp_cap + min(qb, nq - 1)) * 32);
        lo = *reinterpret_cast<const uint4*>(qp); hi = *reinterpret_cast<const uint4*>(qp + 4);
        c[0] = lo.x; c[1] = lo.y; c[2] = lo.z; c[3] = lo.w; c[4] = hi.x; c[5] = hi.y; c[6] = hi.z; c[7] = hi.w;
    }
    const int per = (nt + MATCH_SPLIT - 1) / MATCH_SPLIT;
    const int t0 = seg * per, t1 = min(t0 + per, nt);
    const uint32_t* tp = reinterpret_cast<const uint32_t*>(desc + (long)tf * kp_cap * 32);
    uint32_t ka1 = 0xFFFFFFFFu, ka2 = 0xFFFFFFFFu, kc1 = 0xFFFFFFFFu, kc2 = 0xFFFFFFFFu;
    int j = t0;
    for (; j + 4 <= t1; j += 4) {
        // four wave-uniform train descriptors (scalar loads) in flight per iteration
        const uint32_t* b = tp + (long)j * 8;
        uint32_t t[32];
#pragma unroll
        for (int k = 0; k < 32; k++) t[k] = b[k];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            match_update(a, t + 8 * u, (uint32_t)(j + u), ka1, ka2);
            match_update(c, t + 8 * u, (uint32_t)(j + u), kc1, kc2);
        }
    }
    for (; j < t1; j++) {
        const uint32_t* b = tp + (long)j * 8;
        match_update(a, b, (uint32_t)j, ka1, ka2);
        match_update(c, b, (uint32_t)j, kc1, kc2);
    }
    uint2* out = seg_best + ((long)p * MATCH_SPLIT + seg) * kp_cap;
    if (qa < nq) out[qa] = make_uint2(ka1, ka2);
    if (qb < nq) out[qb] = make_uint2(kc1, kc2);
}

// merge segments, ratio test, ordered compaction.  One workgroup per pair.
__global__ __launch_bounds__(1024) void k_ratio(const uint2* __restrict__ seg_best, const int* __restrict__ nkp, int kp_cap, int nA,
                                                const vo_keypoint* __restrict__ kps, double ratio, int pair0,
                                                int4* __restrict__ mq, float2* __restrict__ q1, float2* __restrict__ q2,
                                                int* __restrict__ nmatch, int* __restrict__ pair_status, int2* __restrict__ pair_nkp,
                                                const int* __restrict__ frame_status, int4* __restrict__ raw_out) {
    __shared__ int wtot[16];
    __shared__ int s_running;
    const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qf = p, tf = nA + p;
    const int nq = nkp[qf], nt = nkp[tf];
    const long gp = (long)(pair0 + p);
    if (tid == 0) { s_running = 0; pair_nkp[gp] = make_int2(nq, nt); }
    __syncthreads();
    const int fst = frame_status[qf] != VO_OK ? frame_status[qf] : frame_status[tf];
    if (nt < 2 || fst != VO_OK) {
        if (tid == 0) { nmatch[gp] = 0; pair_status[gp] = fst != VO_OK ? fst : VO_ERR_TOO_FEW; }
        return;
    }
    for (int base = 0; base < nq; base += 1024) {
        const int q = base + tid;
        bool keep = false;
        uint32_t k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;
        if (q < nq) {
#pragma unroll
            for (int s = 0; s < MATCH_SPLIT; s++) {
                const uint2 v = seg_best[((long)p * MATCH_SPLIT + s) * kp_cap + q];
                uint32_t lo = min(v.x, k1), hi = max(v.x, k1);
                k2 = min(k2, hi); k1 = lo;
                lo = min(v.y, k1); hi = max(v.y, k1);
                k2 = min(k2, hi); k1 = lo;
            }
            const int d1 = (int)(k1 >> 16), d2 = (int)(k2 >> 16);
            keep = (double)(float)d1 < ratio * (double)(float)d2;  // m.distance < ratio*n.distance in Python doubles
            if (raw_out) raw_out[q] = make_int4((int)(k1 & 0xFFFF), d1, (int)(k2 & 0xFFFF), d2);
        }
        const unsigned long long m = __ballot(keep);
        if (lane == 0) wtot[wave] = __popcll(m);
        __syncthreads();
        int wbase = 0, tot = 0;
        for (int w2 = 0; w2 < 16; w2++) {
            if (w2 < wave) wbase += wtot[w2];
            tot += wtot[w2];
        }
        const int running = s_running;
        if (keep) {
            const int o = running + wbase + __popcll(m & ((1ull << lane) - 1ull));
            const int ti = (int)(k1 & 0xFFFF);
            const long ob = gp * kp_cap + o;
            mq[ob] = make_int4(q, ti, (int)(k1 >> 16), (int)(k2 >> 16));
            const vo_keypoint ka = kps[(long)qf * kp_cap + q], kb = kps[(long)tf * kp_cap + ti];
            q1[ob] = make_float2(ka.x, ka.y);
            q2[ob] = make_float2(kb.x, kb.y);
        }
        __syncthreads();
        if (tid == 0) s_running = running + tot;
        __syncthreads();
    }
    if (tid == 0) { nmatch[gp] = s_running; pair_status[gp] = VO_OK; }
}

// pairs of the chunk: query frame p, train frame nA + p (descriptors of the last detect)
int vo_launch_match_impl(vo_ctx* c, int P, int nA, double ratio, int pair0, int4* raw_out, hipStream_t st) {
    const int kp_cap = c->kp_cap;
    dim3 grid((kp_cap + MATCH_QPB - 1) / MATCH_QPB, MATCH_SPLIT, P);
    KT(c, st, VO_K_MATCH, hipLaunchKernelGGL(k_match, grid, dim3(256), 0, st, (const uint8_t*)c->d_desc.p, (const int*)c->d_nkp.p, kp_cap, nA, (uint2*)c->d_match.p));
    KT(c, st, VO_K_RATIO, hipLaunchKernelGGL(k_ratio, dim3(P), dim3(1024), 0, st, (const uint2*)c->d_match.p, (const int*)c->d_nkp.p, kp_cap, nA,
                       (const vo_keypoint*)c->d_kp.p, ratio, pair0, (int4*)c->d_mq.p, (float2*)c->d_q1.p, (float2*)c->d_q2.p,
                       (int*)c->d_nmatch.p, (int*)c->d_pair_status.p, (int2*)c->d_pair_nkp.p, (const int*)c->d_frame_status.p, raw_out));
    HIP_TRY(c, hipGetLastError());
    return VO_OK;
}
