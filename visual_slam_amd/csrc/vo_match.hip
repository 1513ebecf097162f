// vo_match.hip -- brute-force Hamming 2-NN + Lowe ratio test on MI355X (gfx950).
//
// Stands behind knnMatch(features1, features2, k=2) and the ratio loop at
// /root/reference/libs/vo_utils.py:98-108, with the north-star's exact
// BFMatcher(NORM_HAMMING) semantics: per query the two smallest
// (distance, trainIdx) in lexicographic order.
//
// k_match computes the distance matrix on the matrix cores.  With train bits a in {0,1} and query
// bits b recoded as (1 - 2b) in {+1,-1}, one 32x32 tile of Hamming distances is
//     D[t][q] = |b_q| + sum_k a_tk (1 - 2 b_qk) = |a_t| + |b_q| - 2 a_t.b_q
// i.e. a K = 256 contraction: 4 x v_mfma_scale_f32_32x32x64_f8f6f4 with fp4 (E2M1) operands (round 3; rounds 1-2 used 8 x
// v_mfma_i32_32x32x32_i8, which runs 32 k per 38 cycles where the fp4 form runs 64 k per 45, tools/ubench_mfma_fp4.hip).
// A set train bit is the nibble 0x2 (1.0), a query bit the nibble 0x2 / 0xA (+1 / -1), and both block scales are 2^6, so a
// product is 4096 and the low 12 bits of the sum are free: the f32 accumulator is preloaded with 4096*256 + 0x800 + row --
// every value an integer below 2^24, so the arithmetic is exact -- and the MFMA result IS the packed key (biased distance
// << 12 | biased block-relative trainIdx).  Keys are positive floats, whose bit patterns order like the integers they hold:
// the order of the raw bits is the lexicographic (distance, trainIdx) order, +inf is the sentinel, the re-basing by -32 per
// block (so that the preload never changes; at most 64 blocks per segment) is an exact float subtraction.
// A wavefront keeps MM_NQ query tiles (B operands, expanded once, 16 VGPRs each) in registers; the 4 wavefronts of a workgroup
// share each 32-train block: every lane loads ONE packed dword, expands its 32 bits to nibbles and stores them to a
// double-buffered LDS image (8 regions of 528 B per block) that all wavefronts read back as MFMA A operands (ds_read_b128), so
// HBM/L2 only ever see packed descriptors.  The accumulator layout puts one query on a lane and 16 trains in its registers, so
// the 2-NN scan is 2 VALU ops per distance (v_med3_u32, v_min_u32), scheduled between the next tile's MFMAs.  The train set
// is split into segments (grid.y); k_ratio merges the segments, applies the ratio test in Python-double arithmetic and
// compacts the survivors in ascending query order.
// Lane map of the instruction (checked on exact data by tools/ubench_mfma_fp4.hip): lane (n, h) holds row / column n,
// k = 32 h + j as nibble j of its four operand dwords; D as every 32 x 32 MFMA.  k-step s of half h is dword 4 h + s of the
// descriptor (any k order works as long as both operands use it).
#include <algorithm>

#include "vo_internal.h"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

// query tiles (of 32) a wavefront keeps in registers: 3 = 158 VGPRs, three wavefronts per SIMD.  4 / 5 / 6 tiles (178 / 200 / 231
// VGPRs, two per SIMD; more MFMAs per barrier) measured 59 / 50 / 45.5 us alone against 48.3 us, all the same end to end.
#ifndef VO_MATCH_NQ
#define VO_MATCH_NQ 3
#endif
// a launch splits the train set into as few segments as still give this many workgroups: every segment repeats the query
// tiles' load + expansion, k_ratio merges the segments, and a workgroup's loop is only as long as its segment.  32 pairs x
// 3000: 2 segments of 47 blocks instead of 4 of 24: +1.4 % end to end (57.1 -> 57.9 k pairs/s, three A/B runs each on one box).
#ifndef VO_MATCH_FILL_WGS
#define VO_MATCH_FILL_WGS 512
#endif
#define MATCH_SEG_TRAINS 2016         // trains per segment: 63 blocks of 32 (12-bit block-relative index)
#define BYTE(d, k) (((d) >> (8 * (k))) & 0xFFu)

__device__ __forceinline__ void key_update(uint32_t key, uint32_t& k1, uint32_t& k2) {
    // k1 <= k2: the new second-smallest is the median of (key, k1, k2); this form selects v_med3_u32
    k2 = min(max(key, k1), max(min(key, k1), k2));
    k1 = min(k1, key);
}

#define MM_QPB (4 * MM_NQ * 32)
#define MM_REGION 528                // LDS bytes per (k-step, half) region: 32 trains x 16 B + 16 B pad
#define MM_BUF (8 * MM_REGION)

// 8 descriptor bits -> 8 nibbles, nibble j = 2 * bit j (E2M1 1.0 for a set bit)
__device__ __forceinline__ uint32_t spread8_nib(uint32_t x) {
    uint32_t t = (x | (x << 12)) & 0x000F000Fu;
    t = (t | (t << 6)) & 0x03030303u;
    t = (t | (t << 3)) & 0x11111111u;
    return t << 1;
}
__device__ __forceinline__ v4i expand_dword4(uint32_t w) {
    return v4i{(int)spread8_nib(BYTE(w, 0)), (int)spread8_nib(BYTE(w, 1)), (int)spread8_nib(BYTE(w, 2)), (int)spread8_nib(BYTE(w, 3))};
}
__device__ __forceinline__ v16f mfma4(const v4i& a, const v4i& b, const v16f& c) {
    const v8i a8 = {a[0], a[1], a[2], a[3], 0, 0, 0, 0}, b8 = {b[0], b[1], b[2], b[3], 0, 0, 0, 0};
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c, 4, 4, 0, 133, 0, 133);   // fp4 x fp4, scales 2^(133 - 127)
}

// (v_min_f32 -- full rate where v_min_u32 is half rate, and the same order on these keys -- was tried for the scan through inline
// asm: 49.9 against 48.9 us, no gain; and an asm statement that reads an MFMA result gets no hazard wait states from the compiler:
// scheduled ahead of the v_med3 it read stale registers.)
template <int MM_NQ>
__global__ __launch_bounds__(256) void k_match(const uint8_t* __restrict__ desc, const int* __restrict__ nkp, int kp_cap, int nA,
                                                int nseg, uint2* __restrict__ seg_best, int xb0) {
    __shared__ __attribute__((aligned(16))) uint8_t abuf[2 * MM_BUF];
    const int p = blockIdx.z, seg = blockIdx.y;
    const int qf = p, tf = nA + p;
    const int nq = nkp[qf], nt = nkp[tf];
    const int xb = (int)blockIdx.x + xb0;
    if (xb * MM_QPB >= nq) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, h = lane >> 5;
    const int q0 = xb * MM_QPB + wave * (MM_NQ * 32);
    v4i B[MM_NQ][4];
    int pc[MM_NQ];
#pragma unroll
    for (int u = 0; u < MM_NQ; u++) {
        const int q = min(q0 + 32 * u + n, nq - 1);
        const uint4 raw = *reinterpret_cast<const uint4*>(desc + ((long)qf * kp_cap + q) * 32 + 16 * h);
        const uint32_t rw[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int s = 0; s < 4; s++) {
            B[u][s] = expand_dword4(rw[s]);
#pragma unroll
            for (int j = 0; j < 4; j++) B[u][s][j] = (int)(((uint32_t)B[u][s][j] << 2) | 0x22222222u);   // bit 1 -> 0xA (-1), bit 0 -> 0x2 (+1)
        }
        const int half = __popc(raw.x) + __popc(raw.y) + __popc(raw.z) + __popc(raw.w);
        pc[u] = half + __shfl_xor(half, 32);
    }
    uint32_t k1[MM_NQ], k2[MM_NQ];
#pragma unroll
    for (int u = 0; u < MM_NQ; u++) k1[u] = k2[u] = 0x7F800000u;   // +inf
    const int per = (((nt + nseg - 1) / nseg) + 31) & ~31;
    const int t0 = seg * per, t1 = min(t0 + per, nt);
    const int nblk = t0 < t1 ? (t1 - t0 + 31) >> 5 : 0;
    if (nblk > 0) {   // uniform over the workgroup
        v16f cinit;
#pragma unroll
        for (int i = 0; i < 16; i++) cinit[i] = (float)(4096 * 256 + 0x800 + (i & 3) + 8 * (i >> 2));
        // producer role: thread tid owns dword (tid & 7) of train (tid >> 3) of the block: half = dword >> 2, k-step = dword & 3
        const int pm = tid >> 3, pd = tid & 7;
        const uint32_t* tp = reinterpret_cast<const uint32_t*>(desc + (long)tf * kp_cap * 32) + pd;
        uint8_t* wr = abuf + ((2 * (pd & 3) + (pd >> 2)) * MM_REGION + pm * 16);   // region index = 2 * kstep + half
        const uint8_t* rd = abuf + h * MM_REGION + n * 16;
        *reinterpret_cast<v4i*>(wr) = expand_dword4(tp[(long)min(t0 + pm, nt - 1) * 8]);
        uint32_t raw = nblk > 1 ? tp[(long)min(t0 + 32 + pm, nt - 1) * 8] : 0u;
        __syncthreads();
        v4i A0[4], A1[4];
#pragma unroll
        for (int s = 0; s < 4; s++) A0[s] = *reinterpret_cast<const v4i*>(rd + 2 * s * MM_REGION);
        v16f accP;
#pragma unroll
        for (int i = 0; i < 16; i++) accP[i] = __uint_as_float(0x7F800000u);
        auto block = [&](v4i (&Ac)[4], v4i (&An)[4], int it) {
            if (it + 1 < nblk) {
                *reinterpret_cast<v4i*>(wr + ((it + 1) & 1) * MM_BUF) = expand_dword4(raw);
                if (it + 2 < nblk) raw = tp[(long)min(t0 + 32 * it + 64 + pm, nt - 1) * 8];
            } else if (t0 + 32 * nblk > t1) {
                // ragged last block: rows at or beyond t1 get +1024 on the distance, which reads as a sentinel
#pragma unroll
                for (int i = 0; i < 16; i++)
                    if (t0 + 32 * it + (i & 3) + 8 * (i >> 2) + 4 * h >= t1) cinit[i] += (float)(4096 * 1024);
            }
#pragma unroll
            for (int u = 0; u < MM_NQ; u++) {
                if (u == MM_NQ - 1) {
                    __syncthreads();   // block it+1 is complete in LDS; every wavefront has consumed block it-1's image
                    if (it + 1 < nblk) {
                        const uint8_t* r2 = rd + ((it + 1) & 1) * MM_BUF;
#pragma unroll
                        for (int s = 0; s < 4; s++) An[s] = *reinterpret_cast<const v4i*>(r2 + 2 * s * MM_REGION);
                    }
                }
                const int up = (u + MM_NQ - 1) % MM_NQ;    // the previous tile's query tile
                k1[up] = __float_as_uint(__uint_as_float(k1[up]) - 32.f);   // re-base to the block of accP (inf stays inf)
                k2[up] = __float_as_uint(__uint_as_float(k2[up]) - 32.f);
                // four MFMAs and the 32 VALU ops of the previous tile's scan; the group barriers ask the scheduler for
                // MFMA, 8 VALU, MFMA, 8 VALU ... (the MFMA intrinsic is free to move at the IR level, so source order alone
                // does not hold it in place)
                v16f acc = mfma4(Ac[0], B[u][0], cinit);
#pragma unroll
                for (int s = 1; s < 4; s++) acc = mfma4(Ac[s], B[u][s], acc);
#pragma unroll
                for (int i = 0; i < 16; i++) key_update(__float_as_uint(accP[i]), k1[up], k2[up]);
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);   // eight VALU
                }
                accP = acc;
            }
        };
        for (int it = 0; it < nblk; it += 2) {
            block(A0, A1, it);
            if (it + 1 < nblk) block(A1, A0, it + 1);
        }
        k1[MM_NQ - 1] = __float_as_uint(__uint_as_float(k1[MM_NQ - 1]) - 32.f);
        k2[MM_NQ - 1] = __float_as_uint(__uint_as_float(k2[MM_NQ - 1]) - 32.f);
#pragma unroll
        for (int i = 0; i < 16; i++) key_update(__float_as_uint(accP[i]), k1[MM_NQ - 1], k2[MM_NQ - 1]);
    }
    uint2* out = seg_best + ((long)p * nseg + seg) * kp_cap;
    const int jlast = t0 + 32 * (nblk - 1);
#pragma unroll
    for (int u = 0; u < MM_NQ; u++) {
        // keys below 2^22 are real: distance = (key >> 12) - 256 + |b_q|, trainIdx = jlast + (key & 0xFFF) - 0x800 + 4h
        const float f1 = __uint_as_float(k1[u]), f2 = __uint_as_float(k2[u]);
        const uint32_t x1 = f1 < 4194304.f ? (uint32_t)f1 : 0xFFFFFFFFu, x2 = f2 < 4194304.f ? (uint32_t)f2 : 0xFFFFFFFFu;
        const uint32_t a1 = (x1 >> 22) == 0 ? ((((x1 >> 12) + pc[u] - 256) << 16) + (uint32_t)(jlast + (int)(x1 & 0xFFF) - 0x800 + 4 * h)) : 0xFFFFFFFFu;
        const uint32_t a2 = (x2 >> 22) == 0 ? ((((x2 >> 12) + pc[u] - 256) << 16) + (uint32_t)(jlast + (int)(x2 & 0xFFF) - 0x800 + 4 * h)) : 0xFFFFFFFFu;
        const uint32_t b1 = __shfl_xor(a1, 32), b2 = __shfl_xor(a2, 32);
        const uint32_t m1 = min(a1, b1), m2 = min(max(a1, b1), min(a2, b2));
        const int q = q0 + 32 * u + n;
        if (h == 0 && q < nq) out[q] = make_uint2(m1, m2);
    }
}

// merge segments, ratio test, ordered compaction.  One workgroup per pair.
__global__ __launch_bounds__(1024) void k_ratio(const uint2* __restrict__ seg_best, int nseg, const int* __restrict__ nkp, int kp_cap, int nA,
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
            for (int s = 0; s < nseg; s++) {
                const uint2 v = seg_best[((long)p * nseg + s) * kp_cap + q];
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
            if (kps) {   // uniform; null for the descriptor-only entry point (vo_match_knn2_ratio_noalloc)
                const vo_keypoint ka = kps[(long)qf * kp_cap + q], kb = kps[(long)tf * kp_cap + ti];
                q1[ob] = make_float2(ka.x, ka.y);
                q2[ob] = make_float2(kb.x, kb.y);
            }
        }
        __syncthreads();
        if (tid == 0) s_running = running + tot;
        __syncthreads();
    }
    if (tid == 0) { nmatch[gp] = s_running; pair_status[gp] = VO_OK; }
}

// segments of the train set that d_match has room for: 8 (a single pair then still makes 64 workgroups), and no more than
// MATCH_SEG_TRAINS trains each
int vo_match_segments(int kp_cap) { return std::max(8, (kp_cap + MATCH_SEG_TRAINS - 1) / MATCH_SEG_TRAINS); }

// pairs of the chunk: query frame p, train frame nA + p (descriptors of the last detect)
int vo_launch_match_impl(vo_ctx* c, int P, int nA, double ratio, int pair0, int4* raw_out, hipStream_t st) {
    const int kp_cap = c->kp_cap;
    // d_match holds vo_match_segments(kp_cap) segments per pair; a launch uses as few of them as still give VO_MATCH_FILL_WGS
    // workgroups (every segment repeats the query tiles' load + expansion, and a workgroup's loop is only as long as its segment)
    // Query blocks: a frame has nfeatures keypoints unless responses tie at the cut (then up to kp_cap).  The main launch covers
    // the blocks up to nfeatures -- every workgroup of it has work, and with two resident workgroups per CU (248 VGPRs) 32 pairs x
    // 8 blocks x 2 segments are exactly one round; launched over kp_cap, a quarter of the grid exited at once and the dispatcher
    // spread the working ones unevenly (wavefront slots 65 % occupied over the kernel's time).  The blocks beyond nfeatures get
    // a launch of their own, whose workgroups exit at their first test unless a frame did tie.
    const int qpb = 128 * VO_MATCH_NQ;
    const int gx_all = (kp_cap + qpb - 1) / qpb;
    const int gx = std::min(gx_all, (std::min(c->params.nfeatures, kp_cap) + qpb - 1) / qpb);
    const int nseg_min = (kp_cap + MATCH_SEG_TRAINS - 1) / MATCH_SEG_TRAINS;
    const int nseg = std::min(vo_match_segments(kp_cap), std::max(nseg_min, (VO_MATCH_FILL_WGS + gx * P - 1) / (gx * P)));
    KT(c, st, VO_K_MATCH, hipLaunchKernelGGL(k_match<VO_MATCH_NQ>, dim3(gx, nseg, P), dim3(256), 0, st, (const uint8_t*)c->d_desc.p,
                                             (const int*)c->d_nkp.p, kp_cap, nA, nseg, (uint2*)c->d_match.p, 0));
    if (gx_all > gx)
        hipLaunchKernelGGL(k_match<VO_MATCH_NQ>, dim3(gx_all - gx, nseg, P), dim3(256), 0, st, (const uint8_t*)c->d_desc.p, (const int*)c->d_nkp.p,
                           kp_cap, nA, nseg, (uint2*)c->d_match.p, gx);
    KT(c, st, VO_K_RATIO, hipLaunchKernelGGL(k_ratio, dim3(P), dim3(1024), 0, st, (const uint2*)c->d_match.p, nseg, (const int*)c->d_nkp.p, kp_cap, nA,
                       (const vo_keypoint*)c->d_kp.p, ratio, pair0, (int4*)c->mq(), (float2*)c->q1(), (float2*)c->q2(),
                       (int*)c->nmatch(), (int*)c->pair_status(), (int2*)c->pair_nkp(), (const int*)c->d_frame_status.p, raw_out));
    HIP_TRY(c, hipGetLastError());
    return VO_OK;
}

// ------------------------------------------------------------------ the entry point that allocates nothing
// vo_match_knn2_ratio_noalloc (include/vo_mi355.h): k_match + k_ratio with their workspace (the staged descriptors,
// the per-segment best/second keys, the compacted match rows and a few counters) in ONE file-scope device array
// instead of the context's hipMalloc'ed buffers.  Each GPU holds its own copy of the array at its own address:
// vo_noalloc_init (from vo_ctx_create, with the context's device current) asks the runtime for that address once per
// device; host code only ever uses that pointer -- the array's name is a host-side stand-in, not a device address.
#define NA_CAP VO_NOALLOC_MAX_DESC
#define NA_NSEG 4                                             // vo_match_segments(NA_CAP): 4 x 1024 trains
#define NA_OFF_DESC 0L                                        // uint8 [2][NA_CAP][32]: query frame, train frame
#define NA_OFF_SEG (NA_OFF_DESC + 2L * NA_CAP * 32)           // uint2 [NA_NSEG][NA_CAP]
#define NA_OFF_MQ (NA_OFF_SEG + (long)NA_NSEG * NA_CAP * 8)   // int4 [NA_CAP]
#define NA_OFF_SMALL (NA_OFF_MQ + (long)NA_CAP * 16)          // int nkp[2], frame_status[2], nmatch, pair_status, pair_nkp[2]
#define NA_BYTES (NA_OFF_SMALL + 256)
static_assert(NA_BYTES == 459008, "the documented size of the no-allocation scratch (include/vo_mi355.h, DESIGN.md, INTEGRATION.md)");
static __device__ __align__(256) unsigned char scratch[NA_BYTES];   // 459,008 B per GPU, held until the process ends
static void* g_scratch_ptr[64];                                     // per device: the address of ITS copy of `scratch`

int vo_noalloc_init(vo_ctx* c) {
    if (c->device < 0 || c->device >= 64) return VO_OK;   // such a context reports the missing set-up at the call
    if (g_scratch_ptr[c->device]) return VO_OK;
    void* p = nullptr;
    HIP_TRY(c, hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    g_scratch_ptr[c->device] = p;
    return VO_OK;
}

__global__ void k_noalloc_setup(int* __restrict__ small, int nq, int nt) {
    if (threadIdx.x == 0) {
        small[0] = nq; small[1] = nt;    // nkp of the query / train frame
        small[2] = 0; small[3] = 0;      // frame_status
        small[4] = 0; small[5] = 0;      // nmatch, pair_status (k_ratio writes both)
    }
}

__global__ void k_noalloc_out(const int4* __restrict__ mq, const int* __restrict__ small, int cap, int4* __restrict__ matches, int* __restrict__ n_out) {
    const int n = small[4];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *n_out = n;
    if (i < n && i < cap) matches[i] = mq[i];
}

extern "C" int vo_match_knn2_ratio_noalloc(vo_ctx* c, const uint8_t* q_dev, int nq, const uint8_t* t_dev, int nt, double ratio,
                                           int32_t* matches_dev, int cap, int32_t* n_dev, void* stream) {
    if (!c) return VO_ERR_INVALID;
    if (!q_dev || !t_dev || !n_dev || nq < 0 || nt < 0 || cap < 0 || (cap > 0 && !matches_dev))
        return vo_fail(c, VO_ERR_INVALID, "vo_match_knn2_ratio_noalloc: bad arguments");
    if (nq > NA_CAP || nt > NA_CAP)
        return vo_fail(c, VO_ERR_CAPACITY, "vo_match_knn2_ratio_noalloc: more than VO_NOALLOC_MAX_DESC (4096) descriptors in a set");
    if (nt < 2) return vo_fail(c, VO_ERR_TOO_FEW, "knnMatch(k=2) needs at least 2 train descriptors");
    if (c->device < 0 || c->device >= 64 || !g_scratch_ptr[c->device])
        return vo_fail(c, VO_ERR_HIP, "vo_match_knn2_ratio_noalloc: the device's scratch address was not resolved at vo_ctx_create");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    uint8_t* base = (uint8_t*)g_scratch_ptr[c->device];
    uint8_t* desc = base + NA_OFF_DESC;
    uint2* seg = (uint2*)(base + NA_OFF_SEG);
    int4* mq = (int4*)(base + NA_OFF_MQ);
    int* small = (int*)(base + NA_OFF_SMALL);
    hipLaunchKernelGGL(k_noalloc_setup, dim3(1), dim3(64), 0, st, small, nq, nt);
    if (nq) HIP_TRY(c, hipMemcpyAsync(desc, q_dev, (size_t)nq * 32, hipMemcpyDeviceToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(desc + (long)NA_CAP * 32, t_dev, (size_t)nt * 32, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(k_match<VO_MATCH_NQ>, dim3((NA_CAP + 128 * VO_MATCH_NQ - 1) / (128 * VO_MATCH_NQ), NA_NSEG, 1), dim3(256), 0, st, (const uint8_t*)desc, (const int*)small, NA_CAP, 1, NA_NSEG, seg, 0);
    hipLaunchKernelGGL(k_ratio, dim3(1), dim3(1024), 0, st, (const uint2*)seg, NA_NSEG, (const int*)small, NA_CAP, 1, (const vo_keypoint*)nullptr, ratio, 0, mq,
                       (float2*)nullptr, (float2*)nullptr, small + 4, small + 5, (int2*)(small + 6), (const int*)(small + 2), (int4*)nullptr);
    hipLaunchKernelGGL(k_noalloc_out, dim3((NA_CAP + 255) / 256), dim3(256), 0, st, (const int4*)mq, (const int*)small, cap, (int4*)matches_dev, (int*)n_dev);
    HIP_TRY(c, hipGetLastError());
    return VO_OK;
}
