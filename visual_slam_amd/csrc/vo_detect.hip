// vo_detect.hip -- ORB detect + describe for batches of frames on MI355X (gfx950).
//
// Stands behind orb.detectAndCompute(img, None) at /root/reference/libs/vo_utils.py:94-95
// (OpenCV 4.8.0 ORB_Impl::detectAndCompute).  Stages, one launch each over
// every frame of the chunk:
//   k_pyramid (x7)  cascaded INTER_LINEAR_EXACT resize, 8.8 fixed point (4 px x 4 rows per lane)
//   k_fast_tiles    FAST-9/16 + cornerScore + 3x3 NMS on 60x30 LDS tiles: pretest -> polarity
//                   queues -> arc scoring -> queue-driven NMS -> bitmap-rank compaction   [VALU bound]
//   k_fast_emit     candidates to per-level lists in raster order (one workgroup per row of tiles prefixes the per-tile
//                   and per-(row, tile) counts of k_fast_tiles along both axes in LDS)
//   k_select        KeyPointsFilter::retainBest: data-parallel replay of libstdc++
//                   nth_element + partition (tests/introselect_model.py), lists in LDS
//   k_harris        7x7 Harris response at the kept candidates
//   k_select        second retainBest on the Harris response
//   k_orient        intensity-centroid angle (v_sad_u8 / v_dot4_u32_u8 moments) + KeyPoint records
//   k_blur          7x7 sigma=2 Gaussian of every level (sepFilter2D arithmetic), second stream
//   k_rbrief        rotated BRIEF-256 from LDS-staged patches, one wavefront per keypoint
// The per-frame gather/tile kernels use an XCD-contiguous block order (see xcd_block below).
// All integer stages are bit exact by construction; float stages are compiled with
// -ffp-contract=off and spell out fmaf where the CPU build of OpenCV contracts.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "../../include/vo_sincosf.h"
#include "vo_internal.h"

// ------------------------------------------------------------------ XCD-aware block order
// Workgroups are dealt round-robin over the 8 XCDs (block b and b+8 share an L2).  Every per-frame kernel
// is launched as a 1-D grid padded to a multiple of 8 blocks and maps block b to work item
// (b % 8) * G/8 + b / 8, so each XCD walks a contiguous run of frames: a frame's pixels (and the
// halos neighbouring tiles share) are fetched into ONE L2 instead of eight.  Items are ordered
// x fastest, then y, then frame.
__device__ __forceinline__ int xcd_contiguous(int b, int grid) { return (b & 7) * (grid >> 3) + (b >> 3); }
static inline int pad8(int n) { return (n + 7) & ~7; }
struct VBlock { int x, y, f; bool valid; };
__device__ __forceinline__ VBlock xcd_block(int nx, int ny, int nf) {
    const int v = xcd_contiguous(blockIdx.x, gridDim.x);
    VBlock b;
    b.valid = v < nx * ny * nf;
    b.f = v / (nx * ny);
    const int r = v - b.f * (nx * ny);
    b.y = r / nx;
    b.x = r - b.y * nx;
    return b;
}


// ------------------------------------------------------------------ host: geometry
static inline int cvRoundF(float v) { return (int)nearbyintf(v); }
static inline int cvRoundD(double v) { return (int)nearbyint(v); }
static inline int cvFloorD(double v) { int i = (int)v; return i - (i > v); }
static inline long align_up(long v, long a) { return (v + a - 1) / a * a; }

static void linear_coeffs(int src, int dst, ResizeCoef* out) {
    // interpolationLinear::getCoeffs of resize_bitExact
    double inv_scale = (double)dst / src;
    double scale = 1.0 / inv_scale;
    for (int d = 0; d < dst; d++) {
        double f = scale * ((double)d + 0.5) - 0.5;
        int i = cvFloorD(f);
        // packed: source index | fraction c1 << 16 (c0 = 256 - c1); clamped ends have c1 = 0
        ResizeCoef c = 0;
        if (i >= 0 && src > 1) {
            if (i < src - 1) c = (uint32_t)i | ((uint32_t)cvRoundD((f - (double)i) * 256.0) << 16);
            else c = (uint32_t)(src - 1);
        }
        out[d] = c;
    }
}

int vo_build_geometry(vo_ctx* c, int w, int h) {
    if (c->geom_valid && c->geom.w == w && c->geom.h == h) return VO_OK;
    const vo_orb_params& p = c->params;
    if (w > c->max_w || h > c->max_h) return vo_fail(c, VO_ERR_INVALID, "image larger than the context was created for");
    if (w < 64 || h < 64 || w > 4095 || h > 4095) return vo_fail(c, VO_ERR_INVALID, "image size out of range (64..4095)");
    Geom g;
    memset(&g, 0, sizeof(g));
    g.nlevels = p.nlevels; g.w = w; g.h = h;
    g.fast_threshold = p.fast_threshold; g.blur_mode = p.blur_mode; g.fma_atan = p.fma_atan; g.fma_blur = p.fma_blur; g.trig_mode = p.trig_mode;
    const double scaleFactor = (double)p.scale_factor;
    long pyr_ofs = 0, blur_ofs = 0, cand_ofs = 0;
    int tile_base = 0, seg_base = 0, btile_base = 0, rt = 0;
    for (int l = 0; l < p.nlevels; l++) {
        LevelGeom& L = g.lv[l];
        L.scale = (float)std::pow(scaleFactor, (double)l);
        float inv = 1.f / L.scale;
        L.w = cvRoundF(w * inv);
        L.h = cvRoundF(h * inv);
        if (L.w < 8 || L.h < 8) return vo_fail(c, VO_ERR_INVALID, "pyramid level too small");
        L.pitch = (int)align_up(L.w, 64);
        L.pyr_ofs = pyr_ofs;
        if (l > 0) pyr_ofs += align_up((long)L.pitch * L.h, 256);
        L.bpitch = (int)align_up(L.w, 64);
        L.blur_ofs = blur_ofs;
        blur_ofs += align_up((long)L.bpitch * L.h, 256);
        bool has = (L.w > 2 * VO_EDGE) && (L.h > 2 * VO_EDGE);
        L.iw = has ? L.w - 2 * VO_EDGE : 0;
        L.ih = has ? L.h - 2 * VO_EDGE : 0;
        L.ntx = (L.iw + VO_TILE_W - 1) / VO_TILE_W;
        L.nty = (L.ih + VO_TILE_H - 1) / VO_TILE_H;
        L.tile_base = tile_base;
        tile_base += L.ntx * L.nty;
        L.seg_base = seg_base;
        seg_base += (int)align_up((long)L.ih * L.ntx, 16);
        L.cand_cap = (int)align_up((long)((L.iw + 1) / 2) * ((L.ih + 1) / 2), 64);
        L.cand_ofs = cand_ofs;
        cand_ofs += L.cand_cap;
        if (l > 0) {
            L.xtab = rt; rt += (int)align_up(L.w, 4);   // 16-byte aligned rows of 4 coefficients
            L.ytab = rt; rt += (int)align_up(L.h, 4);
        }
        L.bntx = (L.w + VO_BLUR_TW - 1) / VO_BLUR_TW;
        L.bnty = (L.h + VO_BLUR_TH - 1) / VO_BLUR_TH;
        L.btile_base = btile_base;
        btile_base += L.bntx * L.bnty;
    }
    // per-level quota
    {
        float factor = (float)(1.0 / scaleFactor);
        float nd = p.nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)p.nlevels));
        int sum = 0;
        for (int l = 0; l < p.nlevels - 1; l++) {
            g.lv[l].quota = cvRoundF(nd);
            sum += g.lv[l].quota;
            nd *= factor;
        }
        g.lv[p.nlevels - 1].quota = std::max(p.nfeatures - sum, 0);
    }
    g.ntiles = tile_base; g.nsegs = seg_base; g.nbtiles = btile_base;
    g.pyr_frame = align_up(pyr_ofs, 256); g.blur_frame = align_up(blur_ofs, 256);
    g.cand_frame = cand_ofs;
    g.kp_cap = c->kp_cap;

    c->h_tiles.assign(std::max(g.ntiles, 1), TileDesc{0, 0, 0, 0});
    c->h_btiles.assign(std::max(g.nbtiles, 1), TileDesc{0, 0, 0, 0});
    c->h_rtab.assign(std::max(rt, 1), 0);
    for (int l = 0; l < p.nlevels; l++) {
        const LevelGeom& L = g.lv[l];
        for (int ty = 0; ty < L.nty; ty++)
            for (int tx = 0; tx < L.ntx; tx++) c->h_tiles[L.tile_base + ty * L.ntx + tx] = TileDesc{(short)l, (short)tx, (short)ty, 0};
        for (int ty = 0; ty < L.bnty; ty++)
            for (int tx = 0; tx < L.bntx; tx++) c->h_btiles[L.btile_base + ty * L.bntx + tx] = TileDesc{(short)l, (short)tx, (short)ty, 0};
        if (l > 0) {
            linear_coeffs(g.lv[l - 1].w, L.w, &c->h_rtab[L.xtab]);
            linear_coeffs(g.lv[l - 1].h, L.h, &c->h_rtab[L.ytab]);
            // the tables are read four entries at a time: the padding repeats the last coefficient, so a lane never has to
            // clamp an index (outputs past the level's edge are computed from it and masked out of the store)
            for (long d = L.w; d < align_up(L.w, 4); d++) c->h_rtab[L.xtab + d] = c->h_rtab[L.xtab + L.w - 1];
            for (long d = L.h; d < align_up(L.h, 4); d++) c->h_rtab[L.ytab + d] = c->h_rtab[L.ytab + L.h - 1];
        }
    }
    // workspace capacity check against what vo_ctx_create allocated (sized for max_w x max_h)
    const long F = c->max_frames;
    if ((size_t)(g.pyr_frame * F) > c->d_pyr.bytes || (size_t)(g.blur_frame * F) > c->d_blur.bytes ||
        (size_t)(g.cand_frame * F * 4) > c->d_key.bytes || (size_t)((long)g.ntiles * F * VO_TILE_CAP * 4) > c->d_tile_list.bytes ||
        (size_t)((long)g.nsegs * F) > c->d_seg_cnt.bytes || c->h_tiles.size() * sizeof(TileDesc) > c->d_tiles.bytes ||
        c->h_btiles.size() * sizeof(TileDesc) > c->d_btiles.bytes || c->h_rtab.size() * sizeof(ResizeCoef) > c->d_rtab.bytes)
        return vo_fail(c, VO_ERR_CAPACITY, "workspace too small for this image size");
    HIP_TRY(c, hipMemcpy(c->d_geom.p, &g, sizeof(g), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_tiles.p, c->h_tiles.data(), c->h_tiles.size() * sizeof(TileDesc), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_btiles.p, c->h_btiles.data(), c->h_btiles.size() * sizeof(TileDesc), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_rtab.p, c->h_rtab.data(), c->h_rtab.size() * sizeof(ResizeCoef), hipMemcpyHostToDevice));
    c->geom = g;
    c->geom_valid = true;
    return VO_OK;
}

// ------------------------------------------------------------------ pyramid
// level l from level l-1: H = c0x*p[i] + c1x*p[i+1] (u16), out = (c0y*H0 + c1y*H1 + 32768) >> 16.
// One lane makes 4 adjacent pixels: one 16-byte coefficient load, one 8-byte load from each of
// the two source rows (the 4 outputs read at most 7 consecutive source bytes for scale <= 2),
// one dword store.
typedef unsigned short pyr_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pyr_dot2(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_udot2(__builtin_bit_cast(pyr_u16x2, a), __builtin_bit_cast(pyr_u16x2, b), c, false);
}
#define PYR_ROWS 4   // output rows per lane: their source loads are issued together (the kernel is latency bound)
__global__ __launch_bounds__(256) void k_pyramid(const Geom* __restrict__ g, FrameSrc src, uint8_t* __restrict__ pyr,
                                                 const ResizeCoef* __restrict__ rt, int l) {
    const LevelGeom& L = g->lv[l];
    const int f = blockIdx.z;
    const int x4 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int y0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * PYR_ROWS;
    if (y0 >= L.h || x4 >= L.w) return;
    int spitch;
    const uint8_t* sp = level_ptr(g, src, pyr, f, l - 1, spitch);
    const int sw = g->lv[l - 1].w, sh = g->lv[l - 1].h;
    // coefficient tables: 32-bit offsets from a uniform base, four entries per load (the tables' padding repeats the last entry)
    const ResizeCoef* const xt = rt + L.xtab;
    const ResizeCoef* const yt = rt + L.ytab;
    const uint4 cx4 = *reinterpret_cast<const uint4*>(xt + (uint32_t)x4);
    const uint4 cy4 = *reinterpret_cast<const uint4*>(yt + (uint32_t)y0);
    const uint32_t cx[4] = {cx4.x, cx4.y, cx4.z, cx4.w};
    const int nvalid = min(4, L.w - x4);
    // first source column of the lane's 8-byte window.  At the right edge the window is pulled back inside the row: the last
    // lanes of a row then stay on the fast path (a wavefront whose edge lane took the general path below executed BOTH, and
    // that was a third of all wavefronts).  A pixel at relative position 7 has no right neighbour in the window; that only
    // happens for source column sw-1, whose coefficient pair is (256, 0), so whatever the selector picks is multiplied by 0.
    const int olast = (int)(cx[3] & 0xFFFF);
    const int o0 = min((int)(cx[0] & 0xFFFF), max(sw - 8, 0));
    uint8_t* const lbase = pyr + (long)f * g->pyr_frame + L.pyr_ofs;             // uniform
    const uint32_t dofs = (uint32_t)y0 * (uint32_t)L.pitch + (uint32_t)x4;
    static_assert(PYR_ROWS == 4, "one uint4 of row coefficients per lane");
    const uint32_t cyv[PYR_ROWS] = {cy4.x, cy4.y, cy4.z, cy4.w};
    // every pixel's two taps lie inside the window, or the only one that does not is the right tap of source column sw - 1
    if ((olast - o0 < 7 || (olast - o0 == 7 && olast == sw - 1)) && o0 + 8 <= sw) {
        uint2 a[PYR_ROWS], b[PYR_ROWS];
#pragma unroll
        for (int r = 0; r < PYR_ROWS; r++) {
            // 32-bit lane offsets from the level's (uniform) base: no 64-bit multiply-adds per load
            const uint32_t oy = cyv[r] & 0xFFFF;
            __builtin_memcpy(&a[r], sp + (__umul24(oy, (uint32_t)spitch) + (uint32_t)o0), 8);        // rows and pitches are < 2^12
            __builtin_memcpy(&b[r], sp + (__umul24(min(oy + 1u, (uint32_t)sh - 1u), (uint32_t)spitch) + (uint32_t)o0), 8);
        }
        // v_perm_b32 selectors placing source bytes (rel, rel+1) of the 8 loaded ones as a u16 pair, and the weight pair
        // c0 | c1 << 16 -- the same for all rows of the lane; H = c0*p[i] + c1*p[i+1] is then one v_dot2_u32_u16, and
        // so is c0y*H0 + c1y*H1 + 32768 on the pair H0 | H1 << 16 (H < 2^16); the pixel is byte 2 of that sum (< 2^24)
        uint32_t sel[4], cxp[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t rel = (cx[i] & 0xFFFF) - (uint32_t)o0;
            sel[i] = 0x0C010C00u + rel * 0x00010001u;
            cxp[i] = 256u + (cx[i] >> 16) * 65535u;
        }
#pragma unroll
        for (int r = 0; r < PYR_ROWS; r++) {
            const uint32_t cyp = 256u + (cyv[r] >> 16) * 65535u;
            uint32_t v[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t h0 = pyr_dot2(__builtin_amdgcn_perm(a[r].y, a[r].x, sel[i]), cxp[i], 0u);
                const uint32_t h1 = pyr_dot2(__builtin_amdgcn_perm(b[r].y, b[r].x, sel[i]), cxp[i], 0u);
                v[i] = pyr_dot2(h0 | (h1 << 16), cyp, 32768u);
            }
            // bytes 2 of v0..v3; columns past the level's width (the last group of a row) are written as 0, as before
            const uint32_t p01 = __builtin_amdgcn_perm(v[1], v[0], 0x0C0C0602u), p23 = __builtin_amdgcn_perm(v[3], v[2], 0x0C0C0602u);
            uint32_t packed = p01 | (p23 << 16);
            if (nvalid < 4) packed &= 0xFFFFFFFFu >> (8 * (4 - nvalid));
            if (y0 + r < L.h) *reinterpret_cast<uint32_t*>(lbase + (dofs + (uint32_t)r * (uint32_t)L.pitch)) = packed;
        }
    } else {
#pragma unroll
        for (int r = 0; r < PYR_ROWS; r++) {
            if (y0 + r >= L.h) break;
            const int oy = (int)(cyv[r] & 0xFFFF);
            const uint8_t* r0 = sp + (long)oy * spitch;
            const uint8_t* r1 = sp + (long)min(oy + 1, sh - 1) * spitch;
            const uint32_t c1y = cyv[r] >> 16, c0y = 256 - c1y;
            uint32_t packed = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (i < nvalid) {
                    const int ox = (int)(cx[i] & 0xFFFF), ox1 = min(ox + 1, sw - 1);
                    const uint32_t c1 = cx[i] >> 16, c0 = 256 - c1;
                    const uint32_t h0 = c0 * r0[ox] + c1 * r0[ox1], h1 = c0 * r1[ox] + c1 * r1[ox1];
                    packed |= ((c0y * h0 + c1y * h1 + 32768u) >> 16) << (8 * i);
                }
            }
            *reinterpret_cast<uint32_t*>(lbase + (dofs + (uint32_t)r * (uint32_t)L.pitch)) = packed;
        }
    }
}

// ------------------------------------------------------------------ FAST-9/16
// The segment test is evaluated in min/max form, without per-pixel bit masks:
//   darker  arc of 9 all < v-t   <=>  min over the 16 arcs of (max over the arc of p) < v - t
//   brighter arc of 9 all > v+t  <=>  max over the 16 arcs of (min over the arc of p) > v + t
// and cornerScore<16> = max(v - min_arcs max9(p), max_arcs min9(p) - v) - 1.
#define PXP 80  // LDS pixel tile pitch (18 dwords loaded; runs of 4 positions read up to column 75)
#define PXO 3   // column of the LDS tile that holds image column x0-4 (the tile is loaded from a dword boundary)
#define PXR 38  // LDS pixel tile rows: 30 outputs + 2 ring + 2*3
#define SCP 64  // score tile pitch: 64 x 32 positions, (0,0) = image (x0-1, y0-1)
// circle offsets k = 0..15 as (dx,dy): (0,3)(1,3)(2,2)(3,1)(3,0)(3,-1)(2,-2)(1,-3)(0,-3)(-1,-3)(-2,-2)(-3,-1)(-3,0)(-3,1)(-2,2)(-1,3)
#define CIRC(k) (((k) == 0 ? 3 : (k) == 1 ? 3 : (k) == 2 ? 2 : (k) == 3 ? 1 : (k) == 4 ? 0 : (k) == 5 ? -1 : (k) == 6 ? -2 : (k) == 7 ? -3 : \
                  (k) == 8 ? -3 : (k) == 9 ? -3 : (k) == 10 ? -2 : (k) == 11 ? -1 : (k) == 12 ? 0 : (k) == 13 ? 1 : (k) == 14 ? 2 : 3) * PXP + \
                 ((k) == 0 ? 0 : (k) == 1 ? 1 : (k) == 2 ? 2 : (k) == 3 ? 3 : (k) == 4 ? 3 : (k) == 5 ? 3 : (k) == 6 ? 2 : (k) == 7 ? 1 : \
                  (k) == 8 ? 0 : (k) == 9 ? -1 : (k) == 10 ? -2 : (k) == 11 ? -3 : (k) == 12 ? -3 : (k) == 13 ? -3 : (k) == 14 ? -2 : -1))

// max over the 16 arcs of 9 of (min over the arc) [BRIGHT] or min over the arcs of (max over the arc) [!BRIGHT];
// arcs of 9 as three arcs of 3, all in 3-operand min/max: 40 VALU ops per polarity
template <bool BRIGHT>
__device__ __forceinline__ int fast_arc_extreme(const uint8_t* p) {
    int x[16];
#pragma unroll
    for (int k = 0; k < 16; k++) x[k] = p[CIRC(k)];
    int e3[16];
#pragma unroll
    for (int k = 0; k < 16; k++)
        e3[k] = BRIGHT ? min(min(x[k], x[(k + 1) & 15]), x[(k + 2) & 15]) : max(max(x[k], x[(k + 1) & 15]), x[(k + 2) & 15]);
    int e9[16];
#pragma unroll
    for (int k = 0; k < 16; k++)
        e9[k] = BRIGHT ? min(min(e3[k], e3[(k + 3) & 15]), e3[(k + 6) & 15]) : max(max(e3[k], e3[(k + 3) & 15]), e3[(k + 6) & 15]);
    int best = e9[15];
#pragma unroll
    for (int k = 0; k < 14; k += 2) best = BRIGHT ? max(max(best, e9[k]), e9[k + 1]) : min(min(best, e9[k]), e9[k + 1]);
    return BRIGHT ? max(best, e9[14]) : min(best, e9[14]);
}

// byte k (0..3) of dword d
#define BYTE(d, k) (((d) >> (8 * (k))) & 0xFFu)

// per byte max(x - t, 0): the bytes as two pairs of 16-bit fields, v_pk_sub_u16 with clamp; T2 = t | t << 16
typedef unsigned short fast_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t fast_sat_sub_bytes(uint32_t x, uint32_t T2) {
    const fast_u16x2 t = __builtin_bit_cast(fast_u16x2, T2);
    const uint32_t e = __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(fast_u16x2, x & 0x00FF00FFu), t));
    const uint32_t o = __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(fast_u16x2, (x >> 8) & 0x00FF00FFu), t));
    return e | (o << 8);
}

// inclusive prefix sum over the 64 lanes of a wavefront with DPP row shifts / row broadcasts (6 VALU ops)
__device__ __forceinline__ int wave_incl_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ int lanes_below(unsigned long long m) {
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

#define FQ_CAP 4096
#ifndef VO_SELECT_SMALL_CAP
#define VO_SELECT_SMALL_CAP 3584
#endif
#ifndef VO_SELECT_BIG_LEVELS
#define VO_SELECT_BIG_LEVELS 3   // levels whose first retainBest gets the 80 KB LDS variant
#endif
__global__ __launch_bounds__(256) void k_fast_tiles(const Geom* __restrict__ g, FrameSrc src, const uint8_t* __restrict__ pyr,
                                                    const TileDesc* __restrict__ tiles, uint32_t* __restrict__ tile_list,
                                                    uint8_t* __restrict__ seg_cnt, uint32_t* __restrict__ tile_cnt, int nframes) {
    __shared__ __attribute__((aligned(16))) uint8_t px[PXR * PXP];
    __shared__ __attribute__((aligned(16))) uint8_t sc[32 * SCP + 8];      // scores of the 64 x 32 positions
    __shared__ uint32_t bm[64];          // survivors (NMS + border) of the 60 x 30 outputs, one bit each: row r = dwords 2r, 2r+1
    __shared__ int bmpre[64];            // exclusive prefix of the popcounts of bm
    __shared__ uint16_t queue[FQ_CAP];   // brighter-arc candidates from the front, darker-arc candidates from the back
    __shared__ int qcnt;                 // brighter count | darker count << 16
    __shared__ uint16_t corner[64 * 32]; // positions with a score > 0
    __shared__ int ccnt;
    const VBlock vb = xcd_block(g->ntiles, 1, nframes);
    if (!vb.valid) return;
    const int t = vb.x, f = vb.f, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const TileDesc td = tiles[t];
    const LevelGeom& L = g->lv[td.level];
    const int Lw = L.w, Lh = L.h;
    int pitch;
    const uint8_t* img = level_ptr(g, src, pyr, f, td.level, pitch);
    const int x0 = VO_EDGE + td.tx * VO_TILE_W, y0 = VO_EDGE + td.ty * VO_TILE_H;
    // 38 rows x 72 bytes from the dword boundary xa = x0-4-PXO (a multiple of 4), as 18 dwords per row;
    // 252 lanes cover 14 rows per pass
    const int xa = x0 - 4 - PXO;
    if (tid == 0) { qcnt = 0; ccnt = 0; }
    if (tid >= 192) bm[tid - 192] = 0;
    if (tid < 252) {
        const int r0 = tid / 18, k = tid - r0 * 18;
        const int x = xa + 4 * k;
        if (xa + 72 <= Lw && y0 - 4 + PXR <= Lh) {
            // interior tile (workgroup-uniform): no clamping, one address and two strides
            const uint8_t* p = img + (long)(y0 - 4 + r0) * pitch + x;
            uint32_t* q = reinterpret_cast<uint32_t*>(&px[r0 * PXP + 4 * k]);
            uint32_t v0, v1, v2 = 0;
            __builtin_memcpy(&v0, p, 4);                       // level 0 rows may be unaligned (caller's stride)
            __builtin_memcpy(&v1, p + 14L * pitch, 4);
            if (r0 + 28 < PXR) __builtin_memcpy(&v2, p + 28L * pitch, 4);
            q[0] = v0;
            q[14 * PXP / 4] = v1;
            if (r0 + 28 < PXR) q[28 * PXP / 4] = v2;
        } else {
            const bool whole = x + 3 < Lw;
#pragma unroll
            for (int pass = 0; pass < 3; pass++) {
                const int r = r0 + 14 * pass;
                if (r < PXR) {
                    const int y = min(y0 - 4 + r, Lh - 1);
                    const uint8_t* rp = img + (long)y * pitch;
                    uint32_t v;
                    if (whole) {
                        __builtin_memcpy(&v, rp + x, 4);
                    } else if (x < Lw) {
                        // the dword crosses the level's right edge: the row's last four pixels, shifted so that column x is byte 0.
                        // Columns >= Lw come out 0; no position FAST evaluates (3 px inside the level) reads them.
                        __builtin_memcpy(&v, rp + (Lw - 4), 4);
                        v >>= 8 * (x - (Lw - 4));
                    } else {
                        v = 0;
                    }
                    *reinterpret_cast<uint32_t*>(&px[r * PXP + 4 * k]) = v;
                }
            }
        }
    }
    __syncthreads();
    const int thr = g->fast_threshold;
    // Stage 1: every score position (64 x 32), 4 adjacent positions per lane and iteration, read as
    // aligned LDS dwords.  Score column c <-> LDS column c+6, score row r <-> LDS row r+3.
    // A 9-arc contains one pixel of every opposite pair, so with the four even pairs
    //   min_k max(p_k, p_k+8) > v+t   (brighter)   or   max_k min(p_k, p_k+8) < v-t   (darker)  is necessary.
    // Survivors go to a brighter and/or a darker queue so that stage 2 runs dense, polarity-uniform
    // wavefronts; one LDS atomic per wavefront and iteration reserves the queue space.
    // Score positions that can matter: an output must lie VO_EDGE inside the level (stage 3), and its 3 x 3 neighbourhood
    // reaches one position further -- image x <= Lw - VO_EDGE, y <= Lh - VO_EDGE.  (FAST itself is defined up to 3 px from
    // the border, but scores out there only feed outputs that the border filter drops.)
    static_assert(VO_EDGE >= 4, "the needed positions lie inside FAST's own 3-px border");
    const int cmax = min(VO_TILE_W + 2, Lw - VO_EDGE - x0 + 2);
    const int rmax = Lh - VO_EDGE - y0 + 2;
#ifndef VO_FAST_MINMAX_PRETEST
    const uint32_t T2 = (uint32_t)thr | ((uint32_t)thr << 16);
    // bit 7 of byte i set iff position k4 + i of this lane's run is a position of the level (the row test is made per iteration)
    const int nval = min(max(cmax - (int)((tid & 15) * 4), 0), 4);
    const uint32_t colmask = nval >= 4 ? 0x80808080u : (0x80808080u & ((1u << (8 * nval)) - 1u));
#endif
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
    for (int it = 0; it < 2; it++) {
        // 16 runs of 4 positions per row, 16 rows per iteration.  The two rows that share a half-wavefront (32 lanes = one LDS
        // pass) are 4 apart: with the 20-dword row pitch their dwords fall into disjoint halves of the 32 banks
        // (4 x 20 = 16 mod 32), where adjacent rows (offset 20) collide in 12 of 16 banks.
        const int slot = tid >> 4;               // 0..15
        const int r = 16 * it + ((slot >> 3) << 3) + ((slot & 1) << 2) + ((slot >> 1) & 3), k4 = (tid & 15) * 4;
        *reinterpret_cast<uint32_t*>(&sc[r * SCP + k4]) = 0;
        // rows at or below the level's last FAST row (bottom tiles: 9 % of all tile rows lie outside their level) have no
        // position to test; the smallest row of this wavefront's four is 16 it + {0, 2, 8, 10}[wave] (scalar branch)
        if (16 * it + ((wave_s >> 1) << 3) + ((wave_s & 1) << 1) >= rmax) continue;
        const uint32_t* rowC = reinterpret_cast<const uint32_t*>(&px[(r + 3) * PXP + k4]);
        const uint32_t* rowP2 = reinterpret_cast<const uint32_t*>(&px[(r + 5) * PXP + k4]);
        const uint32_t* rowM2 = reinterpret_cast<const uint32_t*>(&px[(r + 1) * PXP + k4]);
        const uint32_t* rowP3 = reinterpret_cast<const uint32_t*>(&px[(r + 6) * PXP + k4]);
        const uint32_t* rowM3 = reinterpret_cast<const uint32_t*>(&px[(r + 0) * PXP + k4]);
        const uint32_t C0 = rowC[0], C1 = rowC[1], C2 = rowC[2], C3 = rowC[3];
        const uint32_t P21 = rowP2[1], P22 = rowP2[2], M21 = rowM2[1], M22 = rowM2[2];
        const uint32_t P31 = rowP3[1], P32 = rowP3[2], M31 = rowM3[1], M32 = rowM3[2];
#ifdef VO_FAST_MINMAX_PRETEST
        // position i: centre = LDS col k4+6+i;  mm = min over pairs of max, xm = max over pairs of min
#define PRETEST(i, v, a0, a8, a2, a10, a4, a12, a6, a14)                                                             \
        const int mm##i = (int)min(min(max(a0, a8), max(a2, a10)), min(max(a4, a12), max(a6, a14)));                  \
        const int xm##i = (int)max(max(min(a0, a8), min(a2, a10)), max(min(a4, a12), min(a6, a14)));                  \
        const int th##i = (r < rmax && k4 + i < cmax) ? thr : 4096;                                                   \
        const bool pb##i = mm##i > (int)(v) + th##i, pd##i = xm##i < (int)(v) - th##i;                                  \
        const unsigned long long mb##i = __ballot(pb##i), md##i = __ballot(pd##i);
        PRETEST(0, BYTE(C1, 2), BYTE(P31, 2), BYTE(M31, 2), BYTE(P22, 0), BYTE(M21, 0), BYTE(C2, 1), BYTE(C0, 3), BYTE(M22, 0), BYTE(P21, 0))
        PRETEST(1, BYTE(C1, 3), BYTE(P31, 3), BYTE(M31, 3), BYTE(P22, 1), BYTE(M21, 1), BYTE(C2, 2), BYTE(C1, 0), BYTE(M22, 1), BYTE(P21, 1))
        PRETEST(2, BYTE(C2, 0), BYTE(P32, 0), BYTE(M32, 0), BYTE(P22, 2), BYTE(M21, 2), BYTE(C2, 3), BYTE(C1, 1), BYTE(M22, 2), BYTE(P21, 2))
        PRETEST(3, BYTE(C2, 1), BYTE(P32, 1), BYTE(M32, 1), BYTE(P22, 3), BYTE(M21, 3), BYTE(C3, 0), BYTE(C1, 2), BYTE(M22, 3), BYTE(P21, 3))
#undef PRETEST
#else
        // Byte-parallel form of the same test (round 3): the four positions of the lane are the four bytes of a dword, and
        // v_lerp_u8 -- per byte (a + b + c) >> 1 on a 9-bit sum -- is a per-byte unsigned compare: bit 7 of lerp(a, ~b, 0) is
        // a > b, bit 7 of lerp(a, ~b, 1) is a >= b.  Per pixel window (the circle pixel k of all four positions, one aligned
        // dword or one v_alignbit of two) that is ONE instruction for four "p_k > v + t" and one for four "p_k >= v - t",
        // where the min / max form spent 12 SDWA min / max per POSITION; the pair logic is plain and / or on the bit-7 lanes.
        // Thresholds per byte: nvt = 255 - min(v + t, 255) and nvmt = 255 - max(v - t, 0) (saturating 16-bit-field
        // subtractions, once per centre dword), so "never" cases (v + t > 255, v - t < 0) come out false by themselves.
        uint32_t Rb, Rd;
        {
            const uint32_t V = __builtin_amdgcn_alignbit(C2, C1, 16);
            const uint32_t nV = ~V;
            const uint32_t nvt = fast_sat_sub_bytes(nV, T2);
            const uint32_t nvmt = ~fast_sat_sub_bytes(V, T2);
            const uint32_t a0 = __builtin_amdgcn_alignbit(P32, P31, 16), a8 = __builtin_amdgcn_alignbit(M32, M31, 16);
            const uint32_t a4 = __builtin_amdgcn_alignbit(C3, C2, 8), a12 = __builtin_amdgcn_alignbit(C1, C0, 24);
            const uint32_t a2 = P22, a14 = P21, a6 = M22, a10 = M21;
#define BR(a) __builtin_amdgcn_lerp((a), nvt, 0u)              /* bit 7: a > v + t */
#define NL(a) __builtin_amdgcn_lerp((a), nvmt, 0x01010101u)    /* bit 7: a >= v - t, i.e. NOT darker */
            const uint32_t vm = r < rmax ? colmask : 0u;
            // (adding the two odd pairs (1, 9), (7, 15), whose pixels lie in the row +-3 dwords loaded anyway, measured SLOWER: 161
            // against 155 us per launch -- 16 more VALU instructions per run of 4, and only a few per cent fewer survivors)
            Rb = (BR(a0) | BR(a8)) & (BR(a2) | BR(a10)) & (BR(a4) | BR(a12)) & (BR(a6) | BR(a14)) & vm;
            Rd = ~((NL(a0) & NL(a8)) | (NL(a2) & NL(a10)) | (NL(a4) & NL(a12)) | (NL(a6) & NL(a14))) & vm;
#undef BR
#undef NL
        }
        // opaque to the optimiser from here: otherwise every use of a position's predicate (its ballot, its queue write) is
        // re-derived from the pair terms with a bit-select of its own
        asm volatile("" : "+v"(Rb), "+v"(Rd));
        const bool pb0 = BYTE(Rb, 0) != 0, pb1 = BYTE(Rb, 1) != 0, pb2 = BYTE(Rb, 2) != 0, pb3 = BYTE(Rb, 3) != 0;
        const bool pd0 = BYTE(Rd, 0) != 0, pd1 = BYTE(Rd, 1) != 0, pd2 = BYTE(Rd, 2) != 0, pd3 = BYTE(Rd, 3) != 0;
        const unsigned long long mb0 = __ballot(pb0), mb1 = __ballot(pb1), mb2 = __ballot(pb2), mb3 = __ballot(pb3);
        const unsigned long long md0 = __ballot(pd0), md1 = __ballot(pd1), md2 = __ballot(pd2), md3 = __ballot(pd3);
#endif
        const int nb0 = (int)__popcll(mb0), nb1 = (int)__popcll(mb1), nb2 = (int)__popcll(mb2), nb3 = (int)__popcll(mb3);
        const int nd0 = (int)__popcll(md0), nd1 = (int)__popcll(md1), nd2 = (int)__popcll(md2), nd3 = (int)__popcll(md3);
        const int add = (nb0 + nb1 + nb2 + nb3) | ((nd0 + nd1 + nd2 + nd3) << 16);
        int base = 0;
        if (lane == 0) base = atomicAdd(&qcnt, add);
        base = __builtin_amdgcn_readfirstlane(base);
        const int ob = base & 0xFFFF, od = FQ_CAP - 1 - (base >> 16);
        const int pos = r * 64 + k4;
        if (pb0) queue[ob + lanes_below(mb0)] = (uint16_t)pos;
        if (pb1) queue[ob + nb0 + lanes_below(mb1)] = (uint16_t)(pos + 1);
        if (pb2) queue[ob + nb0 + nb1 + lanes_below(mb2)] = (uint16_t)(pos + 2);
        if (pb3) queue[ob + nb0 + nb1 + nb2 + lanes_below(mb3)] = (uint16_t)(pos + 3);
        if (pd0) queue[od - lanes_below(md0)] = (uint16_t)pos;
        if (pd1) queue[od - nd0 - lanes_below(md1)] = (uint16_t)(pos + 1);
        if (pd2) queue[od - nd0 - nd1 - lanes_below(md2)] = (uint16_t)(pos + 2);
        if (pd3) queue[od - nd0 - nd1 - nd2 - lanes_below(md3)] = (uint16_t)(pos + 3);
    }
    __syncthreads();
    // Stage 2: arc extremes + cornerScore of the queued positions, both queues in one pass: the darker queue starts at
    // the next multiple of 64 after the brighter one, so every wavefront is polarity-uniform.  A position cannot have a
    // brighter AND a darker arc of 9 (two arcs of 9 on a circle of 16 overlap), so at most one of its two visits finds
    // a score > 0 and stores it (the score tile was zeroed in stage 1).  Corners are pushed to a third list.
    const int nb = qcnt & 0xFFFF, nd = qcnt >> 16;
    const int nb64 = (nb + 63) & ~63;
    for (int q = tid; q < nb64 + nd; q += 256) {
        int i = -1, val = 0;
        if (q < nb64) {
            if (q < nb) {
                i = queue[q];
                const uint8_t* p = &px[((i >> 6) + 3) * PXP + (i & 63) + 3 + PXO];
                const int best = fast_arc_extreme<true>(p) - (int)p[0];
                val = best > thr ? best - 1 : 0;
            }
        } else {
            i = queue[FQ_CAP - 1 - (q - nb64)];
            const uint8_t* p = &px[((i >> 6) + 3) * PXP + (i & 63) + 3 + PXO];
            const int best = (int)p[0] - fast_arc_extreme<false>(p);
            val = best > thr ? best - 1 : 0;
        }
        const unsigned long long m = __ballot(val > 0);
        if (m) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&ccnt, (int)__popcll(m));
            base = __builtin_amdgcn_readfirstlane(base);
            if (val > 0) {
                sc[i] = (uint8_t)val;
                corner[base + lanes_below(m)] = (uint16_t)i;
            }
        }
    }
    __syncthreads();
    // Stage 3: strict 3x3 non-max suppression + border filter at the corners; a survivor sets its bit in a 60 x 30
    // bitmap (row r = dwords 2r, 2r+1), whose bit order is raster order
    const int nc = ccnt;
    for (int q = tid; q < nc; q += 256) {
        const int i = corner[q];
        const int r = i >> 6, c = i & 63;                      // score position; output (r-1, c-1)
        const uint8_t* s0 = &sc[i];
        const int xi = x0 - 1 + c, yi = y0 - 1 + r;
        const int s = s0[0];
        // (nine byte reads: reading the 3 x 3 neighbourhood as three UNALIGNED dwords measured slower, 161 against 156 us per
        // launch -- misaligned LDS accesses stall the pipe)
        if (r >= 1 && r <= VO_TILE_H && c >= 1 && c <= VO_TILE_W && xi < Lw - VO_EDGE && yi < Lh - VO_EDGE &&
            s > s0[-1] && s > s0[1] && s > s0[-SCP - 1] && s > s0[-SCP] && s > s0[-SCP + 1] && s > s0[SCP - 1] && s > s0[SCP] && s > s0[SCP + 1])
            atomicOr(&bm[(r - 1) * 2 + ((c - 1) >> 5)], 1u << ((c - 1) & 31));
    }
    __syncthreads();
    // Stage 4: ranks from the bitmap.  The first wavefront scans the 60 dword popcounts (exclusive prefix per
    // dword, count per row); then every survivor's rank = prefix of its dword + set bits below it.
    if (wave == 0) {
        const int v = lane < 2 * VO_TILE_H ? (int)__popc(bm[lane]) : 0;
        const int incl = wave_incl_scan(v);
        bmpre[lane] = incl - v;
        if (lane == 63) tile_cnt[(long)f * g->ntiles + t] = (uint32_t)incl;   // survivors of the tile: k_fast_emit sums these over the earlier tile rows
        const int nxt = __shfl_down(v, 1);
        const int r = lane >> 1;
        if ((lane & 1) == 0 && r < VO_TILE_H && td.ty * VO_TILE_H + r < L.ih)
            seg_cnt[(long)f * g->nsegs + L.seg_base + (long)(td.ty * VO_TILE_H + r) * L.ntx + td.tx] = (uint8_t)(v + nxt);
    }
    __syncthreads();
    uint32_t* tl = tile_list + ((long)f * g->ntiles + t) * VO_TILE_CAP;
    for (int q = tid; q < nc; q += 256) {
        const int i = corner[q];
        const int r = (i >> 6) - 1, c = (i & 63) - 1;          // output position
        if (r < 0 || r >= VO_TILE_H || c < 0 || c >= VO_TILE_W) continue;
        const int d = r * 2 + (c >> 5), bit = c & 31;
        const uint32_t w = bm[d];
        if ((w >> bit) & 1u) tl[bmpre[d] + (int)__popc(w & ((1u << bit) - 1u))] = PACK_XYS(x0 + c, y0 + r, sc[i]);
    }
}

// tile lists -> per-level candidate lists in raster order: key = FAST score, payload = packed x,y,score.
// A candidate's slot is the number of candidates before it in raster order.  One workgroup takes one ROW OF TILES
// (up to 30 image rows x ntx tile columns): the slot of entry e of tile x in image row r is
//   (all candidates of the tile rows above: sum of their tiles' totals)
//   + (rows above r in this tile row, all tile columns) + (row r, tile columns left of x)      -> rowbase[r] + rowl[r][x]
//   + (e - entries of tile x in the rows above r)                                                -> e - colpre[x][r]
// all from the tile row's block of per-(row, tile) counts (rows * ntx bytes, contiguous) that k_fast_tiles wrote, prefixed
// along both axes in LDS.  (Round 1 ran a scan kernel per level between k_fast_tiles and a per-tile emit kernel.)
// The workgroup of a level's last tile row also writes the level's total.
#define EM_MAXX 72   // tile columns: (4095 - 62) / 60 rounded up is 68
static_assert((4095 - 2 * VO_EDGE + VO_TILE_W - 1) / VO_TILE_W <= EM_MAXX && VO_TILE_H <= 32 && VO_TILE_CAP < 65536, "k_fast_emit LDS tables (vo_build_geometry accepts widths up to 4095)");
__global__ __launch_bounds__(256) void k_fast_emit(const Geom* __restrict__ g, const uint32_t* __restrict__ tile_list,
                                                   const uint8_t* __restrict__ seg_cnt, const uint32_t* __restrict__ tile_cnt,
                                                   int* __restrict__ cnt_fast, float* __restrict__ key, uint32_t* __restrict__ payload,
                                                   int ntrows, int nframes) {
    __shared__ uint8_t A[VO_TILE_H * EM_MAXX];         // counts [r][x], pitch ntx
    __shared__ uint16_t rowl[VO_TILE_H * EM_MAXX];     // candidates of row r left of tile x
    __shared__ uint16_t colpre[EM_MAXX * 32];          // candidates of tile x above row r
    __shared__ int rowtot[32], rowbase[32], tiletot[EM_MAXX + 1], tilepre[EM_MAXX + 1], wsum[4];
    const VBlock vb = xcd_block(ntrows, 1, nframes);
    if (!vb.valid) return;
    const int f = vb.f, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int l = 0, ty = vb.x;
    while (l < g->nlevels - 1 && ty >= g->lv[l].nty) { ty -= g->lv[l].nty; l++; }
    const LevelGeom& L = g->lv[l];
    const int ntx = L.ntx;
    if (vb.x == 0 && tid < g->nlevels && g->lv[tid].ntx * g->lv[tid].nty == 0) cnt_fast[f * VO_MAX_LEVELS + tid] = 0;   // levels without tiles
    const int rows = min(VO_TILE_H, L.ih - ty * VO_TILE_H);
    const int nbytes = rows * ntx;
    const uint8_t* blk = seg_cnt + (long)f * g->nsegs + L.seg_base + (long)ty * VO_TILE_H * ntx;
    for (int i = tid; i < nbytes; i += 256) A[i] = blk[i];
    int above = 0;
    {
        const uint32_t* tc = tile_cnt + (long)f * g->ntiles + L.tile_base;
        for (int i = tid; i < ty * ntx; i += 256) above += (int)tc[i];
    }
    above = wave_incl_scan(above);
    if (lane == 63) wsum[wave] = above;
    __syncthreads();
    above = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    // prefixes along the two axes: lanes 0..rows-1 of wavefront 0 walk their row, lanes of wavefronts 1.. walk a tile column
    if (tid < rows) {
        int acc = 0;
        for (int x = 0; x < ntx; x++) { rowl[tid * ntx + x] = (uint16_t)acc; acc += A[tid * ntx + x]; }
        rowtot[tid] = acc;
    } else if (tid >= 64 && tid - 64 < ntx) {
        const int x = tid - 64;
        int acc = 0;
        for (int r = 0; r < rows; r++) { colpre[x * 32 + r] = (uint16_t)acc; acc += A[r * ntx + x]; }
        tiletot[x] = acc;
    }
    __syncthreads();
    if (wave == 0) {
        const int v = lane < rows ? rowtot[lane] : 0;
        const int incl = wave_incl_scan(v);
        if (lane < rows) rowbase[lane] = above + incl - v;
        if (lane == 63 && ty == L.nty - 1) cnt_fast[f * VO_MAX_LEVELS + l] = above + incl;
    } else if (wave == 1) {
        int carry = 0;
        for (int x0 = 0; x0 < ntx; x0 += 64) {
            const int x = x0 + lane;
            const int v = x < ntx ? tiletot[x] : 0;
            const int incl = wave_incl_scan(v);
            if (x < ntx) tilepre[x] = carry + incl - v;
            carry += __builtin_amdgcn_readlane(incl, 63);
        }
        if (lane == 0) tilepre[ntx] = carry;
    }
    __syncthreads();
    const int total = tilepre[ntx];
    const uint32_t* tl0 = tile_list + ((long)f * g->ntiles + L.tile_base + (long)ty * ntx) * VO_TILE_CAP;
    const long cb = (long)f * g->cand_frame + L.cand_ofs;
    for (int i = tid; i < total; i += 256) {
        int lo = 0, hi = ntx - 1;                              // the tile whose run of entries holds i
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (tilepre[mid] <= i) lo = mid; else hi = mid - 1;
        }
        const int x = lo, e = i - tilepre[x];
        const uint32_t p = tl0[(long)x * VO_TILE_CAP + e];
        const int r = UNPACK_Y(p) - (VO_EDGE + ty * VO_TILE_H);   // a tile's list is in raster order: the row is in the record
        const long o = cb + rowbase[r] + rowl[r * ntx + x] + (e - colpre[x * 32 + r]);
        key[o] = (float)UNPACK_S(p);
        payload[o] = p;
    }
}

// ------------------------------------------------------------------ retainBest
// Sequential pieces of libstdc++ (bits/stl_algo.h, bits/stl_heap.h) on (key, payload) pairs;
// comp(a, b) = key[a] > key[b].  Executed by one lane.
template <typename PT>
__device__ inline void sel_swap(float* key, PT* pay, int i, int j) {
    float k = key[i]; key[i] = key[j]; key[j] = k;
    PT p = pay[i]; pay[i] = pay[j]; pay[j] = p;
}
template <typename PT>
__device__ inline void sel_median_to_first(float* key, PT* pay, int result, int a, int b, int c) {
    float ka = key[a], kb = key[b], kc = key[c];
    if (ka > kb) {
        if (kb > kc) sel_swap(key, pay, result, b);
        else if (ka > kc) sel_swap(key, pay, result, c);
        else sel_swap(key, pay, result, a);
    } else if (ka > kc) sel_swap(key, pay, result, a);
    else if (kb > kc) sel_swap(key, pay, result, c);
    else sel_swap(key, pay, result, b);
}
template <typename PT>
__device__ inline void sel_insertion_sort(float* key, PT* pay, int first, int last) {
    for (int i = first + 1; i < last; i++)
        for (int j = i; j > first && key[j] > key[j - 1]; j--) sel_swap(key, pay, j, j - 1);
}
template <typename PT>
__device__ inline void sel_adjust_heap(float* key, PT* pay, int first, int hole, int len, float vk, PT vp) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (key[first + child] > key[first + child - 1]) child--;
        key[first + hole] = key[first + child]; pay[first + hole] = pay[first + child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        key[first + hole] = key[first + child - 1]; pay[first + hole] = pay[first + child - 1];
        hole = child - 1;
    }
    int parent = (hole - 1) / 2;
    while (hole > top && key[first + parent] > vk) {
        key[first + hole] = key[first + parent]; pay[first + hole] = pay[first + parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    key[first + hole] = vk; pay[first + hole] = vp;
}
template <typename PT>
__device__ inline void sel_heap_select(float* key, PT* pay, int first, int nth, int last) {
    const int middle = nth + 1, len = middle - first;
    if (len >= 2) {
        for (int parent = (len - 2) / 2;; parent--) {
            sel_adjust_heap(key, pay, first, parent, len, key[first + parent], pay[first + parent]);
            if (parent == 0) break;
        }
    }
    for (int i = middle; i < last; i++)
        if (key[i] > key[first]) {
            float vk = key[i]; PT vp = pay[i];
            key[i] = key[first]; pay[i] = pay[first];
            sel_adjust_heap(key, pay, first, 0, len, vk, vp);
        }
    sel_swap(key, pay, first, nth);
}

// Hoare partition of [lo,hi) resolved in closed form (tests/introselect_model.py).
// MODE 0: __unguarded_partition around pivot p: left stoppers !(k > p), right stoppers !(p > k);
//         returns the cut.
// MODE 1: std::partition with pred k >= p: left stoppers !(k >= p), right stoppers (k >= p);
//         returns the number of elements satisfying pred.
// With L the left stoppers in ascending and R the right stoppers in descending position order, the scan swaps the
// pairs (L[k], R[k]) while L[k] < R[k]; L[k] - R[k] increases with k, so "L[k] < R[k]" holds exactly for k < K and
// every pair can decide for itself.  One pass classifies and writes each wavefront's stoppers, in order, to the
// wavefront's own segment of lpos/rpos (positions relative to lo); L[k] / R[k] are then looked up through the
// per-wavefront counts.  Three barriers per partition.
template <int NT, int MODE, typename PT, typename QT>
__device__ int sel_partition(float* key, PT* pay, QT* lpos, QT* rpos, int lo, int hi, float p, int* s_wl, int* s_wr, int* s_K) {
    constexpr int NW = NT / 64;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int len = hi - lo;
    const int chunk = (((len + NW - 1) / NW) + 63) & ~63;
    const int wlo = lo + w * chunk, whi = min(wlo + chunk, hi);
    const unsigned long long lt = (1ull << lane) - 1ull;
    if (tid == 0) *s_K = 0;
    int cl = 0, cr = 0;
    const int seg = w * chunk;
    for (int b = wlo; b < whi; b += 64) {
        const int i = b + lane;
        const bool in = i < whi;
        const float k = in ? key[i] : 0.f;
        const bool isL = in && (MODE == 0 ? !(k > p) : !(k >= p));
        const bool isR = in && (MODE == 0 ? !(p > k) : (k >= p));
        const unsigned long long mL = __ballot(isL), mR = __ballot(isR);
        if (isL) lpos[seg + cl + (int)__popcll(mL & lt)] = (QT)(i - lo);
        if (isR) rpos[seg + cr + (int)__popcll(mR & lt)] = (QT)(i - lo);
        cl += (int)__popcll(mL);
        cr += (int)__popcll(mR);
    }
    if (lane == 0) { s_wl[w] = cl; s_wr[w] = cr; }
    __syncthreads();
    int baseL[NW + 1], baseR[NW + 1];
    baseL[0] = baseR[0] = 0;
#pragma unroll
    for (int w2 = 0; w2 < NW; w2++) { baseL[w2 + 1] = baseL[w2] + s_wl[w2]; baseR[w2 + 1] = baseR[w2] + s_wr[w2]; }
    const int totL = baseL[NW], totR = baseR[NW];
    auto Lat = [&](int k) -> int {          // k-th left stopper, ascending
        int w2 = 0, bs = 0;
#pragma unroll
        for (int q = 1; q < NW; q++) { const bool ge = k >= baseL[q]; w2 += ge ? 1 : 0; bs = ge ? baseL[q] : bs; }
        return (int)lpos[w2 * chunk + (k - bs)];
    };
    auto Rat = [&](int k) -> int {          // k-th right stopper, descending
        const int j = totR - 1 - k;
        int w2 = 0, bs = 0;
#pragma unroll
        for (int q = 1; q < NW; q++) { const bool ge = j >= baseR[q]; w2 += ge ? 1 : 0; bs = ge ? baseR[q] : bs; }
        return (int)rpos[w2 * chunk + (j - bs)];
    };
    const int m = min(totL, totR);
    int ck = 0;
    for (int k = tid; k < m; k += NT) {
        const int a = Lat(k), b = Rat(k);
        if (a < b) { sel_swap(key, pay, lo + a, lo + b); ck++; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ck += __shfl_xor(ck, o);
    if (lane == 0 && ck) atomicAdd(s_K, ck);
    __syncthreads();
    const int K = *s_K;
    int ret;
    if (MODE == 0) {
        if (K == 0) ret = lo + Lat(0);
        else {
            ret = lo + Rat(K - 1);
            if (K < totL) ret = min(ret, lo + Lat(K));
        }
    } else {
        ret = totR;
    }
    __syncthreads();
    return ret;
}

// The tail of introselect on a range of at most 64 elements, by ONE wavefront with the elements in registers
// (lane i = position first + i): every remaining median-of-3 + Hoare partition step is shuffles, ballots and a
// 64-entry LDS table of stopper positions -- no barriers and no memory round trips, where the block-wide steps
// spend ~7 us each regardless of the range length.  Same closed form as sel_partition.  Returns false if the
// depth limit was hit (the caller then runs the heap-select fallback on memory).
template <typename PT>
__device__ bool sel_small_wave(float* key, PT* pay, int& first, int& last, int nth, int& depth, int* tabL, int* tabR) {
    const int lane = threadIdx.x & 63;
    const int base = first, len = last - first;
    float k = lane < len ? key[base + lane] : 0.f;
    PT p = lane < len ? pay[base + lane] : (PT)0;
    int lo = 0, hi = len;
    const int nthr = nth - base;
    const unsigned long long below = (1ull << lane) - 1ull, above = lane == 63 ? 0ull : ~((2ull << lane) - 1ull);
    bool ok = true;
    while (hi - lo > 3) {
        if (depth == 0) { ok = false; break; }
        --depth;
        // __move_median_to_first(result = lo, a = lo+1, b = lo + (hi-lo)/2, c = hi-1)
        const int a = lo + 1, b = lo + (hi - lo) / 2, c = hi - 1;
        const float ka = __shfl(k, a), kb = __shfl(k, b), kc = __shfl(k, c);
        int m;
        if (ka > kb) m = (kb > kc) ? b : ((ka > kc) ? c : a);
        else m = (ka > kc) ? a : ((kb > kc) ? c : b);
        {
            const float k_lo = __shfl(k, lo), k_m = __shfl(k, m);
            const PT p_lo = __shfl(p, lo), p_m = __shfl(p, m);
            if (lane == lo) { k = k_m; p = p_m; }
            else if (lane == m) { k = k_lo; p = p_lo; }
        }
        const float pv = __shfl(k, lo);
        // __unguarded_partition over [lo+1, hi)
        const bool in = lane >= lo + 1 && lane < hi;
        const bool isL = in && !(k > pv), isR = in && !(pv > k);
        const unsigned long long mL = __ballot(isL), mR = __ballot(isR);
        const int totL = (int)__popcll(mL), totR = (int)__popcll(mR);
        const int rankL = (int)__popcll(mL & below), rankR = (int)__popcll(mR & above);
        if (isL) tabL[rankL] = lane;       // ascending positions
        if (isR) tabR[rankR] = lane;       // descending positions
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int mn = min(totL, totR);
        const bool cpair = lane < mn && tabL[lane] < tabR[lane];
        const int K = (int)__popcll(__ballot(cpair));
        int partner = -1;
        if (isL && rankL < K) partner = tabR[rankL];
        else if (isR && rankR < K) partner = tabL[rankR];
        const float k2 = __shfl(k, partner < 0 ? lane : partner);
        const PT p2 = __shfl(p, partner < 0 ? lane : partner);
        if (partner >= 0) { k = k2; p = p2; }
        int cut;
        if (K == 0) cut = tabL[0];
        else {
            cut = tabR[K - 1];
            if (K < totL) cut = min(cut, tabL[K]);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (cut <= nthr) lo = cut;
        else hi = cut;
    }
    if (ok) {
        // __insertion_sort of the <= 3 survivors, replayed on wave-uniform copies
        const int cnt = hi - lo;
        float e[3]; PT q[3];
#pragma unroll
        for (int t = 0; t < 3; t++) { e[t] = __shfl(k, min(lo + t, 63)); q[t] = __shfl(p, min(lo + t, 63)); }
        for (int i = 1; i < cnt; i++)
            for (int j = i; j > 0 && e[j] > e[j - 1]; j--) {
                const float te = e[j]; e[j] = e[j - 1]; e[j - 1] = te;
                const PT tq = q[j]; q[j] = q[j - 1]; q[j - 1] = tq;
            }
#pragma unroll
        for (int t = 0; t < 3; t++)
            if (t < cnt && lane == lo + t) { k = e[t]; p = q[t]; }
    }
    if (lane < len) { key[base + lane] = k; pay[base + lane] = p; }
    first = base + lo;
    last = base + hi;
    return ok;
}

// KeyPointsFilter::retainBest(list, mult*quota) for every (frame, level) list of the chunk.
// The replay is a chain of ~log2(n) partitions, each a few dependent passes over the list plus barriers, so it is
// bound by latency, not throughput.  Lists of up to CAP candidates are copied to LDS (keys + a 16-bit permutation),
// permuted there by the same code and only the kept prefix is written back; longer lists run in place in global
// memory.  CAP is chosen so that TWO workgroups fit a CU (10 B per candidate: CAP 8000 = 80 KB for the first
// retainBest, whose lists reach ~6000 FAST candidates per level; CAP 2048 for the second): with one workgroup per
// CU the launch was slower than the global-memory version.  Per 64-frame launch: 0.089 -> 0.054 ms (first),
// 0.056 -> 0.035 ms (second).
template <int NT, typename PT, typename QT>
__device__ int retain_best(float* key, PT* pay, QT* lpos, QT* rpos, int n, int np, int* s_wl, int* s_wr, int* s_K, int* tabL, int* tabR) {
    const int tid = threadIdx.x;
    int first = 0, last = n;
    const int nth = np - 1;
    int depth = 2 * (31 - __clz(n));
    bool heap_done = false, sorted = false;
    while (last - first > 3) {
        if (last - first <= 64) {
            // every wavefront replays the tail redundantly on its own registers would race on memory: wavefront 0 does it,
            // the range bookkeeping (first, last, depth) is recomputed identically by all from what it leaves in LDS
            if (tid < 64) {
                const bool ok = sel_small_wave(key, pay, first, last, nth, depth, tabL, tabR);
                if (tid == 0) { s_wl[0] = first; s_wl[1] = last; s_wl[2] = depth; s_wl[3] = ok ? 1 : 0; }
            }
            __syncthreads();
            first = s_wl[0]; last = s_wl[1]; depth = s_wl[2];
            const bool ok = s_wl[3] != 0;
            __syncthreads();
            if (ok) { sorted = true; break; }
            // depth limit inside the tail: fall through to the heap-select branch below
        }
        if (depth == 0) {
            if (tid == 0) sel_heap_select(key, pay, first, nth, last);
            __syncthreads();
            heap_done = true;
            break;
        }
        --depth;
        if (tid == 0) sel_median_to_first(key, pay, first, first + 1, first + (last - first) / 2, last - 1);
        __syncthreads();
        const float p = key[first];
        const int cut = sel_partition<NT, 0>(key, pay, lpos, rpos, first + 1, last, p, s_wl, s_wr, s_K);
        if (cut <= nth) first = cut;
        else last = cut;
    }
    if (!heap_done && !sorted && tid == 0) sel_insertion_sort(key, pay, first, last);
    __syncthreads();
    const float amb = key[np - 1];
    const int totT = sel_partition<NT, 1>(key, pay, lpos, rpos, np, n, amb, s_wl, s_wr, s_K);
    return np + totT;
}

// LDS layout for lists of up to CAP <= 65536 candidates: keys (4 B), the permutation as 16-bit original indices
// and 16-bit stopper positions (relative to the partition range): 10 B per candidate, so CAP 8192 = 80 KB and two
// workgroups fit a CU.  The payloads are gathered through the permutation at the end (staged in the stopper arrays).
template <int NT, int CAP>
__global__ __launch_bounds__(NT) void k_select(const Geom* __restrict__ g, float* key, uint32_t* pay, uint32_t* lpos, uint32_t* rpos,
                                               const int* __restrict__ cnt_in, int* __restrict__ cnt_out, int mult, int l0) {
    __shared__ int s_wl[NT / 64 < 4 ? 4 : NT / 64], s_wr[NT / 64], s_K, s_tabL[64], s_tabR[64];
    __shared__ float l_key[CAP];
    __shared__ uint16_t l_idx[CAP];
    __shared__ __attribute__((aligned(4))) uint16_t l_pos[2 * CAP];
    const int l = l0 + blockIdx.x, f = blockIdx.y, tid = threadIdx.x;
    const LevelGeom& L = g->lv[l];
    const long base = (long)f * g->cand_frame + L.cand_ofs;
    key += base; pay += base; lpos += base; rpos += base;
    const int n = cnt_in[f * VO_MAX_LEVELS + l];
    const int np = mult * L.quota;
    if (!(n > np)) {
        if (tid == 0) cnt_out[f * VO_MAX_LEVELS + l] = n;
        return;
    }
    if (np == 0) {
        if (tid == 0) cnt_out[f * VO_MAX_LEVELS + l] = 0;
        return;
    }
    int kept;
    if (n <= CAP) {
        for (int i = tid; i < n; i += NT) { l_key[i] = key[i]; l_idx[i] = (uint16_t)i; }
        __syncthreads();
        kept = retain_best<NT>(l_key, l_idx, l_pos, l_pos + CAP, n, np, s_wl, s_wr, &s_K, s_tabL, s_tabR);
        uint32_t* stage = reinterpret_cast<uint32_t*>(l_pos);       // kept <= n <= CAP dwords
        for (int i = tid; i < kept; i += NT) stage[i] = pay[l_idx[i]];
        __syncthreads();
        for (int i = tid; i < kept; i += NT) { key[i] = l_key[i]; pay[i] = stage[i]; }
    } else {
        kept = retain_best<NT>(key, pay, lpos, rpos, n, np, s_wl, s_wr, &s_K, s_tabL, s_tabR);
    }
    if (tid == 0) cnt_out[f * VO_MAX_LEVELS + l] = kept;
}

// ------------------------------------------------------------------ Harris (HarrisResponses, blockSize 7)
#define HARRIS_BX 8    // workgroups per (level, frame)
__global__ __launch_bounds__(256) void k_harris(const Geom* __restrict__ g, FrameSrc src, const uint8_t* __restrict__ pyr,
                                                const int* __restrict__ cnt, float* __restrict__ key, const uint32_t* __restrict__ payload, int nframes) {
    const VBlock vb = xcd_block(HARRIS_BX, g->nlevels, nframes);
    if (!vb.valid) return;
    const int l = vb.y, f = vb.f;
    const LevelGeom& L = g->lv[l];
    const int n = cnt[f * VO_MAX_LEVELS + l];
    int pitch;
    const uint8_t* img = level_ptr(g, src, pyr, f, l, pitch);
    const long cb = (long)f * g->cand_frame + L.cand_ofs;
    const float scale = 1.f / ((1 << 2) * 7 * 255.f);
    const float scale_sq_sq = scale * scale * scale * scale;
    for (int i = vb.x * 256 + threadIdx.x; i < n; i += HARRIS_BX * 256) {
        const uint32_t p = payload[cb + i];
        const uint8_t* c0 = img + (long)(UNPACK_Y(p) - 4) * pitch + (UNPACK_X(p) - 4);
        // 9x9 window, each row as three (unaligned) dwords; Ix, Iy at the inner 7x7
        int a = 0, b = 0, c = 0;
        uint8_t r0[12], r1[12], r2[12];
        __builtin_memcpy(r0, c0, 12);
        __builtin_memcpy(r1, c0 + pitch, 12);
#pragma unroll
        for (int y = 1; y <= 7; y++) {
            __builtin_memcpy(r2, c0 + (long)(y + 1) * pitch, 12);
#pragma unroll
            for (int x = 1; x <= 7; x++) {
                int Ix = ((int)r1[x + 1] - (int)r1[x - 1]) * 2 + ((int)r0[x + 1] - (int)r0[x - 1]) + ((int)r2[x + 1] - (int)r2[x - 1]);
                int Iy = ((int)r2[x] - (int)r0[x]) * 2 + ((int)r2[x - 1] - (int)r0[x - 1]) + ((int)r2[x + 1] - (int)r0[x + 1]);
                a += Ix * Ix;
                b += Iy * Iy;
                c += Ix * Iy;
            }
#pragma unroll
            for (int j = 0; j < 12; j++) { r0[j] = r1[j]; r1[j] = r2[j]; }
        }
        const float fa = (float)a, fb = (float)b, fc = (float)c;
        const float t1 = fa * fb;
        const float t2 = fc * fc;
        const float s = fa + fb;
        const float t4 = 0.04f * s;
        const float t5 = t4 * s;
        key[cb + i] = ((t1 - t2) - t5) * scale_sq_sq;
    }
}

// ------------------------------------------------------------------ orientation + KeyPoint records
__device__ __forceinline__ float fast_atan2_deg(float y, float x, bool use_fma) {
    const float p1 = 0.9997878412794807f * (float)(180 / M_PI);
    const float p3 = -0.3258083974640975f * (float)(180 / M_PI);
    const float p5 = 0.1555786518463281f * (float)(180 / M_PI);
    const float p7 = -0.04432655554792128f * (float)(180 / M_PI);
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        if (use_fma) a = __fmaf_rn(__fmaf_rn(__fmaf_rn(p7, c2, p5), c2, p3), c2, p1) * c;
        else a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        if (use_fma) a = __fmaf_rn(-__fmaf_rn(__fmaf_rn(__fmaf_rn(p7, c2, p5), c2, p3), c2, p1), c, 90.f);
        else a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

__constant__ int c_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};

// One wavefront per keypoint: lane = (row group, 4-byte column group); the 31 x 32-byte window
// of the disc is read as 4 unaligned dwords per lane, integer moments reduced across the wave.
#define ORIENT_BX 16   // workgroups per (level, frame)
#define ORIENT_KPB 64  // keypoints per workgroup batch
__global__ __launch_bounds__(256) void k_orient(const Geom* __restrict__ g, FrameSrc src, const uint8_t* __restrict__ pyr,
                                                const int* __restrict__ cnt_sel2, const float* __restrict__ key,
                                                const uint32_t* __restrict__ payload, vo_keypoint* __restrict__ kps,
                                                uint32_t* __restrict__ kplvl, int* __restrict__ nkp, int* __restrict__ frame_status, int nframes,
                                                uint8_t* __restrict__ bneed) {
    const VBlock vb = xcd_block(ORIENT_BX, g->nlevels, nframes);
    if (!vb.valid) return;
    const int l = vb.y, f = vb.f;
    const LevelGeom& L = g->lv[l];
    int off = 0, total = 0;
    for (int l2 = 0; l2 < g->nlevels; l2++) {
        int c = cnt_sel2[f * VO_MAX_LEVELS + l2];
        if (l2 < l) off += c;
        total += c;
    }
    const int n = cnt_sel2[f * VO_MAX_LEVELS + l];
    if (l == 0 && vb.x == 0 && threadIdx.x == 0) {
        nkp[f] = min(total, g->kp_cap);
        frame_status[f] = total > g->kp_cap ? VO_ERR_CAPACITY : VO_OK;
    }
    int pitch;
    const uint8_t* img = level_ptr(g, src, pyr, f, l, pitch);
    const long cb = (long)f * g->cand_frame + L.cand_ofs;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rg = lane >> 3, u0 = -16 + 4 * (lane & 7);
    // A lane owns 4 adjacent columns u0..u0+3 of (up to) 4 rows v = -15 + rg + 8*it of the 31 x 31 window.
    // Per row: the disc limit |u| <= umax[|v|] becomes a byte mask, so the row's contribution is
    //   rs = b0+b1+b2+b3 (v_sad_u8),  sum u*b = u0*rs + (0*b0+1*b1+2*b2+3*b3) (v_dot4_u32_u8).
    int vrow[4];
    uint32_t bmask[4];
#pragma unroll
    for (int it = 0; it < 4; it++) {
        const int v = -15 + rg + 8 * it;
        const int um = v <= 15 ? c_umax[v < 0 ? -v : v] : -1;
        vrow[it] = min(v, 15);     // row 16 does not exist: its mask is empty, the load stays inside the window
        uint32_t m = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int u = u0 + j;
            if ((u < 0 ? -u : u) <= um) m |= 0xFFu << (8 * j);
        }
        bmask[it] = m;
    }
    __shared__ int s_m10[ORIENT_KPB], s_m01[ORIENT_KPB];
    // batches of 64 keypoints: each wavefront reduces the moments of 16 of them, then one lane per keypoint
    // does the arctangent and writes the KeyPoint record
    for (int i0 = vb.x * ORIENT_KPB; i0 < n; i0 += ORIENT_BX * ORIENT_KPB) {
        uint32_t my_p = 0;
        if (lane < ORIENT_KPB / 4 && i0 + wave + 4 * lane < n) my_p = payload[cb + i0 + wave + 4 * lane];
        for (int j = 0; j < ORIENT_KPB / 4; j++) {
            const int kk = wave + 4 * j;
            if (i0 + kk >= n) break;
            const uint32_t p = __shfl(my_p, j);
            const uint8_t* center = img + (long)UNPACK_Y(p) * pitch + UNPACK_X(p);
            int m10 = 0, m01 = 0;
#pragma unroll
            for (int it = 0; it < 4; it++) {
                uint32_t d;
                __builtin_memcpy(&d, center + (long)vrow[it] * pitch + u0, 4);
                d &= bmask[it];
                const int rs = (int)__builtin_amdgcn_sad_u8(d, 0u, 0u);
                m10 += u0 * rs + (int)__builtin_amdgcn_udot4(d, 0x03020100u, 0u, false);
                m01 += vrow[it] * rs;
            }
            m10 = wave_incl_scan(m10);
            m01 = wave_incl_scan(m01);
            if (lane == 63) { s_m10[kk] = m10; s_m01[kk] = m01; }
        }
        __syncthreads();
        const int i = i0 + threadIdx.x;
        if (threadIdx.x < ORIENT_KPB && i < n && off + i < g->kp_cap) {
            const uint32_t p = payload[cb + i];
            const int x0 = UNPACK_X(p), y0 = UNPACK_Y(p);
            vo_keypoint k;
            k.x = (float)x0 * L.scale;
            k.y = (float)y0 * L.scale;
            k.size = 31 * L.scale;
            k.angle = fast_atan2_deg((float)s_m01[threadIdx.x], (float)s_m10[threadIdx.x], g->fma_atan != 0);
            k.response = key[cb + i];
            k.octave = l;
            const long o = (long)f * g->kp_cap + off + i;
            kps[o] = k;
            kplvl[o] = PACK_XYS(x0, y0, l);
            if (bneed) {
                // the blur tiles (64 x 64) that this keypoint's descriptor patch reaches: rBRIEF samples within 19 px (RB_R).
                // Up to four byte stores of the same value; k_blur computes only marked tiles and clears the marks.
                uint8_t* nb = bneed + (long)f * g->nbtiles + L.btile_base;
                const int tx0 = max(x0 - 19, 0) / VO_BLUR_TW, tx1 = min(x0 + 19, L.w - 1) / VO_BLUR_TW;
                const int ty0 = max(y0 - 19, 0) / VO_BLUR_TH, ty1 = min(y0 + 19, L.h - 1) / VO_BLUR_TH;
                nb[ty0 * L.bntx + tx0] = 1; nb[ty0 * L.bntx + tx1] = 1; nb[ty1 * L.bntx + tx0] = 1; nb[ty1 * L.bntx + tx1] = 1;
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ 7x7 sigma=2 blur
__device__ __forceinline__ int reflect101c(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return min(max(i, 0), n - 1);
}

struct BlurK { float k[4]; };  // k[0] = centre weight, k[i] = weight at distance i
__device__ __forceinline__ float comp4(const float4& v, int j) { return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w; }
__device__ __forceinline__ int comp4(const int4& v, int j) { return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w; }

#define BL_SP 80                      // LDS source tile pitch (bytes): 72 used
#define BL_RPL 4                      // output rows per lane in the column pass (VO_BLUR_TH = 16 * BL_RPL)
#define BL_ROWS (VO_BLUR_TH + 6)
static_assert(VO_BLUR_TH == 16 * BL_RPL && VO_BLUR_TW == 64 && BL_ROWS % 14 == 0, "k_blur lane mapping");

// Tile = 64 x VO_BLUR_TH outputs.  Source window = (TH+6) rows x 72 bytes from the dword boundary
// x0-4.  Row pass: one lane makes 4 adjacent outputs from 3 LDS dwords (10 bytes -> float by
// v_cvt_f32_ubyteN) and stores a float4; column pass: one lane makes 4 adjacent outputs on each of
// BL_RPL consecutive rows from BL_RPL+6 float4 rows held in registers (every row-pass value is read
// from LDS 2.5 times instead of 7) and stores BL_RPL dwords.  The multiply/add order is
// sepFilter2D's.
// MODE (blur_mode) and FMA (vo_orb_params.fma_blur) are template parameters: as run-time flags they left a branch per output pixel.
template <int MODE, bool FMA>
__global__ __launch_bounds__(256) void k_blur(const Geom* __restrict__ g, FrameSrc src, const uint8_t* __restrict__ pyr,
                                              uint8_t* __restrict__ blur, const TileDesc* __restrict__ btiles, BlurK bk, int nframes,
                                              uint8_t* __restrict__ need) {
    __shared__ __attribute__((aligned(16))) uint8_t sp[BL_ROWS * BL_SP];
    __shared__ __attribute__((aligned(16))) float rowf[BL_ROWS * VO_BLUR_TW];
    const VBlock vb = xcd_block(g->nbtiles, 1, nframes);
    if (!vb.valid) return;
    const int f = vb.f, tid = threadIdx.x;
    // batch contexts: only the tiles that a descriptor patch reaches (k_orient marked them; the mark is cleared below, after the
    // barrier that follows every thread's read of it).  On real frames a third of the tiles holds no keypoint (sky, road).
    uint8_t* const mark = need ? need + (long)f * g->nbtiles + vb.x : nullptr;
    if (mark && *mark == 0) return;
    const TileDesc td = btiles[vb.x];
    const LevelGeom& L = g->lv[td.level];
    int pitch;
    const uint8_t* img = level_ptr(g, src, pyr, f, td.level, pitch);
    const int x0 = td.tx * VO_BLUR_TW, y0 = td.ty * VO_BLUR_TH;
    const int nrows = min(BL_ROWS, L.h - y0 + 6);      // source rows the tile's outputs (y < L.h) reach
    if (x0 - 4 >= 0 && x0 + 68 <= L.w && y0 - 3 >= 0 && y0 - 3 + BL_ROWS <= L.h) {
        // interior tile (workgroup-uniform): no reflection, one address and one stride per lane (252 lanes x 14 rows x 5)
        if (tid < 252) {
            const int r0 = tid / 18, k = tid - r0 * 18;
            const uint32_t po = (uint32_t)(y0 - 3 + r0) * (uint32_t)pitch + (uint32_t)(x0 - 4 + 4 * k);
            uint32_t* q = reinterpret_cast<uint32_t*>(&sp[r0 * BL_SP + 4 * k]);
            uint32_t v[BL_ROWS / 14];
#pragma unroll
            for (int j = 0; j < BL_ROWS / 14; j++) __builtin_memcpy(&v[j], img + (po + 14u * (uint32_t)j * (uint32_t)pitch), 4);
#pragma unroll
            for (int j = 0; j < BL_ROWS / 14; j++) q[14 * j * BL_SP / 4] = v[j];
        }
    } else if (tid < 252) {
        // border tile (54 % of the tiles of a KITTI-size pyramid): the same lane mapping -- a lane owns one dword column and
        // 5 rows, so the whole-dword test is per lane and only the rows are reflected; bytes are fetched one by one only
        // where the dword itself crosses the level's left or right edge
        const int r0 = tid / 18, k = tid - r0 * 18;
        const int x = x0 - 4 + 4 * k;
        const bool whole = x >= 0 && x + 3 < L.w;
#pragma unroll
        for (int j = 0; j < BL_ROWS / 14; j++) {
            const int r = r0 + 14 * j;
            if (r < nrows) {
                // 32-bit offsets from the level's (uniform) base
                const uint32_t ro = (uint32_t)reflect101c(y0 - 3 + r, L.h) * (uint32_t)pitch;
                uint32_t v;
                if (whole) {
                    __builtin_memcpy(&v, img + (ro + (uint32_t)x), 4);
                } else if (x < 0) {
                    // x = -4 (tile column 0): columns -4 .. -1 reflect to 4, 3, 2, 1 = the dword at column 1, bytes reversed
                    uint32_t d;
                    __builtin_memcpy(&d, img + (ro + 1u), 4);
                    v = __builtin_amdgcn_perm(0u, d, 0x00010203u);
                } else {
                    // at or beyond the right edge: with [a b c d] the row's last four pixels the reflected row continues
                    // d | c b a ?, and the outputs that are stored (x < L.w) reach no further than column L.w + 2 (= a)
                    uint32_t d;
                    __builtin_memcpy(&d, img + (ro + (uint32_t)(L.w - 4)), 4);
                    const unsigned long long e = (unsigned long long)d | ((unsigned long long)__builtin_amdgcn_perm(0u, d, 0x0C000102u) << 32);
                    const int sft = x - (L.w - 4);
                    v = sft < 8 ? (uint32_t)(e >> (8 * sft)) : 0u;
                }
                *reinterpret_cast<uint32_t*>(&sp[r * BL_SP + 4 * k]) = v;
            }
        }
    }
    __syncthreads();
    if (mark && tid == 0) *mark = 0;
    constexpr bool use_fma = FMA;
    constexpr int mode = MODE;
    // the weights live in VGPRs: a VOP2 multiply/fma with an SGPR operand issues at half rate on gfx950 (tools/ubench2.hip)
    float kw[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        kw[i] = bk.k[i];
        asm volatile("" : "+v"(kw[i]));
    }
    int* rowi = reinterpret_cast<int*>(rowf);
    for (int r = tid >> 4, c4 = (tid & 15) * 4; r < nrows; r += 16) {
        const uint32_t* q = reinterpret_cast<const uint32_t*>(&sp[r * BL_SP + c4]);
        const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
        // bytes c4+1 .. c4+10 of the window: b[j] = source pixel at output x0+c4+j-3
        uint32_t b[10];
        b[0] = (d0 >> 8) & 0xFF; b[1] = (d0 >> 16) & 0xFF; b[2] = d0 >> 24;
        b[3] = d1 & 0xFF; b[4] = (d1 >> 8) & 0xFF; b[5] = (d1 >> 16) & 0xFF; b[6] = d1 >> 24;
        b[7] = d2 & 0xFF; b[8] = (d2 >> 8) & 0xFF; b[9] = (d2 >> 16) & 0xFF;
        if (mode == 1) {
            int op[4];
#pragma unroll
            for (int j = 0; j < 4; j++)
                op[j] = 18 * (int)(b[j] + b[j + 6]) + 34 * (int)(b[j + 1] + b[j + 5]) + 49 * (int)(b[j + 2] + b[j + 4]) + 55 * (int)b[j + 3];
            *reinterpret_cast<int4*>(&rowi[r * VO_BLUR_TW + c4]) = make_int4(op[0], op[1], op[2], op[3]);
        } else {
            float fb[10];
#pragma unroll
            for (int j = 0; j < 10; j++) fb[j] = (float)b[j];
            float op[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                // RowFilter<uchar,float>: s = k[0]*p[-3]; s += k[i]*p[i-3] left to right
                float acc = kw[3] * fb[j];
                if (use_fma) {
                    acc = __fmaf_rn(kw[2], fb[j + 1], acc);
                    acc = __fmaf_rn(kw[1], fb[j + 2], acc);
                    acc = __fmaf_rn(kw[0], fb[j + 3], acc);
                    acc = __fmaf_rn(kw[1], fb[j + 4], acc);
                    acc = __fmaf_rn(kw[2], fb[j + 5], acc);
                    acc = __fmaf_rn(kw[3], fb[j + 6], acc);
                } else {
                    acc = acc + kw[2] * fb[j + 1];
                    acc = acc + kw[1] * fb[j + 2];
                    acc = acc + kw[0] * fb[j + 3];
                    acc = acc + kw[1] * fb[j + 4];
                    acc = acc + kw[2] * fb[j + 5];
                    acc = acc + kw[3] * fb[j + 6];
                }
                op[j] = acc;
            }
            *reinterpret_cast<float4*>(&rowf[r * VO_BLUR_TW + c4]) = make_float4(op[0], op[1], op[2], op[3]);
        }
    }
    __syncthreads();
    {
        const int c4 = (tid & 15) * 4, r0 = (tid >> 4) * BL_RPL;
        if (y0 + r0 >= L.h || x0 + c4 >= L.w) return;
        uint32_t packed[BL_RPL];
        if (mode == 1) {
            // rows at or beyond nrows were not written by the row pass: they only reach outputs below the level's last row,
            // which are not stored
            int4 q[BL_RPL + 6];
#pragma unroll
            for (int k = 0; k < BL_RPL + 6; k++) q[k] = *reinterpret_cast<const int4*>(&rowi[(r0 + k) * VO_BLUR_TW + c4]);
#pragma unroll
            for (int o = 0; o < BL_RPL; o++) {
                packed[o] = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int s = 18 * (comp4(q[o], j) + comp4(q[o + 6], j)) + 34 * (comp4(q[o + 1], j) + comp4(q[o + 5], j)) +
                                  49 * (comp4(q[o + 2], j) + comp4(q[o + 4], j)) + 55 * comp4(q[o + 3], j);
                    // s >= 0 here, so an unsigned shift + umin is the saturate_cast.  (Written this way on purpose:
                    // hipcc 7.2 folds `clamp(x >> 16, 0, 255)` pairs into v_ashr_pk_u8_i32 and then ORs byte 2
                    // over its upper half, which the instruction does not leave zero on gfx950 -- wrong pixels.)
                    packed[o] |= min((uint32_t)(s + 32768) >> 16, 255u) << (8 * j);
                }
            }
        } else {
            float4 q[BL_RPL + 6];
#pragma unroll
            for (int k = 0; k < BL_RPL + 6; k++) q[k] = *reinterpret_cast<const float4*>(&rowf[(r0 + k) * VO_BLUR_TW + c4]);
#pragma unroll
            for (int o = 0; o < BL_RPL; o++) {
                packed[o] = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    // SymmColumnFilter: s = k0*row[0] + delta; s += k[i]*(row[i] + row[-i])
                    // (delta = 0 is not added: the product of a positive weight and a row sum >= +0 is never -0, so `+ 0.f` is the identity)
                    float acc = kw[0] * comp4(q[o + 3], j);
                    const float t1 = comp4(q[o + 4], j) + comp4(q[o + 2], j);
                    const float t2 = comp4(q[o + 5], j) + comp4(q[o + 1], j);
                    const float t3 = comp4(q[o + 6], j) + comp4(q[o], j);
                    if (use_fma) {
                        acc = __fmaf_rn(kw[1], t1, acc);
                        acc = __fmaf_rn(kw[2], t2, acc);
                        acc = __fmaf_rn(kw[3], t3, acc);
                    } else {
                        acc = acc + kw[1] * t1;
                        acc = acc + kw[2] * t2;
                        acc = acc + kw[3] * t3;
                    }
                    // saturate_cast<uchar>(cvRound(acc)): 0 <= acc <= 255 * (sum of weights) < 255.5, so after the
                    // round-half-even the value is an integer in [0, 255] and the byte conversion is exact
                    packed[o] = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_rintf(acc), j, packed[o]);
                }
            }
        }
        uint8_t* dp = blur + (long)f * g->blur_frame + L.blur_ofs + (long)(y0 + r0) * L.bpitch + x0 + c4;
#pragma unroll
        for (int o = 0; o < BL_RPL; o++)
            if (y0 + r0 + o < L.h) *reinterpret_cast<uint32_t*>(dp + (long)o * L.bpitch) = packed[o];
    }
}

// ------------------------------------------------------------------ rBRIEF-256, one wavefront per keypoint
// A block takes 64 keypoints: first one lane per keypoint evaluates cos/sin (double precision,
// rounded to float, as cv2 does) so the trigonometry is not repeated by all 64 lanes of the
// describing wave; then each of the 4 waves describes 16 keypoints, 4 test pairs per lane.
// The rotated pattern reaches at most 19 px from the keypoint, so the wavefront first copies the
// 39-row patch to LDS with row-coalesced dword loads (44 B per row from a 4-byte boundary: ~40
// cache lines per keypoint instead of ~35 per gather instruction) and samples it there; the next
// keypoint's patch is loaded into registers while the current one is sampled.
#define RB_KPB 64
#define RB_R 19                       // sampling reach: |(x,y)| <= sqrt(13^2 + 13^2) = 18.4, rounded
#define RB_ROWS (2 * RB_R + 1)        // 39
#define RB_DW 11                      // dwords per patch row (39 B + up to 3 B of alignment slack)
#define RB_PITCH 48
#define RB_LOADS ((RB_ROWS * RB_DW + 63) / 64)   // 7 dwords per lane
static_assert(64 * (RB_LOADS - 1) <= RB_ROWS * RB_DW, "all but a lane's last dword lie inside the patch");
__global__ __launch_bounds__(256) void k_rbrief(const Geom* __restrict__ g, const uint8_t* __restrict__ blur, const signed char* __restrict__ pat_g,
                                                const int* __restrict__ nkp, const vo_keypoint* __restrict__ kps,
                                                const uint32_t* __restrict__ kplvl, uint8_t* __restrict__ desc, int nbx, int nframes) {
    __shared__ __attribute__((aligned(16))) signed char pat[1024];
    __shared__ float s_cos[RB_KPB], s_sin[RB_KPB];
    __shared__ int s_pitch[VO_MAX_LEVELS];
    __shared__ long s_ofs[VO_MAX_LEVELS];
    __shared__ __attribute__((aligned(16))) uint8_t patch[4][RB_ROWS * RB_PITCH];
    // XCD-aware order: blocks are dealt round-robin over the 8 XCDs, so block b and b+8 share an L2; give each
    // XCD a contiguous run of (frame, keypoint block) work so that a frame's blurred levels are fetched by one L2
    const int v = xcd_contiguous(blockIdx.x, gridDim.x);
    if (v >= nbx * nframes) return;
    const int f = v / nbx, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = nkp[f];
    const int k0 = (v - f * nbx) * RB_KPB;
    if (k0 >= n) return;
    reinterpret_cast<int*>(pat)[tid] = reinterpret_cast<const int*>(pat_g)[tid];
    if (tid < RB_KPB && k0 + tid < n) {
        float angle = kps[(long)f * g->kp_cap + k0 + tid].angle;
        angle *= (float)(M_PI / 180.f);
        if (g->trig_mode == 1) {   // the float overloads: glibc's cosf / sinf, restated (include/vo_sincosf.h)
            s_cos[tid] = vo_glibc_cosf(angle);
            s_sin[tid] = vo_glibc_sinf(angle);
        } else {
            s_cos[tid] = (float)cos((double)angle);
            s_sin[tid] = (float)sin((double)angle);
        }
    }
    if (tid >= 64 && tid < 64 + VO_MAX_LEVELS) {
        s_pitch[tid - 64] = g->lv[tid - 64].bpitch;
        s_ofs[tid - 64] = g->lv[tid - 64].blur_ofs;
    }
    // the wavefront's 16 keypoints are k0 + wave + 4i: lane i fetches the packed (x, y, level) of the i-th
    const long fo = (long)f * g->kp_cap;
    uint32_t my_kl = 0;
    if (lane < RB_KPB / 4 && k0 + wave + 4 * lane < n) my_kl = kplvl[fo + k0 + wave + 4 * lane];
    __syncthreads();
    // this lane's 4 test pairs (8 pattern points)
    float px0[4], py0[4], px1[4], py1[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const signed char* pp = &pat[(lane * 4 + j) * 4];
        px0[j] = (float)pp[0]; py0[j] = (float)pp[1]; px1[j] = (float)pp[2]; py1[j] = (float)pp[3];
        // keep the 16 floats in registers: without the pin the compiler re-converts them from the packed bytes for every keypoint
        asm volatile("" : "+v"(px0[j]), "+v"(py0[j]), "+v"(px1[j]), "+v"(py1[j]));
    }
    const uint8_t* fblur = blur + (long)f * g->blur_frame;
    uint8_t* mypatch = patch[wave];
    // this lane's share of a patch: dwords i = lane + 64t -> (row i / 11, dword i % 11)
    int l_src[RB_LOADS], l_dst[RB_LOADS];   // row * 1 (scaled by pitch at use) and LDS byte offset
#pragma unroll
    for (int t = 0; t < RB_LOADS; t++) {
        const int i = lane + 64 * t, r = i / RB_DW, d = i - r * RB_DW;
        l_src[t] = i < RB_ROWS * RB_DW ? (r << 8 | d) : -1;
        l_dst[t] = r * RB_PITCH + 4 * d;
    }
    uint32_t regs[RB_LOADS];
    int cur_ax = 0;
    auto fetch = [&](int i, int& ax) {   // global -> registers for the wavefront's i-th keypoint
        const uint32_t kl = __shfl(my_kl, i);
        const int lvl = UNPACK_S(kl) & (VO_MAX_LEVELS - 1);
        const int pitch = s_pitch[lvl];
        const int x0 = UNPACK_X(kl) - RB_R;
        ax = x0 & 3;                                       // patch column 0 sits at byte ax of the LDS rows
        const uint8_t* base = fblur + s_ofs[lvl] + (long)(UNPACK_Y(kl) - RB_R) * pitch + (x0 - ax);
#pragma unroll
        for (int t = 0; t < RB_LOADS; t++)
            if (t < RB_LOADS - 1 || l_src[t] >= 0)   // only the last of a lane's dwords can lie beyond the patch (64 t + lane < 429 for t < 6)
                regs[t] = *reinterpret_cast<const uint32_t*>(base + (__umul24((uint32_t)(l_src[t] >> 8), (uint32_t)pitch) + 4u * (uint32_t)(l_src[t] & 0xFF)));
    };
    const int nmine = (min(n - k0, RB_KPB) - wave + 3) >> 2;   // keypoints of this wavefront
    if (nmine > 0) fetch(0, cur_ax);
    for (int i = 0; i < nmine; i++) {
        const int kk = wave + 4 * i;
        const int ax = cur_ax;
#pragma unroll
        for (int t = 0; t < RB_LOADS; t++)
            if (t < RB_LOADS - 1 || l_src[t] >= 0) *reinterpret_cast<uint32_t*>(mypatch + l_dst[t]) = regs[t];
        if (i + 1 < nmine) fetch(i + 1, cur_ax);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint8_t* center = mypatch + RB_R * RB_PITCH + RB_R + ax;
        const float a = s_cos[kk], b = s_sin[kk];
        uint32_t nib = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float xa0 = px0[j] * a, yb0 = py0[j] * b, xb0 = px0[j] * b, ya0 = py0[j] * a;
            const float xa1 = px1[j] * a, yb1 = py1[j] * b, xb1 = px1[j] * b, ya1 = py1[j] * a;
            const int ix0 = __float2int_rn(xa0 - yb0), iy0 = __float2int_rn(xb0 + ya0);
            const int ix1 = __float2int_rn(xa1 - yb1), iy1 = __float2int_rn(xb1 + ya1);
            // |iy| <= 19: a 24-bit multiply (v_mul_i32_i24) instead of the quarter-rate 32-bit one
            const int t0 = center[__mul24(iy0, RB_PITCH) + ix0], t1 = center[__mul24(iy1, RB_PITCH) + ix1];
            nib |= (uint32_t)(t0 < t1) << j;
        }
        const uint32_t hi = __shfl_down(nib, 1);
        if ((lane & 1) == 0) desc[(fo + k0 + kk) * 32 + (lane >> 1)] = (uint8_t)(nib | (hi << 4));
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();   // all lanes have sampled before the patch is overwritten
    }
}

// ------------------------------------------------------------------ host: launch the chain
static void gaussian_kernel7_sigma2(BlurK& bk) {
    // getGaussianKernel(7, 2, CV_32F) (OpenCV computes it in softdouble)
    double t[3], sum = 0;
    const double scale2X = -0.125 / (2.0 * 2.0);
    for (int i = 0, x = 1 - 7; i < 3; i++, x += 2) {
        t[i] = std::exp((double)(x * x) * scale2X);
        sum += t[i];
    }
    sum *= 2;
    sum += 1.0;
    const double mul1 = 1.0 / sum;
    bk.k[0] = (float)(1.0 * mul1);
    bk.k[3] = (float)(t[0] * mul1);
    bk.k[2] = (float)(t[1] * mul1);
    bk.k[1] = (float)(t[2] * mul1);
}

int vo_launch_detect(vo_ctx* c, const FrameSrc& src, int F, hipStream_t st) {
    const Geom& g = c->geom;
    const Geom* dg = (const Geom*)c->d_geom.p;
    uint8_t* pyr = (uint8_t*)c->d_pyr.p;
    uint8_t* blur = (uint8_t*)c->d_blur.p;
    const TileDesc* tiles = (const TileDesc*)c->d_tiles.p;
    float* key = (float*)c->d_key.p;
    uint32_t* payload = (uint32_t*)c->d_payload.p;
    int* cnt_fast = (int*)c->d_cnt_fast.p;
    int* cnt_sel1 = (int*)c->d_cnt_sel1.p;
    int* cnt_sel2 = (int*)c->d_cnt_sel2.p;
    if (F < 1 || F > c->max_frames) return vo_fail(c, VO_ERR_INVALID, "frame count exceeds the context's chunk size");

    for (int l = 1; l < g.nlevels; l++) {
        const LevelGeom& L = g.lv[l];
        dim3 grid((L.w + 255) / 256, (L.h + 4 * PYR_ROWS - 1) / (4 * PYR_ROWS), F);   // plain order: the XCD-contiguous order measured 7 % slower here (no inter-block reuse)
        KT(c, st, VO_K_PYRAMID, hipLaunchKernelGGL(k_pyramid, grid, dim3(256), 0, st, dg, src, pyr, (const ResizeCoef*)c->d_rtab.p, l));
    }
    // The blur only needs the pyramid: outside the timing mode, and while at most two contexts share the device, it runs on the context's second stream, next to the
    // FAST -> retainBest -> Harris -> retainBest -> orientation chain (whose retainBest replays are latency bound and
    // leave most of the chip idle), and joins before the descriptors.
    BlurK bk;
    gaussian_kernel7_sigma2(bk);
    auto blur_fn = g.blur_mode == 1 ? k_blur<1, false> : (g.fma_blur ? k_blur<0, true> : k_blur<0, false>);
    static const bool serial = getenv("VO_SERIAL") != nullptr;   // profiling aid: one kernel at a time, so trace durations are per kernel
    const bool fork_blur = !serial && !c->timing && c->debug_stop == 0 && c->aux_stream != nullptr && vo_live_contexts(c->device) <= 2;
    // Contexts that hold more than one pair blur only the tiles a descriptor patch reaches (the blur then has to come after the
    // orientation kernel, which marks them: not in the forked arrangement).  A single-pair context blurs whole levels, so that
    // vo_debug_read(VO_DBG_BLUR_LEVEL) returns complete images there.
    uint8_t* const bneed = (!fork_blur && c->debug_stop == 0 && c->max_pairs > 1) ? (uint8_t*)c->d_bneed.p : nullptr;
    if (fork_blur) {
        HIP_TRY(c, hipEventRecord(c->ev_fork, st));
        HIP_TRY(c, hipStreamWaitEvent(c->aux_stream, c->ev_fork, 0));
        hipLaunchKernelGGL(blur_fn, dim3(pad8(g.nbtiles * F)), dim3(256), 0, c->aux_stream, dg, src, pyr, blur, (const TileDesc*)c->d_btiles.p, bk, F, (uint8_t*)nullptr);
        HIP_TRY(c, hipEventRecord(c->ev_join, c->aux_stream));
    }
    if (g.ntiles > 0) {
        KT(c, st, VO_K_FAST_TILES, hipLaunchKernelGGL(k_fast_tiles, dim3(pad8(g.ntiles * F)), dim3(256), 0, st, dg, src, pyr, tiles, (uint32_t*)c->d_tile_list.p, (uint8_t*)c->d_seg_cnt.p, (uint32_t*)c->d_seg_ofs.p, F));
        // d_seg_ofs holds the per-tile totals (ntiles <= nsegs per frame)
        int ntrows = 0;
        for (int l = 0; l < g.nlevels; l++) ntrows += g.lv[l].nty;
        KT(c, st, VO_K_FAST_EMIT, hipLaunchKernelGGL(k_fast_emit, dim3(pad8(ntrows * F)), dim3(256), 0, st, dg, (const uint32_t*)c->d_tile_list.p, (const uint8_t*)c->d_seg_cnt.p,
                           (const uint32_t*)c->d_seg_ofs.p, cnt_fast, key, payload, ntrows, F));
    }
    if (c->debug_stop == 1) { HIP_TRY(c, hipGetLastError()); return VO_OK; }
    // long candidate lists (4K frames: ~10^5 per level) are pass-bound, short ones barrier-bound
    if ((long)g.w * g.h > 2000000L)
        KT(c, st, VO_K_SELECT1, hipLaunchKernelGGL((k_select<1024, 64>), dim3(g.nlevels, F), dim3(1024), 0, st, dg, key, payload, (uint32_t*)c->d_lpos.p, (uint32_t*)c->d_rpos.p, cnt_fast, cnt_sel1, 2, 0));
    else {
        // The lists of the small levels are short (level l holds about 1.44^-l of level 0's candidates): they get a variant with
        // 30 KB instead of 80 KB of LDS per workgroup, so that the LDS-tiled kernels of the other lanes (FAST, blur, rBRIEF,
        // matcher) can share a CU with the replay instead of waiting for it.  A list longer than CAP runs from global memory.
        // A context without extra lanes has nothing to share the CU with: all levels in one launch (0.054 against 0.083 ms).
        const int split = c->single_lane ? g.nlevels : std::min(VO_SELECT_BIG_LEVELS, g.nlevels);
        KT(c, st, VO_K_SELECT1, hipLaunchKernelGGL((k_select<256, 8000>), dim3(split, F), dim3(256), 0, st, dg, key, payload, (uint32_t*)c->d_lpos.p, (uint32_t*)c->d_rpos.p, cnt_fast, cnt_sel1, 2, 0));
        if (g.nlevels > split)
            KT(c, st, VO_K_SELECT1, hipLaunchKernelGGL((k_select<256, VO_SELECT_SMALL_CAP>), dim3(g.nlevels - split, F), dim3(256), 0, st, dg, key, payload, (uint32_t*)c->d_lpos.p, (uint32_t*)c->d_rpos.p, cnt_fast, cnt_sel1, 2, split));
    }
    KT(c, st, VO_K_HARRIS, hipLaunchKernelGGL(k_harris, dim3(pad8(HARRIS_BX * g.nlevels * F)), dim3(256), 0, st, dg, src, pyr, cnt_sel1, key, payload, F));
    if (c->debug_stop == 2) { HIP_TRY(c, hipGetLastError()); return VO_OK; }
    KT(c, st, VO_K_SELECT2, hipLaunchKernelGGL((k_select<256, 2048>), dim3(g.nlevels, F), dim3(256), 0, st, dg, key, payload, (uint32_t*)c->d_lpos.p, (uint32_t*)c->d_rpos.p, cnt_sel1, cnt_sel2, 1, 0));
    KT(c, st, VO_K_ORIENT, hipLaunchKernelGGL(k_orient, dim3(pad8(ORIENT_BX * g.nlevels * F)), dim3(256), 0, st, dg, src, pyr, cnt_sel2, key, payload, (vo_keypoint*)c->d_kp.p,
                       (uint32_t*)c->d_kplvl.p, (int*)c->d_nkp.p, (int*)c->d_frame_status.p, F, bneed));
    if (fork_blur) HIP_TRY(c, hipStreamWaitEvent(st, c->ev_join, 0));
    else KT(c, st, VO_K_BLUR, hipLaunchKernelGGL(blur_fn, dim3(pad8(g.nbtiles * F)), dim3(256), 0, st, dg, src, pyr, blur, (const TileDesc*)c->d_btiles.p, bk, F, bneed));
    const int rb_nbx = (g.kp_cap + RB_KPB - 1) / RB_KPB;
    KT(c, st, VO_K_RBRIEF, hipLaunchKernelGGL(k_rbrief, dim3(pad8(rb_nbx * F)), dim3(256), 0, st, dg, blur, (const signed char*)c->d_pat.p, (const int*)c->d_nkp.p,
                       (const vo_keypoint*)c->d_kp.p, (const uint32_t*)c->d_kplvl.p, (uint8_t*)c->d_desc.p, rb_nbx, F));
    HIP_TRY(c, hipGetLastError());
    return VO_OK;
}
