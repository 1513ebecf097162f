// vo_internal.h -- shared definitions of the MI355X visual-odometry library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/vo_mi355.h"

#define VO_MAX_LEVELS 8
#define VO_EDGE 31          // runByImageBorder(edgeThreshold)
#define VO_TILE_W 60        // FAST tile (output pixels): 60 x 30, i.e. a 64 x 32 score region = 2 runs of 4 per lane
#define VO_TILE_H 30
#define VO_TILE_CAP 512     // a 60x30 tile holds at most 30*15 = 450 strict 3x3 maxima
#define VO_BLUR_TW 64
#define VO_BLUR_TH 64
#define VO_MAX_LANES 6      // chunk workspaces (+ streams) per context, used round-robin (measured: 4 lanes 50.0 k, 6: 53.0 k, 8: 51.7 k pairs/s)

struct LevelGeom {
    int w, h, pitch;        // level size; row pitch of the level inside the pyramid slab
    int quota;              // n_l
    float scale;            // layerScale[l]
    long pyr_ofs;           // byte offset of the level in a frame's pyramid slab (level 0 unused)
    long blur_ofs;          // byte offset in a frame's blurred slab; pitch = bpitch
    int bpitch;
    int iw, ih;             // interior (border filtered) region [31, w-31) x [31, h-31); 0 if empty
    int ntx, nty;           // FAST tiles over the interior
    int tile_base;          // first tile index of the level inside a frame
    int seg_base;           // first (row, tile column) segment of the level inside a frame
    int cand_cap;           // capacity of the level's candidate list
    long cand_ofs;          // element offset of the level's list inside a frame's candidate arrays
    int xtab, ytab;         // offsets into the resize coefficient table (level >= 1)
    int btile_base, bntx, bnty;  // blur tiles
};

struct Geom {
    int nlevels, w, h;
    int ntiles;             // FAST tiles per frame
    int nsegs;              // segments per frame (padded)
    int nbtiles;            // blur tiles per frame
    long pyr_frame;         // bytes per frame in the pyramid slab
    long blur_frame;        // bytes per frame in the blurred slab
    long cand_frame;        // candidate elements per frame
    int kp_cap;             // keypoints per frame (capacity)
    int fast_threshold, blur_mode, fma_atan, fma_blur, trig_mode;
    LevelGeom lv[VO_MAX_LEVELS];
};

struct TileDesc { short level, tx, ty, pad; };

// resize coefficient: source offset and the 8-bit fraction c1 (c0 = 256 - c1)
typedef uint32_t ResizeCoef;   // source index | 8-bit fraction c1 << 16

// where a chunk's frames come from: frames [0,nA) are a + i*frame_stride, [nA, ...) are b + ...
struct FrameSrc {
    const uint8_t* a;
    const uint8_t* b;
    int nA;
    int pitch;
    long frame_stride;
};

__device__ __forceinline__ const uint8_t* frame_ptr(const FrameSrc& s, int f) {
    return f < s.nA ? s.a + (long)f * s.frame_stride : s.b + (long)(f - s.nA) * s.frame_stride;
}

// pointer + pitch of unblurred level l of frame f
__device__ __forceinline__ const uint8_t* level_ptr(const Geom* g, const FrameSrc& s, const uint8_t* pyr, int f, int l, int& pitch) {
    if (l == 0) {
        pitch = s.pitch;
        return frame_ptr(s, f);
    }
    pitch = g->lv[l].pitch;
    return pyr + (long)f * g->pyr_frame + g->lv[l].pyr_ofs;
}

#define PACK_XYS(x, y, s) ((uint32_t)(x) | ((uint32_t)(y) << 12) | ((uint32_t)(s) << 24))
#define UNPACK_X(p) ((int)((p) & 0xFFFu))
#define UNPACK_Y(p) ((int)(((p) >> 12) & 0xFFFu))
#define UNPACK_S(p) ((int)((p) >> 24))

// ---------------------------------------------------------------- context
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

struct vo_ctx {
    int device = 0;
    bool counted = false;   // included in the per-device live-context count
    int max_w = 0, max_h = 0, max_frames = 0, chunk_pairs = 0, max_pairs = 0;
    vo_orb_params params;
    std::string err;
    hipStream_t stream = nullptr;  // own stream for the host-pointer entry points
    hipStream_t aux_stream = nullptr;            // second stream: the blur runs beside the keypoint chain
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_in = nullptr;                  // orders the private stream after the default stream (stream == NULL calls)

    // geometry currently built (w,h) and its device copies
    Geom geom;
    bool geom_valid = false;
    std::vector<TileDesc> h_tiles, h_btiles;
    std::vector<ResizeCoef> h_rtab;
    DevBuf d_geom, d_tiles, d_btiles, d_rtab;

    // per-chunk workspace, sized for max_frames frames of max_w x max_h
    DevBuf d_img;          // staging for host images
    DevBuf d_pyr, d_blur;
    DevBuf d_tile_list, d_seg_cnt, d_seg_ofs;
    DevBuf d_key, d_payload, d_lpos, d_rpos;
    DevBuf d_cnt_fast, d_cnt_sel1, d_cnt_sel2;   // int [F][L]
    DevBuf d_nkp, d_kp, d_kplvl, d_desc, d_frame_status;
    DevBuf d_bneed;        // uint8 [F][nbtiles]: blur tiles that a descriptor patch reaches (set by k_orient, consumed by k_blur)
    // matching / pose workspace (per pair)
    DevBuf d_match;        // int4 [P][kp_cap] best/second per query
    DevBuf d_nmatch, d_mq, d_q1, d_q2;
    DevBuf d_pose;         // PoseWork [P]
    DevBuf d_hyp;          // hypothesis models
    DevBuf d_mask;
    DevBuf d_result;       // vo_pair_result [P]
    DevBuf d_misc;         // small scratch for the single-call entry points
    DevBuf d_pat;          // rBRIEF pattern (1024 x int8)
    DevBuf d_pair_status;  // int [max_pairs]
    DevBuf d_samples, d_nmodels, d_counts, d_vote, d_pair_nkp, d_raw, d_tri;
    int kp_cap = 0;
    // The arrays the match stage hands to the pose stage exist twice ("slots"): while the pose stage of one batch of pairs
    // runs on pose_stream, detect + match of the next batch fill the other slot on the main stream.
    int slot = 0;
    int* nmatch() const { return (int*)d_nmatch.p + (size_t)slot * max_pairs; }
    int* pair_status() const { return (int*)d_pair_status.p + (size_t)slot * max_pairs; }
    int2* pair_nkp() const { return (int2*)d_pair_nkp.p + (size_t)slot * max_pairs; }
    int4* mq() const { return (int4*)d_mq.p + (size_t)slot * max_pairs * kp_cap; }
    float2* q1() const { return (float2*)d_q1.p + (size_t)slot * max_pairs * kp_cap; }
    float2* q2() const { return (float2*)d_q2.p + (size_t)slot * max_pairs * kp_cap; }
    hipStream_t pose_stream = nullptr;
    // Lanes: the context is lane 0; `lanes` are shallow clones owning a chunk workspace + streams of their own and
    // aliasing the geometry tables and pair arrays of `owner` (vo_api.hip: lane_init / lane_alias_shared).
    vo_ctx* owner = nullptr;
    bool single_lane = false;   // the context has no extra lanes (a single-pair context): nothing runs beside its kernels
    std::vector<vo_ctx*> lanes;
    int last_lane = 0;                              // lane of the last chunk (vo_debug_read)
    // asynchronous submissions (vo_pairs_submit_device / vo_pairs_wait): at most two in flight, one per slot
    struct Submission { bool live = false; int B = 0; vo_pair_result* out = nullptr; };
    Submission sub[2];
    unsigned long n_submitted = 0, n_waited = 0;
    hipEvent_t ev_done[2] = {nullptr, nullptr};
    vo_pair_result* h_result = nullptr;             // pinned, [2][max_pairs]: a D2H copy into pageable memory would block the submit
    hipEvent_t ev_lane[2] = {nullptr, nullptr};     // this lane's chunks of the batch in slot s have been enqueued
    int debug_stop = 0;    // 0: run everything; 1: stop after FAST emit; 2: stop after retainBest #1 + Harris
    void* h_pinned = nullptr;
    size_t h_pinned_bytes = 0;

    // last chunk bookkeeping for vo_debug_read
    FrameSrc last_src{};
    int last_frames = 0, last_pairs = 0;
    bool results_valid = false;   // d_result (per-pair status) describes the pair arrays of the current slot: set by the pose waves, cleared by the single-pair entry points

    // per-kernel timing with HIP events on the launch stream (diagnostic mode)
    bool timing = false;
    float kernel_ms[VO_N_KERNELS] = {0};
    int kernel_launches[VO_N_KERNELS] = {0};
    std::vector<hipEvent_t> ev_pool;
    std::vector<int> ev_ids;      // kernel id of each bracketed launch; events 2*i, 2*i+1
};

int vo_live_contexts(int device);   // contexts alive on the device (all threads of the process)
void kt_begin(vo_ctx* c, hipStream_t st, int id);
void kt_end(vo_ctx* c, hipStream_t st);
int kt_collect(vo_ctx* c);   // after the stream is synchronised
#ifdef VO_EXPERIMENT_SKIP
// experiment builds only (tools/skip_sensitivity.sh): leave out the launches of the kernels named by the bit mask in the
// environment, to measure what a kernel costs END TO END when everything overlaps (results are then meaningless)
#include <cstdlib>
static inline bool vo_skip(int id) { const char* e = getenv("VO_SKIP_MASK"); return e && ((atoi(e) >> id) & 1); }
#define KT(c, st, id, launch) \
    do { if (!vo_skip(id)) { kt_begin((c), (st), (id)); launch; kt_end((c), (st)); } } while (0)
#else
#define KT(c, st, id, launch) \
    do { kt_begin((c), (st), (id)); launch; kt_end((c), (st)); } while (0)
#endif

int vo_fail(vo_ctx* c, int code, const std::string& msg);
#define HIP_TRY(c, expr)                                                                         \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) return vo_fail((c), VO_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

// detect.hip
int vo_build_geometry(vo_ctx* c, int w, int h);
int vo_launch_detect(vo_ctx* c, const FrameSrc& src, int F, hipStream_t st);
// match.hip
int vo_match_segments(int kp_cap);   // train-set segments per pair (rows of d_match)
int vo_launch_match_impl(vo_ctx* c, int P, int nA, double ratio, int pair0, int4* raw_out, hipStream_t st);
int vo_noalloc_init(vo_ctx* c);      // resolves this device's copy of the no-allocation scratch array (once per device)
// pose.hip
struct PoseK { double fx, fy, cx, cy; };
// RANSAC (findEssentialMat) for pairs [0,P) of the pair arrays; E/mask/n_inlier left in d_pose / d_mask
int vo_launch_ransac(vo_ctx* c, int P, const double K[9], double prob, double thr, int max_iters, hipStream_t st, bool host_free = false);
// recoverPose for pairs [0,P) using the E in d_pose (or E given per pair by the caller in d_pose)
int vo_launch_recover(vo_ctx* c, int P, const double K[9], double dist_thresh, hipStream_t st);
int vo_launch_triangulate_pairs(vo_ctx* c, int P, const double* dPm, const long* doffsets, float* dX3, float2* dq1, float2* dq2, hipStream_t st);
int vo_launch_triangulate(vo_ctx* c, const double* dP, const float* dx1, const float* dx2, int n, float* dX4, hipStream_t st);
int vo_launch_debug_minimal(vo_ctx* c, int mode, const double* d_in, int n, double* d_out, hipStream_t st);

#define VO_HYP_ROUND 64   // RANSAC hypotheses solved per round
struct PoseWork {
    double E[9];          // best model so far / final
    double R[9], t[3];    // recoverPose result
    double R1[9], R2[9], tv[3];
    unsigned long long rng;
    int niters, iter, max_good, done;
    int status, n_good, pad0, pad1;
};
