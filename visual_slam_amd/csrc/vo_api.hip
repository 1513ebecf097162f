// vo_api.hip -- context, workspace and the extern "C" entry points of libvo_mi355.so
// (declared in include/vo_mi355.h; each stands behind one cv2 call of
// /root/reference/libs/vo_utils.py).  No CPU fallback: every entry point runs HIP kernels.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstddef>
#include <cstring>
#include <functional>

#include "../../include/vo_orb_pattern_31.h"
#include "vo_internal.h"

// contexts alive per device: vo_launch_detect forks the blur onto the context's second stream only while the process
// drives at most two contexts on the device (HIP multiplexes streams onto 4 hardware queues; with more contexts the
// forked blurs compete with other contexts' main streams for a queue: measured 49.8 k vs 52.7 k pairs/s at 4 contexts)
#include <atomic>
static std::atomic<int> g_live[64];
int vo_live_contexts(int device) { return device >= 0 && device < 64 ? g_live[device].load() : 0; }

int vo_fail(vo_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}

// ---- per-kernel timing: events recorded around a launch on the launch stream
void kt_begin(vo_ctx* c, hipStream_t st, int id) {
    if (!c->timing) return;
    const size_t i = c->ev_ids.size();
    while (c->ev_pool.size() < 2 * (i + 1)) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) { c->timing = false; return; }
        c->ev_pool.push_back(e);
    }
    c->ev_ids.push_back(id);
    (void)hipEventRecord(c->ev_pool[2 * i], st);
}
void kt_end(vo_ctx* c, hipStream_t st) {
    if (!c->timing || c->ev_ids.empty()) return;
    (void)hipEventRecord(c->ev_pool[2 * (c->ev_ids.size() - 1) + 1], st);
}
int kt_collect(vo_ctx* c) {
    for (size_t i = 0; i < c->ev_ids.size(); i++) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, c->ev_pool[2 * i], c->ev_pool[2 * i + 1]) == hipSuccess) {
            c->kernel_ms[c->ev_ids[i]] += ms;
            c->kernel_launches[c->ev_ids[i]] += 1;
        }
    }
    c->ev_ids.clear();
    return VO_OK;
}

static int dev_alloc(vo_ctx* c, DevBuf& b, size_t bytes) {
    bytes = std::max<size_t>(bytes, 256);
    HIP_TRY(c, hipMalloc(&b.p, bytes));
    b.bytes = bytes;
    return VO_OK;
}
static void dev_free(DevBuf& b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
}
#define ALLOC(c, buf, bytes)                         \
    do {                                             \
        int _rc = dev_alloc((c), (buf), (bytes));    \
        if (_rc != VO_OK) return _rc;                \
    } while (0)

__global__ void k_results(const PoseWork* __restrict__ pw, const int* __restrict__ nmatch, const int2* __restrict__ pair_nkp, int P,
                          vo_pair_result* __restrict__ out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const PoseWork& w = pw[p];
    vo_pair_result r;
    for (int i = 0; i < 16; i++) r.T[i] = 0;
    if (w.status == VO_OK) {
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) r.T[i * 4 + j] = w.R[i * 3 + j];
            r.T[i * 4 + 3] = w.t[i];
        }
        r.T[15] = 1;
    }
    r.n_kp1 = pair_nkp[p].x;
    r.n_kp2 = pair_nkp[p].y;
    r.n_match = nmatch[p];
    r.n_inlier = w.max_good;
    r.n_good = w.n_good;
    r.status = w.status;
    out[p] = r;
}

extern "C" {

void vo_default_orb_params(vo_orb_params* p) {
    p->nfeatures = 3000;  // libs/vo_utils.py:23
    p->scale_factor = 1.2f;
    p->nlevels = 8;
    p->edge_threshold = 31;
    p->patch_size = 31;
    p->fast_threshold = 20;
    p->blur_mode = 0;
    p->fma_atan = 1;
    p->fma_blur = 1;
    p->trig_mode = 0;
    p->keypoint_capacity = 0;
}

// Streams, events and the per-chunk workspace of ONE lane (the context itself is lane 0; extra lanes are shallow clones
// that own only these and alias everything else).  A chunk of 2*chunk_pairs frames goes through detect + match on one
// lane; consecutive chunks go to different lanes, so that the latency-bound kernels of one chunk (retainBest replay,
// small pyramid levels, scans) overlap the throughput-bound kernels of the others without any help from the caller.
static int lane_init(vo_ctx* c, bool with_aux) {
    const vo_orb_params& p = c->params;
    HIP_TRY(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    // the blur fork (vo_launch_detect) is only used by contexts of at most two lanes: HIP multiplexes streams onto a few
    // hardware queues, and streams that are never used would still take their turn in that assignment
    if (with_aux) HIP_TRY(c, hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking));
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    for (int i = 0; i < 2; i++) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_lane[i], hipEventDisableTiming));
    c->kp_cap = p.keypoint_capacity > 0 ? std::max(p.keypoint_capacity, p.nfeatures) : p.nfeatures + 1024;
    const long F = c->max_frames, kc = c->kp_cap, CP = c->chunk_pairs;
    // upper bounds from the max image size (all quantities are monotone in w, h)
    const long W = c->max_w, H = c->max_h;
    long pyr = 0, blur = 0, cand = 0, tiles = 0, segs = 0, btiles = 0;
    {
        const double sf = (double)p.scale_factor;
        for (int l = 0; l < p.nlevels; l++) {
            const float sc = (float)std::pow(sf, (double)l);
            const long w = (long)nearbyintf(W * (1.f / sc)) + 1, h = (long)nearbyintf(H * (1.f / sc)) + 1;
            const long pitch = (w + 63) / 64 * 64;
            if (l > 0) pyr += (pitch * h + 255) / 256 * 256;
            blur += (pitch * h + 255) / 256 * 256;
            const long iw = std::max(w - 62, 0L), ih = std::max(h - 62, 0L);
            const long ntx = (iw + VO_TILE_W - 1) / VO_TILE_W, nty = (ih + VO_TILE_H - 1) / VO_TILE_H;
            tiles += ntx * nty;
            btiles += ((w + VO_BLUR_TW - 1) / VO_BLUR_TW) * ((h + VO_BLUR_TH - 1) / VO_BLUR_TH);
            segs += (ih * ntx + 15) / 16 * 16;
            cand += (((iw + 1) / 2) * ((ih + 1) / 2) + 63) / 64 * 64;
        }
    }
    ALLOC(c, c->d_pyr, (size_t)F * (pyr + 256));
    ALLOC(c, c->d_blur, (size_t)F * (blur + 256));
    ALLOC(c, c->d_tile_list, (size_t)F * tiles * VO_TILE_CAP * 4);
    ALLOC(c, c->d_seg_cnt, (size_t)F * segs);
    ALLOC(c, c->d_seg_ofs, (size_t)F * segs * 4);
    ALLOC(c, c->d_key, (size_t)F * cand * 4);
    ALLOC(c, c->d_payload, (size_t)F * cand * 4);
    ALLOC(c, c->d_lpos, (size_t)F * cand * 4);
    ALLOC(c, c->d_rpos, (size_t)F * cand * 4);
    ALLOC(c, c->d_cnt_fast, (size_t)F * VO_MAX_LEVELS * 4);
    ALLOC(c, c->d_cnt_sel1, (size_t)F * VO_MAX_LEVELS * 4);
    ALLOC(c, c->d_cnt_sel2, (size_t)F * VO_MAX_LEVELS * 4);
    ALLOC(c, c->d_nkp, (size_t)F * 4);
    ALLOC(c, c->d_frame_status, (size_t)F * 4);
    ALLOC(c, c->d_kp, (size_t)F * kc * sizeof(vo_keypoint));
    ALLOC(c, c->d_kplvl, (size_t)F * kc * 4);
    ALLOC(c, c->d_desc, (size_t)F * kc * 32);
    ALLOC(c, c->d_bneed, (size_t)F * (btiles + 8));
    ALLOC(c, c->d_match, (size_t)CP * vo_match_segments(kc) * kc * sizeof(uint2));
    ALLOC(c, c->d_raw, (size_t)kc * sizeof(int4));
    HIP_TRY(c, hipMemset(c->d_frame_status.p, 0, c->d_frame_status.bytes));
    HIP_TRY(c, hipMemset(c->d_kp.p, 0, c->d_kp.bytes));
    HIP_TRY(c, hipMemset(c->d_bneed.p, 0, c->d_bneed.bytes));
    return VO_OK;
}

static void lane_free(vo_ctx* c) {
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->aux_stream) (void)hipStreamSynchronize(c->aux_stream);
    DevBuf* bufs[] = {&c->d_pyr, &c->d_blur, &c->d_tile_list, &c->d_seg_cnt, &c->d_seg_ofs, &c->d_key, &c->d_payload, &c->d_lpos, &c->d_rpos,
                      &c->d_cnt_fast, &c->d_cnt_sel1, &c->d_cnt_sel2, &c->d_nkp, &c->d_kp, &c->d_kplvl, &c->d_desc, &c->d_frame_status, &c->d_match, &c->d_raw, &c->d_bneed};
    for (DevBuf* b : bufs) dev_free(*b);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    for (int i = 0; i < 2; i++) if (c->ev_lane[i]) (void)hipEventDestroy(c->ev_lane[i]);
    if (c->aux_stream) (void)hipStreamDestroy(c->aux_stream);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->counted) g_live[c->device]--;
}

// buffers of the context that every lane aliases (geometry tables, the pair arrays, the pose stage's workspace)
static void lane_alias_shared(vo_ctx* l, const vo_ctx* c) {
    l->d_geom = c->d_geom; l->d_tiles = c->d_tiles; l->d_btiles = c->d_btiles; l->d_rtab = c->d_rtab; l->d_pat = c->d_pat;
    l->d_nmatch = c->d_nmatch; l->d_pair_status = c->d_pair_status; l->d_pair_nkp = c->d_pair_nkp;
    l->d_mq = c->d_mq; l->d_q1 = c->d_q1; l->d_q2 = c->d_q2;
}

static int ctx_init(vo_ctx* c) {
    const vo_orb_params& p = c->params;
    HIP_TRY(c, hipSetDevice(c->device));
    // extra lanes: one per chunk a pose wave holds, at most VO_MAX_LANES in all (a single-pair context has none)
    int nl = (int)std::min<long>(VO_MAX_LANES, ((long)c->max_pairs + c->chunk_pairs - 1) / c->chunk_pairs);
    if (const char* e = getenv("VO_LANES")) nl = std::max(1, std::min(8, atoi(e)));   // experiments
    c->single_lane = nl == 1;
    int rc = lane_init(c, nl <= 2);
    if (rc != VO_OK) return rc;
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming));
    HIP_TRY(c, hipStreamCreateWithFlags(&c->pose_stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; i++) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_done[i], hipEventDisableTiming));
    HIP_TRY(c, hipHostMalloc((void**)&c->h_result, 2 * (size_t)c->max_pairs * sizeof(vo_pair_result), hipHostMallocDefault));
    const long kc = c->kp_cap, MP = c->max_pairs;
    const long W = c->max_w, H = c->max_h;
    long tiles = 0, btiles = 0, rt = 0;
    {
        const double sf = (double)p.scale_factor;
        for (int l = 0; l < p.nlevels; l++) {
            const float sc = (float)std::pow(sf, (double)l);
            const long w = (long)nearbyintf(W * (1.f / sc)) + 1, h = (long)nearbyintf(H * (1.f / sc)) + 1;
            const long iw = std::max(w - 62, 0L), ih = std::max(h - 62, 0L);
            tiles += ((iw + VO_TILE_W - 1) / VO_TILE_W) * ((ih + VO_TILE_H - 1) / VO_TILE_H);
            btiles += ((w + 63) / 64) * ((h + 15) / 16);
            rt += w + h + 8;
        }
    }
    ALLOC(c, c->d_geom, sizeof(Geom));
    ALLOC(c, c->d_tiles, (tiles + 1) * sizeof(TileDesc));
    ALLOC(c, c->d_btiles, (btiles + 1) * sizeof(TileDesc));
    ALLOC(c, c->d_rtab, (rt + 1) * sizeof(ResizeCoef));
    ALLOC(c, c->d_nmatch, 2 * (size_t)MP * 4);
    ALLOC(c, c->d_pair_status, 2 * (size_t)MP * 4);
    ALLOC(c, c->d_pair_nkp, 2 * (size_t)MP * 8);
    ALLOC(c, c->d_mq, 2 * (size_t)MP * kc * sizeof(int4));
    ALLOC(c, c->d_q1, 2 * (size_t)MP * kc * sizeof(float2));
    ALLOC(c, c->d_q2, 2 * (size_t)MP * kc * sizeof(float2));
    ALLOC(c, c->d_mask, (size_t)MP * kc);
    ALLOC(c, c->d_pose, (size_t)MP * sizeof(PoseWork));
    ALLOC(c, c->d_hyp, (size_t)MP * VO_HYP_ROUND * 90 * sizeof(double));
    ALLOC(c, c->d_samples, (size_t)MP * VO_HYP_ROUND * 5 * 4);
    ALLOC(c, c->d_nmodels, (size_t)MP * VO_HYP_ROUND * 4);
    ALLOC(c, c->d_counts, (size_t)MP * VO_HYP_ROUND * 10 * 4);
    ALLOC(c, c->d_vote, (size_t)MP * 4 * 4);
    ALLOC(c, c->d_result, (size_t)MP * sizeof(vo_pair_result));
    ALLOC(c, c->d_misc, 4096);
    ALLOC(c, c->d_pat, 1024);
    HIP_TRY(c, hipMemcpy(c->d_pat.p, VO_ORB_PATTERN_31, 1024, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemset(c->d_pair_status.p, 0, c->d_pair_status.bytes));
    for (int i = 1; i < nl; i++) {
        vo_ctx* l = new vo_ctx();
        c->lanes.push_back(l);
        l->owner = c; l->device = c->device; l->params = c->params;
        l->max_w = c->max_w; l->max_h = c->max_h; l->chunk_pairs = c->chunk_pairs; l->max_pairs = c->max_pairs; l->max_frames = c->max_frames;
        if (c->device < 64) { g_live[c->device]++; l->counted = true; }
        rc = lane_init(l, nl <= 2);
        if (rc != VO_OK) return vo_fail(c, rc, l->err);
        lane_alias_shared(l, c);
    }
    // staging for the host-image entry points: one chunk per lane
    ALLOC(c, c->d_img, (size_t)c->max_frames * nl * W * H);
    c->err.reserve(256);       // error messages after this point do not touch the heap
    return vo_noalloc_init(c);
}

int vo_ctx_create(int device, int max_w, int max_h, int chunk_pairs, int max_pairs, const vo_orb_params* params, vo_ctx** out) {
    if (!out) return VO_ERR_INVALID;
    *out = nullptr;
    vo_ctx* c = new vo_ctx();
    *out = c;  // returned even on failure so that vo_last_error() can be read; destroy it afterwards
    c->device = device;
    if (params) c->params = *params;
    else vo_default_orb_params(&c->params);
    const vo_orb_params& p = c->params;
    if (max_w < 64 || max_h < 64 || max_w > 4095 || max_h > 4095 || chunk_pairs < 1 || max_pairs < 1 || chunk_pairs > 16384)
        return vo_fail(c, VO_ERR_INVALID, "vo_ctx_create: size arguments out of range");
    if (p.nlevels < 1 || p.nlevels > VO_MAX_LEVELS || p.patch_size != 31 || p.edge_threshold != 31 || p.nfeatures < 1 || p.nfeatures > 60000 ||
        p.fast_threshold < 1 || p.fast_threshold > 254 || (p.blur_mode != 0 && p.blur_mode != 1) || (p.trig_mode != 0 && p.trig_mode != 1) ||
        p.keypoint_capacity < 0 || p.keypoint_capacity > 65535)
        return vo_fail(c, VO_ERR_INVALID, "vo_ctx_create: unsupported ORB parameters");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device || device < 0)
        return vo_fail(c, VO_ERR_HIP, "vo_ctx_create: no such HIP device (this library has no CPU fallback)");
    c->max_w = max_w; c->max_h = max_h;
    c->chunk_pairs = chunk_pairs;
    c->max_pairs = std::max(max_pairs, chunk_pairs);
    c->max_frames = 2 * chunk_pairs;
    if (device < 64) { g_live[device]++; c->counted = true; }
    return ctx_init(c);
}

void vo_ctx_destroy(vo_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->pose_stream) (void)hipStreamSynchronize(c->pose_stream);
    for (vo_ctx* l : c->lanes) { lane_free(l); delete l; }
    lane_free(c);
    DevBuf* bufs[] = {&c->d_geom, &c->d_tiles, &c->d_btiles, &c->d_rtab, &c->d_img, &c->d_nmatch, &c->d_mq, &c->d_q1, &c->d_q2, &c->d_pose,
                      &c->d_hyp, &c->d_mask, &c->d_result, &c->d_misc, &c->d_pat, &c->d_pair_status, &c->d_samples, &c->d_nmodels, &c->d_counts,
                      &c->d_vote, &c->d_pair_nkp, &c->d_tri};
    for (DevBuf* b : bufs) dev_free(*b);
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->ev_in) (void)hipEventDestroy(c->ev_in);
    for (int i = 0; i < 2; i++) if (c->ev_done[i]) (void)hipEventDestroy(c->ev_done[i]);
    if (c->h_result) (void)hipHostFree(c->h_result);
    if (c->pose_stream) (void)hipStreamDestroy(c->pose_stream);
    delete c;
}

const char* vo_last_error(const vo_ctx* c) { return c ? c->err.c_str() : "null context"; }

int vo_set_debug_stop(vo_ctx* c, int stage) {
    if (!c || stage < 0 || stage > 2) return VO_ERR_INVALID;
    c->debug_stop = stage;
    for (vo_ctx* l : c->lanes) l->debug_stop = stage;
    return VO_OK;
}

int vo_set_kernel_timing(vo_ctx* c, int enable) {
    if (!c) return VO_ERR_INVALID;
    c->timing = enable != 0;
    for (int i = 0; i < VO_N_KERNELS; i++) { c->kernel_ms[i] = 0; c->kernel_launches[i] = 0; }
    c->ev_ids.clear();
    return VO_OK;
}
int vo_get_kernel_timing(vo_ctx* c, float ms[VO_N_KERNELS], int32_t launches[VO_N_KERNELS]) {
    if (!c || !ms || !launches) return VO_ERR_INVALID;
    for (int i = 0; i < VO_N_KERNELS; i++) { ms[i] = c->kernel_ms[i]; launches[i] = c->kernel_launches[i]; }
    return VO_OK;
}

// The synchronous entry points reuse the buffers a submission in flight (vo_pairs_submit_device) is working on.
#define REFUSE_IN_FLIGHT(c, name)                                                                                        \
    do {                                                                                                                 \
        if ((c)->n_submitted != (c)->n_waited)                                                                           \
            return vo_fail((c), VO_ERR_INVALID, name ": a submission is in flight on this context, call vo_pairs_wait first"); \
    } while (0)

// -------------------------------------------------------------------- detectAndCompute
// host frames -> tight device frames.  Contiguous rows (stride == w, the usual case) go as ONE linear copy: hipMemcpy2D from
// pageable memory copies row by row (measured 0.13 GB/s for 1241-byte rows against ~20 GB/s for the linear copy).
static hipError_t copy_frames_h2d(void* dst, const uint8_t* src, int w, int stride, size_t rows) {
    if (stride == w) return hipMemcpy(dst, src, (size_t)w * rows, hipMemcpyHostToDevice);
    return hipMemcpy2D(dst, w, src, stride, w, rows, hipMemcpyHostToDevice);
}

// The same copy enqueued on a lane's stream: it is ordered after the lane's previous chunk (whose kernels read the staging
// area it overwrites) and runs beside the other lanes' kernels.  From pinned host memory the call returns at once; from
// pageable memory it returns when the runtime has taken the source, and the host thread is then what bounds the rate.
static hipError_t copy_frames_h2d_async(void* dst, const uint8_t* src, int w, int stride, size_t rows, hipStream_t st) {
    if (stride == w) return hipMemcpyAsync(dst, src, (size_t)w * rows, hipMemcpyHostToDevice, st);
    return hipMemcpy2DAsync(dst, w, src, stride, w, rows, hipMemcpyHostToDevice, st);
}
// a lane's part of the staging buffer (max_frames frames of max_w x max_h each)
static uint8_t* lane_stage(vo_ctx* c, const vo_ctx* L) {
    size_t li = 0;
    for (size_t i = 0; i < c->lanes.size(); i++)
        if (c->lanes[i] == L) li = i + 1;
    return (uint8_t*)c->d_img.p + li * (size_t)c->max_frames * c->max_w * c->max_h;
}

int vo_orb_detect_compute(vo_ctx* c, const uint8_t* img, int w, int h, int stride, vo_keypoint* kps, uint8_t* desc, int cap, int* n) {
    if (!c) return VO_ERR_INVALID;
    if (!img || !n || w <= 0 || h <= 0 || stride < w) return vo_fail(c, VO_ERR_INVALID, "vo_orb_detect_compute: bad arguments");
    REFUSE_IN_FLIGHT(c, "vo_orb_detect_compute");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = vo_build_geometry(c, w, h);
    if (rc != VO_OK) return rc;
    hipStream_t st = c->stream;
    HIP_TRY(c, copy_frames_h2d(c->d_img.p, img, w, stride, (size_t)h));
    FrameSrc src{(const uint8_t*)c->d_img.p, (const uint8_t*)c->d_img.p, 1, w, (long)w * h};
    c->last_src = src; c->last_frames = 1; c->last_pairs = 0; c->last_lane = 0;
    rc = vo_launch_detect(c, src, 1, st);
    if (rc != VO_OK) return rc;
    int hn[2] = {0, 0};
    HIP_TRY(c, hipMemcpyAsync(&hn[0], c->d_nkp.p, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(&hn[1], c->d_frame_status.p, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    if (c->timing) kt_collect(c);
    if (c->debug_stop) { *n = 0; return VO_OK; }
    if (hn[1] != VO_OK) return vo_fail(c, hn[1], "vo_orb_detect_compute: keypoint list overflow (too many response ties)");
    *n = hn[0];
    if (hn[0] > cap) return vo_fail(c, VO_ERR_CAPACITY, "vo_orb_detect_compute: output capacity too small");
    if (kps && hn[0]) HIP_TRY(c, hipMemcpy(kps, c->d_kp.p, (size_t)hn[0] * sizeof(vo_keypoint), hipMemcpyDeviceToHost));
    if (desc && hn[0]) HIP_TRY(c, hipMemcpy(desc, c->d_desc.p, (size_t)hn[0] * 32, hipMemcpyDeviceToHost));
    return VO_OK;
}

// -------------------------------------------------------------------- matcher
static int match_host(vo_ctx* c, const uint8_t* q, int nq, const uint8_t* t, int nt, double ratio, bool want_raw) {
    if (!q || !t || nq < 0 || nt < 0) return vo_fail(c, VO_ERR_INVALID, "matcher: bad arguments");
    if (nq > c->kp_cap || nt > c->kp_cap) return vo_fail(c, VO_ERR_CAPACITY, "matcher: more descriptors than the context's keypoint capacity");
    if (nt < 2) return vo_fail(c, VO_ERR_TOO_FEW, "knnMatch(k=2) needs at least 2 train descriptors");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    c->slot = 0;
    REFUSE_IN_FLIGHT(c, "matcher");
    const long kc = c->kp_cap;
    int hn[2] = {nq, nt};
    HIP_TRY(c, hipMemcpyAsync(c->d_nkp.p, hn, 8, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemsetAsync(c->d_frame_status.p, 0, 8, st));
    if (nq) HIP_TRY(c, hipMemcpyAsync(c->d_desc.p, q, (size_t)nq * 32, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync((uint8_t*)c->d_desc.p + kc * 32, t, (size_t)nt * 32, hipMemcpyHostToDevice, st));
    c->last_pairs = 1; c->last_lane = 0; c->results_valid = false;
    int rc = vo_launch_match_impl(c, 1, 1, ratio, 0, want_raw ? (int4*)c->d_raw.p : nullptr, st);
    if (rc != VO_OK) return rc;
    HIP_TRY(c, hipStreamSynchronize(st));
    return VO_OK;
}

int vo_match_knn2(vo_ctx* c, const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx1, int32_t* d1, int32_t* idx2, int32_t* d2) {
    if (!c) return VO_ERR_INVALID;
    int rc = match_host(c, q, nq, t, nt, 0.45, true);
    if (rc != VO_OK) return rc;
    std::vector<int4> raw(std::max(nq, 1));
    if (nq) HIP_TRY(c, hipMemcpy(raw.data(), c->d_raw.p, (size_t)nq * sizeof(int4), hipMemcpyDeviceToHost));
    for (int i = 0; i < nq; i++) {
        if (idx1) idx1[i] = raw[i].x;
        if (d1) d1[i] = raw[i].y;
        if (idx2) idx2[i] = raw[i].z;
        if (d2) d2[i] = raw[i].w;
    }
    return VO_OK;
}

int vo_match_knn2_ratio(vo_ctx* c, const uint8_t* q, int nq, const uint8_t* t, int nt, double ratio, int32_t* qidx, int32_t* tidx, int32_t* d1,
                        int32_t* d2, int cap, int* n) {
    if (!c) return VO_ERR_INVALID;
    if (!n) return vo_fail(c, VO_ERR_INVALID, "vo_match_knn2_ratio: n is null");
    int rc = match_host(c, q, nq, t, nt, ratio, false);
    if (rc != VO_OK) return rc;
    int nm = 0;
    HIP_TRY(c, hipMemcpy(&nm, c->nmatch(), 4, hipMemcpyDeviceToHost));
    *n = nm;
    if (nm > cap) return vo_fail(c, VO_ERR_CAPACITY, "vo_match_knn2_ratio: output capacity too small");
    std::vector<int4> m(std::max(nm, 1));
    if (nm) HIP_TRY(c, hipMemcpy(m.data(), c->mq(), (size_t)nm * sizeof(int4), hipMemcpyDeviceToHost));
    for (int i = 0; i < nm; i++) {
        if (qidx) qidx[i] = m[i].x;
        if (tidx) tidx[i] = m[i].y;
        if (d1) d1[i] = m[i].z;
        if (d2) d2[i] = m[i].w;
    }
    return VO_OK;
}

// -------------------------------------------------------------------- pose
static int upload_points(vo_ctx* c, const float* q1, const float* q2, int n, hipStream_t st) {
    if (n > c->kp_cap) return vo_fail(c, VO_ERR_CAPACITY, "more points than the context's keypoint capacity");
    REFUSE_IN_FLIGHT(c, "pose entry point");
    int hn = n, zero = 0;
    c->slot = 0;
    c->results_valid = false;    // pair 0 of the pair arrays is overwritten; d_result still describes the last pose wave
    HIP_TRY(c, hipMemcpyAsync(c->nmatch(), &hn, 4, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(c->pair_status(), &zero, 4, hipMemcpyHostToDevice, st));
    if (n) {
        HIP_TRY(c, hipMemcpyAsync(c->q1(), q1, (size_t)n * 8, hipMemcpyHostToDevice, st));
        HIP_TRY(c, hipMemcpyAsync(c->q2(), q2, (size_t)n * 8, hipMemcpyHostToDevice, st));
    }
    HIP_TRY(c, hipStreamSynchronize(st));  // the host scalars above are stack variables
    return VO_OK;
}

int vo_find_essential_ransac(vo_ctx* c, const float* q1, const float* q2, int n, const double K[9], double prob, double thr, int max_iters,
                             double E[9], uint8_t* mask, int* n_inliers) {
    if (!c) return VO_ERR_INVALID;
    if (!q1 || !q2 || !K || !E || n < 0) return vo_fail(c, VO_ERR_INVALID, "vo_find_essential_ransac: bad arguments");
    if (!(prob > 0 && prob < 1)) return vo_fail(c, VO_ERR_INVALID, "vo_find_essential_ransac: confidence must be in (0,1)");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    int rc = upload_points(c, q1, q2, n, st);
    if (rc != VO_OK) return rc;
    c->last_pairs = 1;
    rc = vo_launch_ransac(c, 1, K, prob, thr, max_iters, st);
    if (rc != VO_OK) return rc;
    PoseWork w;
    HIP_TRY(c, hipMemcpyAsync(&w, c->d_pose.p, sizeof(w), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    if (n_inliers) *n_inliers = w.status == VO_OK ? w.max_good : 0;
    if (w.status == VO_ERR_TOO_FEW) return vo_fail(c, VO_ERR_TOO_FEW, "findEssentialMat needs at least 5 points");
    if (w.status != VO_OK) return vo_fail(c, w.status, "findEssentialMat: RANSAC found no model");
    memcpy(E, w.E, sizeof(w.E));
    if (mask && n) HIP_TRY(c, hipMemcpy(mask, c->d_mask.p, (size_t)n, hipMemcpyDeviceToHost));
    return VO_OK;
}

int vo_recover_pose(vo_ctx* c, const double E[9], const float* q1, const float* q2, int n, const double K[9], double dist_thresh, double R[9],
                    double t[3], int* n_good) {
    if (!c) return VO_ERR_INVALID;
    if (!E || !q1 || !q2 || !K || !R || !t || n < 0) return vo_fail(c, VO_ERR_INVALID, "vo_recover_pose: bad arguments");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    int rc = upload_points(c, q1, q2, n, st);
    if (rc != VO_OK) return rc;
    PoseWork w;
    memset(&w, 0, sizeof(w));
    memcpy(w.E, E, sizeof(w.E));
    w.status = VO_OK;
    HIP_TRY(c, hipMemcpy(c->d_pose.p, &w, sizeof(w), hipMemcpyHostToDevice));
    rc = vo_launch_recover(c, 1, K, dist_thresh, st);
    if (rc != VO_OK) return rc;
    HIP_TRY(c, hipMemcpyAsync(&w, c->d_pose.p, sizeof(w), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    memcpy(R, w.R, sizeof(w.R));
    memcpy(t, w.t, sizeof(w.t));
    if (n_good) *n_good = w.n_good;
    return VO_OK;
}

// get_pose (libs/vo_utils.py:136-145) as one call: findEssentialMat(prob 0.999, threshold 1.0, 1000 iterations), recoverPose
// (distance threshold 50) and form_transf, the points uploaded once and one read-back at the end.
int vo_get_pose(vo_ctx* c, const float* q1, const float* q2, int n, const double K[9], double T[16], int* n_inliers, int* n_good) {
    if (!c) return VO_ERR_INVALID;
    if (!q1 || !q2 || !K || !T || n < 0) return vo_fail(c, VO_ERR_INVALID, "vo_get_pose: bad arguments");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    int rc = upload_points(c, q1, q2, n, st);
    if (rc != VO_OK) return rc;
    c->last_pairs = 1;
    rc = vo_launch_ransac(c, 1, K, 0.999, 1.0, 1000, st);
    if (rc == VO_OK) rc = vo_launch_recover(c, 1, K, 50.0, st);     // a pair without a model is skipped by the kernels
    if (rc != VO_OK) return rc;
    PoseWork w;
    HIP_TRY(c, hipMemcpyAsync(&w, c->d_pose.p, sizeof(w), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    if (n_inliers) *n_inliers = w.status == VO_OK ? w.max_good : 0;
    if (n_good) *n_good = w.status == VO_OK ? w.n_good : 0;
    if (w.status == VO_ERR_TOO_FEW) return vo_fail(c, VO_ERR_TOO_FEW, "findEssentialMat needs at least 5 points");
    if (w.status != VO_OK) return vo_fail(c, w.status, "findEssentialMat: RANSAC found no model");
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) T[4 * i + j] = w.R[3 * i + j];
        T[4 * i + 3] = w.t[i];
    }
    T[12] = T[13] = T[14] = 0; T[15] = 1;
    return VO_OK;
}

int vo_triangulate(vo_ctx* c, const double P1[12], const double P2[12], const float* x1, const float* x2, int n, float* X4) {
    if (!c) return VO_ERR_INVALID;
    if (!P1 || !P2 || !x1 || !x2 || !X4 || n < 0) return vo_fail(c, VO_ERR_INVALID, "vo_triangulate: bad arguments");
    if (n == 0) return VO_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const size_t need = 256 + (size_t)n * (8 + 8 + 16);
    if (c->d_tri.bytes < need) {
        dev_free(c->d_tri);
        ALLOC(c, c->d_tri, need * 2);
    }
    uint8_t* base = (uint8_t*)c->d_tri.p;
    double* dP = (double*)base;
    float* dx1 = (float*)(base + 256);
    float* dx2 = dx1 + (size_t)n * 2;
    float* dX = dx2 + (size_t)n * 2;
    HIP_TRY(c, hipMemcpyAsync(dP, P1, 96, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(dP + 12, P2, 96, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(dx1, x1, (size_t)n * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(dx2, x2, (size_t)n * 8, hipMemcpyHostToDevice, st));
    int rc = vo_launch_triangulate(c, dP, dx1, dx2, n, dX, st);
    if (rc != VO_OK) return rc;
    HIP_TRY(c, hipMemcpyAsync(X4, dX, (size_t)n * 16, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    return VO_OK;
}

// -------------------------------------------------------------------- batched pairs
// pose stage (findEssentialMat + recoverPose) over the Pb pairs matched so far, results to host
static int pose_stage(vo_ctx* c, int Pb, const double K[9], vo_pair_result* out, hipStream_t st, const std::vector<hipEvent_t>& after = {}) {
    c->last_pairs = Pb;
    c->results_valid = false;
    for (hipEvent_t e : after) HIP_TRY(c, hipStreamWaitEvent(st, e, 0));
    kt_begin(c, st, VO_K_RANSAC);
    int rc = vo_launch_ransac(c, Pb, K, 0.999, 1.0, 1000, st);
    kt_end(c, st);
    if (rc != VO_OK) return rc;
    kt_begin(c, st, VO_K_RECOVER);
    rc = vo_launch_recover(c, Pb, K, 50.0, st);
    kt_end(c, st);
    if (rc != VO_OK) return rc;
    hipLaunchKernelGGL(k_results, dim3((Pb + 63) / 64), dim3(64), 0, st, (const PoseWork*)c->d_pose.p, (const int*)c->nmatch(),
                       (const int2*)c->pair_nkp(), Pb, (vo_pair_result*)c->d_result.p);
    HIP_TRY(c, hipMemcpyAsync(out, c->d_result.p, (size_t)Pb * sizeof(vo_pair_result), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    if (c->timing) kt_collect(c);
    c->results_valid = true;    // d_result now describes the pair arrays of c->slot
    return VO_OK;
}

// Stream-ordering contract of the *_device entry points: the work is ordered after everything the caller has enqueued on
// `stream` (hipStream_t), or, with stream == NULL, on the device's default stream -- the stream a framework's tensor ops
// (torch's default stream) produce the images on -- so un-synchronised producers are safe.  The kernels themselves run on
// the context's private streams (one per lane + the pose stream); the call returns after the results are in `out`.
static int order_after(vo_ctx* c, hipStream_t producer, const std::vector<hipStream_t>& consumers) {
    HIP_TRY(c, hipEventRecord(c->ev_in, producer));
    for (hipStream_t s : consumers)
        if (s != producer) HIP_TRY(c, hipStreamWaitEvent(s, c->ev_in, 0));
    return VO_OK;
}

static int propagate_geometry(vo_ctx* c) {
    for (vo_ctx* l : c->lanes) { l->geom = c->geom; l->geom_valid = c->geom_valid; }
    return VO_OK;
}

// B pairs in pose waves ("batches") of max_pairs pairs, each made of chunks of `cp` pairs.  chunk(L, st, b0, p0, Pc)
// enqueues detect + match of pairs [b0 + p0, b0 + p0 + Pc) on lane L / stream st, writing rows p0.. of the pair arrays.
// Consecutive chunks go to different lanes (round-robin), all enqueued by this one host thread without a
// synchronisation.  The pose stage of batch b runs on the pose stream after the lanes' events, and its host-synchronised
// RANSAC rounds are issued only after detect + match of batch b+1 have been enqueued, so they overlap them; the pair
// arrays the two stages share exist twice (vo_ctx::slot).  In the timing mode everything runs on ONE stream, one
// kernel at a time (HIP events bracket every launch).
// The host-image entry points put a chunk's host-to-device copy at the head of its chunk() on the same lane stream.
static int run_batches(vo_ctx* c, int B, int cp, const double K[9], vo_pair_result* out, hipStream_t user,
                       const std::function<int(vo_ctx*, hipStream_t, int, int, int)>& chunk) {
    if (c->n_submitted != c->n_waited) return vo_fail(c, VO_ERR_INVALID, "a submission is in flight on this context: call vo_pairs_wait first");
    const bool serial = c->timing;
    const int nl = serial ? 1 : 1 + (int)c->lanes.size();
    auto lane = [&](int i) { return i == 0 ? c : c->lanes[i - 1]; };
    std::vector<hipStream_t> lst(nl);
    for (int i = 0; i < nl; i++) lst[i] = lane(i)->stream;
    if (serial && user) lst[0] = user;
    hipStream_t pst = serial ? lst[0] : c->pose_stream;
    int rc = order_after(c, user, lst);
    if (rc != VO_OK) return rc;
    auto drain = [&](int r) {
        for (hipStream_t s : lst) (void)hipStreamSynchronize(s);
        (void)hipStreamSynchronize(pst);
        c->results_valid = false;
        return r;
    };
    struct Pending { int b0 = -1, Pb = 0, slot = 0; std::vector<hipEvent_t> ev; } pend;
    int nb = 0, nchunk = 0;
    static const bool hosttime = getenv("VO_HOSTTIME") != nullptr;   // diagnostic: how long the host needs to enqueue a call
    const auto ht0 = std::chrono::steady_clock::now();
    for (int b0 = 0; b0 < B; b0 += c->max_pairs, nb++) {
        const int Pb = std::min(c->max_pairs, B - b0);
        const int slot = nb & 1;
        c->slot = slot;
        unsigned used = 0;
        for (int p0 = 0; p0 < Pb; p0 += cp, nchunk++) {
            const int li = nchunk % nl;
            vo_ctx* L = lane(li);
            L->slot = slot;
            rc = chunk(L, lst[li], b0, p0, std::min(cp, Pb - p0));
            if (rc != VO_OK) return drain(L != c ? vo_fail(c, rc, L->err) : rc);
            c->last_lane = li;
            used |= 1u << li;
            if (serial) {
                if (hipStreamSynchronize(lst[0]) != hipSuccess) return vo_fail(c, VO_ERR_HIP, "hipStreamSynchronize failed");
                kt_collect(c);
            }
        }
        if (serial) {
            rc = pose_stage(c, Pb, K, out + b0, pst);
            if (rc != VO_OK) return drain(rc);
            continue;
        }
        Pending cur;
        cur.b0 = b0; cur.Pb = Pb; cur.slot = slot;
        for (int li = 0; li < nl; li++)
            if (used >> li & 1) {
                if (hipEventRecord(lane(li)->ev_lane[slot], lst[li]) != hipSuccess) return drain(vo_fail(c, VO_ERR_HIP, "hipEventRecord failed"));
                cur.ev.push_back(lane(li)->ev_lane[slot]);
            }
        if (pend.b0 >= 0) {
            c->slot = pend.slot;
            rc = pose_stage(c, pend.Pb, K, out + pend.b0, pst, pend.ev);
            c->slot = slot;
            if (rc != VO_OK) return drain(rc);
        }
        pend = cur;
    }
    const auto ht1 = std::chrono::steady_clock::now();
    if (pend.b0 >= 0) {
        c->slot = pend.slot;
        rc = pose_stage(c, pend.Pb, K, out + pend.b0, pst, pend.ev);
        if (rc != VO_OK) return drain(rc);
    }
    if (hosttime) {
        const auto ht2 = std::chrono::steady_clock::now();
        fprintf(stderr, "[vo] %d pairs, %d chunks: enqueue %.3f ms, pose stage + wait %.3f ms\n", B, nchunk,
                std::chrono::duration<double, std::milli>(ht1 - ht0).count(), std::chrono::duration<double, std::milli>(ht2 - ht1).count());
    }
    return VO_OK;
}

int vo_pairs_run_device(vo_ctx* c, const uint8_t* imgsA, const uint8_t* imgsB, int B, int w, int h, int stride, const double K[9], double ratio,
                        vo_pair_result* out, void* stream) {
    if (!c) return VO_ERR_INVALID;
    if (!imgsA || !imgsB || !K || !out || B < 0 || stride < w) return vo_fail(c, VO_ERR_INVALID, "vo_pairs_run_device: bad arguments");
    if (B == 0) return VO_OK;
    REFUSE_IN_FLIGHT(c, "vo_pairs_run_device");   // before the geometry tables are touched: the lanes in flight read them
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = vo_build_geometry(c, w, h);
    if (rc != VO_OK) return rc;
    propagate_geometry(c);
    const long fs = (long)stride * h;
    return run_batches(c, B, c->chunk_pairs, K, out, (hipStream_t)stream, [&](vo_ctx* L, hipStream_t st, int b0, int p0, int Pc) {
        FrameSrc src{imgsA + (long)(b0 + p0) * fs, imgsB + (long)(b0 + p0) * fs, Pc, stride, fs};
        L->last_src = src; L->last_frames = 2 * Pc;
        int r = vo_launch_detect(L, src, 2 * Pc, st);
        if (r != VO_OK) return r;
        return vo_launch_match_impl(L, Pc, Pc, ratio, p0, nullptr, st);
    });
}

// ---- asynchronous form of vo_pairs_run_device: submit returns as soon as the work is enqueued, wait returns the records.
// With two submissions in flight the pose stage of call k (pose stream; a host-free RANSAC schedule) overlaps detect +
// match of call k+1 (lane streams): what a strong-scaling rank with a small per-step share needs (512 pairs per step:
// 42.8 k pairs/s with the synchronous call, whose 3 ms pose tail nothing overlaps).
static int submit_impl(vo_ctx* c, const char* name, int B, int cp, int w, int h, const double K[9], vo_pair_result* out, hipStream_t user,
                       const std::function<int(vo_ctx*, hipStream_t, int, int)>& chunk) {
    if (B > c->max_pairs) return vo_fail(c, VO_ERR_INVALID, std::string(name) + ": more pairs than max_pairs (one pose wave per submission)");
    if (c->timing) return vo_fail(c, VO_ERR_INVALID, std::string(name) + ": not available in the kernel-timing mode");
    if (c->n_submitted - c->n_waited >= 2) return vo_fail(c, VO_ERR_INVALID, std::string(name) + ": two submissions are in flight, call vo_pairs_wait first");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->n_submitted != c->n_waited && c->geom_valid && (c->geom.w != w || c->geom.h != h))
        return vo_fail(c, VO_ERR_INVALID, std::string(name) + ": the frame size cannot change while a submission is in flight");
    int rc = vo_build_geometry(c, w, h);
    if (rc != VO_OK) return rc;
    propagate_geometry(c);
    const int slot = (int)(c->n_submitted & 1);
    const int nl = 1 + (int)c->lanes.size();
    auto lane = [&](int i) { return i == 0 ? c : c->lanes[i - 1]; };
    std::vector<hipStream_t> lst(nl);
    for (int i = 0; i < nl; i++) lst[i] = lane(i)->stream;
    rc = order_after(c, user, lst);
    if (rc != VO_OK) return rc;
    hipStream_t pst = c->pose_stream;
    // A failure after the first enqueue leaves kernels writing this slot's pair arrays while the context looks idle
    // (n_submitted is not advanced): wait for them before returning, as run_batches does on its error paths.
    auto drain = [&](int r) {
        for (hipStream_t s : lst) (void)hipStreamSynchronize(s);
        (void)hipStreamSynchronize(pst);
        return r;
    };
    unsigned used = 0;
    int nchunk = 0;
    for (int p0 = 0; p0 < B; p0 += cp, nchunk++) {
        const int li = (int)((c->n_submitted * 3 + nchunk) % nl);     // successive submissions start on different lanes
        vo_ctx* L = lane(li);
        L->slot = slot;
        rc = chunk(L, lst[li], p0, std::min(cp, B - p0));
        if (rc != VO_OK) return drain(L != c ? vo_fail(c, rc, L->err) : rc);
        c->last_lane = li;
        used |= 1u << li;
    }
    for (int li = 0; li < nl; li++)
        if (used >> li & 1) {
            if (hipEventRecord(lane(li)->ev_lane[slot], lst[li]) != hipSuccess || hipStreamWaitEvent(pst, lane(li)->ev_lane[slot], 0) != hipSuccess)
                return drain(vo_fail(c, VO_ERR_HIP, std::string(name) + ": could not order the pose stream after the lanes"));
        }
    c->slot = slot;
    rc = vo_launch_ransac(c, B, K, 0.999, 1.0, 1000, pst, true);
    if (rc == VO_OK) rc = vo_launch_recover(c, B, K, 50.0, pst);
    if (rc != VO_OK) return drain(rc);
    hipLaunchKernelGGL(k_results, dim3((B + 63) / 64), dim3(64), 0, pst, (const PoseWork*)c->d_pose.p, (const int*)c->nmatch(),
                       (const int2*)c->pair_nkp(), B, (vo_pair_result*)c->d_result.p);
    if (hipMemcpyAsync(c->h_result + (size_t)slot * c->max_pairs, c->d_result.p, (size_t)B * sizeof(vo_pair_result), hipMemcpyDeviceToHost, pst) != hipSuccess ||
        hipEventRecord(c->ev_done[slot], pst) != hipSuccess)
        return drain(vo_fail(c, VO_ERR_HIP, std::string(name) + ": could not enqueue the read-back"));
    c->results_valid = false;   // d_result is being rewritten; vo_pairs_wait marks it valid again
    c->sub[slot].live = true; c->sub[slot].B = B; c->sub[slot].out = out;
    c->n_submitted++;
    return VO_OK;
}

int vo_pairs_submit_device(vo_ctx* c, const uint8_t* imgsA, const uint8_t* imgsB, int B, int w, int h, int stride, const double K[9], double ratio,
                           vo_pair_result* out, void* stream) {
    if (!c) return VO_ERR_INVALID;
    if (!imgsA || !imgsB || !K || !out || B < 1 || stride < w) return vo_fail(c, VO_ERR_INVALID, "vo_pairs_submit_device: bad arguments");
    const long fs = (long)stride * h;
    return submit_impl(c, "vo_pairs_submit_device", B, c->chunk_pairs, w, h, K, out, (hipStream_t)stream, [&](vo_ctx* L, hipStream_t st, int p0, int Pc) {
        FrameSrc src{imgsA + (long)p0 * fs, imgsB + (long)p0 * fs, Pc, stride, fs};
        L->last_src = src; L->last_frames = 2 * Pc;
        int r = vo_launch_detect(L, src, 2 * Pc, st);
        if (r != VO_OK) return r;
        return vo_launch_match_impl(L, Pc, Pc, ratio, p0, nullptr, st);
    });
}

// the same for a sequence of F frames (F - 1 <= max_pairs consecutive pairs, each frame extracted once per chunk)
int vo_sequence_submit_device(vo_ctx* c, const uint8_t* imgs, int F, int w, int h, int stride, const double K[9], double ratio,
                              vo_pair_result* out, void* stream) {
    if (!c) return VO_ERR_INVALID;
    if (!imgs || !K || !out || F < 2 || stride < w) return vo_fail(c, VO_ERR_INVALID, "vo_sequence_submit_device: bad arguments");
    const long fs = (long)stride * h;
    const int cp = std::min(c->chunk_pairs, 2 * c->chunk_pairs - 1);
    return submit_impl(c, "vo_sequence_submit_device", F - 1, cp, w, h, K, out, (hipStream_t)stream, [&](vo_ctx* L, hipStream_t st, int p0, int Pc) {
        const uint8_t* first = imgs + (long)p0 * fs;
        FrameSrc src{first, first, Pc + 1, stride, fs};
        L->last_src = src; L->last_frames = Pc + 1;
        int r = vo_launch_detect(L, src, Pc + 1, st);
        if (r != VO_OK) return r;
        return vo_launch_match_impl(L, Pc, 1, ratio, p0, nullptr, st);
    });
}

// Waits for the OLDEST submission in flight and writes its records to the `out` it was submitted with.
int vo_pairs_wait(vo_ctx* c) {
    if (!c) return VO_ERR_INVALID;
    if (c->n_submitted == c->n_waited) return vo_fail(c, VO_ERR_INVALID, "vo_pairs_wait: nothing in flight");
    HIP_TRY(c, hipSetDevice(c->device));
    const int slot = (int)(c->n_waited & 1);
    const hipError_t e = hipEventSynchronize(c->ev_done[slot]);
    // the submission is retired whatever the outcome: a context whose wait failed must not stay "in flight" for good
    c->sub[slot].live = false;
    c->n_waited++;
    if (e != hipSuccess) {
        c->results_valid = false; c->last_pairs = 0;
        (void)hipDeviceSynchronize();     // nothing of the failed submission may still be writing the pair arrays
        return vo_fail(c, VO_ERR_HIP, std::string("vo_pairs_wait: ") + hipGetErrorString(e) + " (the submission is dropped, its records are not written)");
    }
    memcpy(c->sub[slot].out, c->h_result + (size_t)slot * c->max_pairs, (size_t)c->sub[slot].B * sizeof(vo_pair_result));
    c->slot = slot; c->last_pairs = c->sub[slot].B;     // vo_debug_read / vo_pairs_triangulate see this submission's pair arrays
    // d_result holds the records of the submission ENQUEUED last; it describes this slot only when none is left behind it
    c->results_valid = c->n_submitted == c->n_waited;
    return VO_OK;
}

// A sequence of F frames: pair i is (frame i, frame i+1), as the loop of vo.py:48-190 visits them.
// Every frame is extracted once per chunk (the reference extracts each frame twice, libs/vo_utils.py:94-95);
// consecutive chunks share one frame.  Results are those of vo_pairs_run_device on the same pairs.
int vo_sequence_run_device(vo_ctx* c, const uint8_t* imgs, int F, int w, int h, int stride, const double K[9], double ratio,
                           vo_pair_result* out, void* stream) {
    if (!c) return VO_ERR_INVALID;
    if (!imgs || !K || !out || F < 0 || stride < w) return vo_fail(c, VO_ERR_INVALID, "vo_sequence_run_device: bad arguments");
    if (F < 2) return VO_OK;
    REFUSE_IN_FLIGHT(c, "vo_sequence_run_device");   // before the geometry tables are touched: the lanes in flight read them
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = vo_build_geometry(c, w, h);
    if (rc != VO_OK) return rc;
    propagate_geometry(c);
    const long fs = (long)stride * h;
    // a chunk of Pc pairs needs Pc+1 frames of the 2*chunk_pairs-frame workspace and Pc rows of the match workspace
    const int cp = std::min(c->chunk_pairs, 2 * c->chunk_pairs - 1);
    return run_batches(c, F - 1, cp, K, out, (hipStream_t)stream, [&](vo_ctx* L, hipStream_t st, int b0, int p0, int Pc) {
        const uint8_t* first = imgs + (long)(b0 + p0) * fs;
        FrameSrc src{first, first, Pc + 1, stride, fs};
        L->last_src = src; L->last_frames = Pc + 1;
        int r = vo_launch_detect(L, src, Pc + 1, st);
        if (r != VO_OK) return r;
        return vo_launch_match_impl(L, Pc, 1, ratio, p0, nullptr, st);   // train frame of pair p = frame p + 1
    });
}

// get_matches (libs/vo_utils.py:83-110) in one call: both frames go up together, are extracted in one launch wave and
// matched on the device; only the matched pixel positions come back.
int vo_get_matches(vo_ctx* c, const uint8_t* imgA, const uint8_t* imgB, int w, int h, int stride, double ratio, float* q1, float* q2, int cap, int* n) {
    if (!c) return VO_ERR_INVALID;
    if (!imgA || !imgB || !n || w <= 0 || h <= 0 || stride < w || cap < 0) return vo_fail(c, VO_ERR_INVALID, "vo_get_matches: bad arguments");
    REFUSE_IN_FLIGHT(c, "vo_get_matches");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = vo_build_geometry(c, w, h);
    if (rc != VO_OK) return rc;
    propagate_geometry(c);
    const long fs = (long)w * h;
    if ((size_t)(2 * fs) > c->d_img.bytes) return vo_fail(c, VO_ERR_CAPACITY, "vo_get_matches: staging buffer too small");
    uint8_t* dA = (uint8_t*)c->d_img.p;
    HIP_TRY(c, copy_frames_h2d(dA, imgA, w, stride, (size_t)h));
    HIP_TRY(c, copy_frames_h2d(dA + fs, imgB, w, stride, (size_t)h));
    hipStream_t st = c->stream;
    c->slot = 0; c->last_lane = 0; c->last_pairs = 1; c->results_valid = false;
    FrameSrc src{dA, dA + fs, 1, w, fs};
    c->last_src = src; c->last_frames = 2;
    rc = vo_launch_detect(c, src, 2, st);
    if (rc != VO_OK) return rc;
    rc = vo_launch_match_impl(c, 1, 1, ratio, 0, nullptr, st);
    if (rc != VO_OK) return rc;
    int hs[2] = {0, 0};
    HIP_TRY(c, hipMemcpyAsync(&hs[0], c->nmatch(), 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(&hs[1], c->pair_status(), 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    if (c->timing) kt_collect(c);
    *n = 0;
    if (hs[1] == VO_ERR_TOO_FEW) return vo_fail(c, VO_ERR_TOO_FEW, "knnMatch(k=2) needs at least 2 train descriptors");
    if (hs[1] != VO_OK) return vo_fail(c, hs[1], "vo_get_matches: keypoint list overflow (too many response ties)");
    *n = hs[0];
    if (hs[0] > cap) return vo_fail(c, VO_ERR_CAPACITY, "vo_get_matches: output capacity too small");
    if (hs[0] && q1) HIP_TRY(c, hipMemcpy(q1, c->q1(), (size_t)hs[0] * 8, hipMemcpyDeviceToHost));
    if (hs[0] && q2) HIP_TRY(c, hipMemcpy(q2, c->q2(), (size_t)hs[0] * 8, hipMemcpyDeviceToHost));
    return VO_OK;
}

// Row T for a whole wave of pairs (vo.py:80-93): the ratio-test matches of the last pose wave are still on the device;
// pair p is triangulated with its own P1[p], P2[p] (the caller chains the poses: K @ prev_pose[:3], K @ cur_pose[:3]).
int vo_pairs_triangulate(vo_ctx* c, int P, const double* P1, const double* P2, float* X, float* q1, float* q2, long cap, int64_t* offsets) {
    if (!c) return VO_ERR_INVALID;
    if (!P1 || !P2 || !offsets || P < 0 || cap < 0) return vo_fail(c, VO_ERR_INVALID, "vo_pairs_triangulate: bad arguments");
    REFUSE_IN_FLIGHT(c, "vo_pairs_triangulate");
    // the per-pair status comes from d_result, which only the batched pose waves write: after a single-pair entry point
    // (vo_get_matches, vo_get_pose, the matcher, findEssentialMat ...) it would be a stale record of an earlier wave
    if (!c->results_valid) return vo_fail(c, VO_ERR_INVALID, "vo_pairs_triangulate: the last call on this context was not a batched pose wave (vo_pairs_run*, vo_sequence_run*, vo_pairs_wait)");
    if (P > c->last_pairs) return vo_fail(c, VO_ERR_INVALID, "vo_pairs_triangulate: more pairs than the last pose wave held");
    offsets[0] = 0;
    if (P == 0) return VO_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    std::vector<int> nm(P), stt(P);
    HIP_TRY(c, hipMemcpy(nm.data(), c->nmatch(), (size_t)P * 4, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy2D(stt.data(), 4, (const uint8_t*)c->d_result.p + offsetof(vo_pair_result, status), sizeof(vo_pair_result), 4, P, hipMemcpyDeviceToHost));
    std::vector<long> off(P + 1, 0);
    for (int p = 0; p < P; p++) off[p + 1] = off[p] + (stt[p] == VO_OK ? nm[p] : 0);   // a failed pair (no pose to chain) has no points
    for (int p = 0; p <= P; p++) offsets[p] = off[p];
    const long N = off[P];
    if (N > cap) return vo_fail(c, VO_ERR_CAPACITY, "vo_pairs_triangulate: output capacity too small");
    if (N == 0) return VO_OK;
    if (!X) return vo_fail(c, VO_ERR_INVALID, "vo_pairs_triangulate: X is null");
    const size_t bPm = ((size_t)P * 24 * 8 + 255) / 256 * 256, bOff = ((size_t)(P + 1) * 8 + 255) / 256 * 256, bX = ((size_t)N * 12 + 255) / 256 * 256,
                 bQ = ((size_t)N * 8 + 255) / 256 * 256;
    const size_t need = bPm + bOff + bX + 2 * bQ;
    if (c->d_tri.bytes < need) {
        dev_free(c->d_tri);
        ALLOC(c, c->d_tri, need + need / 2);
    }
    uint8_t* base = (uint8_t*)c->d_tri.p;
    double* dPm = (double*)base;
    long* dOff = (long*)(base + bPm);
    float* dX = (float*)(base + bPm + bOff);
    float2* dq1 = (float2*)(base + bPm + bOff + bX);
    float2* dq2 = (float2*)(base + bPm + bOff + bX + bQ);
    std::vector<double> Pm((size_t)P * 24);
    for (int p = 0; p < P; p++) {
        memcpy(&Pm[(size_t)p * 24], P1 + (size_t)p * 12, 96);
        memcpy(&Pm[(size_t)p * 24 + 12], P2 + (size_t)p * 12, 96);
    }
    HIP_TRY(c, hipMemcpyAsync(dPm, Pm.data(), Pm.size() * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(dOff, off.data(), off.size() * 8, hipMemcpyHostToDevice, st));
    int rc = vo_launch_triangulate_pairs(c, P, dPm, dOff, dX, q1 ? dq1 : nullptr, q2 ? dq2 : nullptr, st);
    if (rc != VO_OK) return rc;
    HIP_TRY(c, hipMemcpyAsync(X, dX, (size_t)N * 12, hipMemcpyDeviceToHost, st));
    if (q1) HIP_TRY(c, hipMemcpyAsync(q1, dq1, (size_t)N * 8, hipMemcpyDeviceToHost, st));
    if (q2) HIP_TRY(c, hipMemcpyAsync(q2, dq2, (size_t)N * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));   // Pm / off are stack-owned: the copies must have left them
    return VO_OK;
}

int vo_sequence_run(vo_ctx* c, const uint8_t* imgs, int F, int w, int h, int stride, const double K[9], double ratio, vo_pair_result* out) {
    if (!c) return VO_ERR_INVALID;
    if (!imgs || !K || !out || F < 0 || stride < w) return vo_fail(c, VO_ERR_INVALID, "vo_sequence_run: bad arguments");
    if (F < 2) return VO_OK;
    REFUSE_IN_FLIGHT(c, "vo_sequence_run");   // before the geometry tables are touched: the lanes in flight read them
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = vo_build_geometry(c, w, h);
    if (rc != VO_OK) return rc;
    propagate_geometry(c);
    // frames are staged through device memory chunk by chunk, each copy on its lane's stream (n pairs = n + 1 frames: the
    // frame two chunks share is copied twice)
    const long fs = (long)w * h;
    const int cp = std::min(c->chunk_pairs, c->max_frames - 1);
    return run_batches(c, F - 1, cp, K, out, nullptr, [&](vo_ctx* L, hipStream_t st, int b0, int p0, int Pc) {
        uint8_t* first = lane_stage(c, L);
        if (copy_frames_h2d_async(first, imgs + (long)(b0 + p0) * stride * h, w, stride, (size_t)h * (Pc + 1), st) != hipSuccess)
            return vo_fail(L, VO_ERR_HIP, "vo_sequence_run: host-to-device copy failed");
        FrameSrc src{first, first, Pc + 1, w, fs};
        L->last_src = src; L->last_frames = Pc + 1;
        int r = vo_launch_detect(L, src, Pc + 1, st);
        if (r != VO_OK) return r;
        return vo_launch_match_impl(L, Pc, 1, ratio, p0, nullptr, st);
    });
}

int vo_pairs_run(vo_ctx* c, const uint8_t* imgsA, const uint8_t* imgsB, int B, int w, int h, int stride, const double K[9], double ratio,
                 vo_pair_result* out) {
    if (!c) return VO_ERR_INVALID;
    if (!imgsA || !imgsB || !K || !out || B < 0 || stride < w) return vo_fail(c, VO_ERR_INVALID, "vo_pairs_run: bad arguments");
    if (B == 0) return VO_OK;
    REFUSE_IN_FLIGHT(c, "vo_pairs_run");   // before the geometry tables are touched: the lanes in flight read them
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = vo_build_geometry(c, w, h);
    if (rc != VO_OK) return rc;
    propagate_geometry(c);
    // frames are staged through device memory chunk by chunk, each copy on its lane's stream: A frames, then B frames
    const long fs = (long)w * h;
    return run_batches(c, B, c->chunk_pairs, K, out, nullptr, [&](vo_ctx* L, hipStream_t st, int b0, int p0, int Pc) {
        uint8_t* dA = lane_stage(c, L);
        uint8_t* dB = dA + (long)Pc * fs;
        if (copy_frames_h2d_async(dA, imgsA + (long)(b0 + p0) * stride * h, w, stride, (size_t)h * Pc, st) != hipSuccess ||
            copy_frames_h2d_async(dB, imgsB + (long)(b0 + p0) * stride * h, w, stride, (size_t)h * Pc, st) != hipSuccess)
            return vo_fail(L, VO_ERR_HIP, "vo_pairs_run: host-to-device copy failed");
        FrameSrc src{dA, dB, Pc, w, fs};
        L->last_src = src; L->last_frames = 2 * Pc;
        int r = vo_launch_detect(L, src, 2 * Pc, st);
        if (r != VO_OK) return r;
        return vo_launch_match_impl(L, Pc, Pc, ratio, p0, nullptr, st);
    });
}

// -------------------------------------------------------------------- stage read-back
int vo_debug_minimal_solver(vo_ctx* c, int mode, const double* in, int n, double* out) {
    if (!c) return VO_ERR_INVALID;
    REFUSE_IN_FLIGHT(c, "vo_debug_minimal_solver");
    if (mode < 0 || mode > 3 || !in || !out || n < 0) return vo_fail(c, VO_ERR_INVALID, "vo_debug_minimal_solver: bad arguments");
    if (n == 0) return VO_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t in_b = (size_t)n * (mode % 2 == 0 ? 11 : 20) * 8, out_b = (size_t)n * (mode % 2 == 0 ? 20 : 91) * 8;
    DevBuf d_in, d_out;   // a test hook: its buffers live for the call only
    ALLOC(c, d_in, in_b);
    int rc = dev_alloc(c, d_out, out_b);
    if (rc != VO_OK) { dev_free(d_in); return rc; }
    if (hipMemcpyAsync(d_in.p, in, in_b, hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = vo_fail(c, VO_ERR_HIP, "vo_debug_minimal_solver: copy in");
    if (rc == VO_OK) rc = vo_launch_debug_minimal(c, mode, (const double*)d_in.p, n, (double*)d_out.p, c->stream);
    if (rc == VO_OK && hipMemcpyAsync(out, d_out.p, out_b, hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = vo_fail(c, VO_ERR_HIP, "vo_debug_minimal_solver: copy out");
    if (hipStreamSynchronize(c->stream) != hipSuccess && rc == VO_OK) rc = vo_fail(c, VO_ERR_HIP, "vo_debug_minimal_solver: kernel failed");
    dev_free(d_in); dev_free(d_out);
    return rc;
}

long vo_debug_read(vo_ctx* c, int what, int index, int level, void* out, long bytes) {
    if (!c || (!c->geom_valid && what <= VO_DBG_DESCRIPTORS)) return VO_ERR_INVALID;
    if (hipSetDevice(c->device) != hipSuccess) return VO_ERR_HIP;
    (void)hipDeviceSynchronize();
    const Geom& g = c->geom;
    const vo_ctx* ln = c->last_lane == 0 ? c : c->lanes[c->last_lane - 1];   // chunk-level buffers: the lane of the last chunk
    const long kc = c->kp_cap;
    auto copy = [&](const void* src, long have) -> long {
        if (out && have > 0 && bytes > 0 && hipMemcpy(out, src, (size_t)std::min(have, bytes), hipMemcpyDeviceToHost) != hipSuccess) return VO_ERR_HIP;
        return have;
    };
    auto read_int = [&](const void* p, long idx) {
        int v = 0;
        (void)hipMemcpy(&v, (const int*)p + idx, 4, hipMemcpyDeviceToHost);
        return v;
    };
    switch (what) {
        case VO_DBG_PYR_LEVEL:
        case VO_DBG_BLUR_LEVEL: {
            if (index < 0 || index >= ln->last_frames || level < 0 || level >= g.nlevels) return VO_ERR_INVALID;
            const LevelGeom& L = g.lv[level];
            const uint8_t* src;
            long pitch;
            if (what == VO_DBG_BLUR_LEVEL) { src = (const uint8_t*)ln->d_blur.p + (long)index * g.blur_frame + L.blur_ofs; pitch = L.bpitch; }
            else if (level == 0) {
                const FrameSrc& s = ln->last_src;
                src = index < s.nA ? s.a + (long)index * s.frame_stride : s.b + (long)(index - s.nA) * s.frame_stride;
                pitch = s.pitch;
            } else { src = (const uint8_t*)ln->d_pyr.p + (long)index * g.pyr_frame + L.pyr_ofs; pitch = L.pitch; }
            const long have = (long)L.w * L.h;
            if (out && bytes >= have && hipMemcpy2D(out, L.w, src, pitch, L.w, L.h, hipMemcpyDeviceToHost) != hipSuccess) return VO_ERR_HIP;
            return have;
        }
        case VO_DBG_FAST_LIST:
        case VO_DBG_SEL1_LIST:
        case VO_DBG_SEL1_HARRIS: {
            if (index < 0 || index >= ln->last_frames || level < 0 || level >= g.nlevels) return VO_ERR_INVALID;
            const int n = read_int(what == VO_DBG_FAST_LIST ? ln->d_cnt_fast.p : ln->d_cnt_sel1.p, (long)index * VO_MAX_LEVELS + level);
            const long off = (long)index * g.cand_frame + g.lv[level].cand_ofs;
            if (what == VO_DBG_SEL1_HARRIS) return copy((const float*)ln->d_key.p + off, (long)n * 4);
            const long have = (long)n * 12;
            if (out && bytes >= have && n > 0) {
                std::vector<uint32_t> pk(n);
                if (hipMemcpy(pk.data(), (const uint32_t*)ln->d_payload.p + off, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess) return VO_ERR_HIP;
                int32_t* o = (int32_t*)out;
                for (int i = 0; i < n; i++) { o[3 * i] = UNPACK_X(pk[i]); o[3 * i + 1] = UNPACK_Y(pk[i]); o[3 * i + 2] = UNPACK_S(pk[i]); }
            }
            return have;
        }
        case VO_DBG_KEYPOINTS:
        case VO_DBG_DESCRIPTORS: {
            if (index < 0 || index >= ln->last_frames) return VO_ERR_INVALID;
            const int n = read_int(ln->d_nkp.p, index);
            if (what == VO_DBG_KEYPOINTS) return copy((const vo_keypoint*)ln->d_kp.p + (long)index * kc, (long)n * sizeof(vo_keypoint));
            return copy((const uint8_t*)ln->d_desc.p + (long)index * kc * 32, (long)n * 32);
        }
        case VO_DBG_MATCHES: {
            if (index < 0 || index >= c->last_pairs) return VO_ERR_INVALID;
            const int n = read_int(c->nmatch(), index);
            return copy((const int4*)c->mq() + (long)index * kc, (long)n * 16);
        }
        case VO_DBG_E_MASK: {
            if (index < 0 || index >= c->last_pairs) return VO_ERR_INVALID;
            const int n = read_int(c->nmatch(), index);
            const long have = 72 + n;
            if (out && bytes >= have) {
                PoseWork w;
                if (hipMemcpy(&w, (const PoseWork*)c->d_pose.p + index, sizeof(w), hipMemcpyDeviceToHost) != hipSuccess) return VO_ERR_HIP;
                memcpy(out, w.E, 72);
                if (n && hipMemcpy((uint8_t*)out + 72, (const uint8_t*)c->d_mask.p + (long)index * kc, n, hipMemcpyDeviceToHost) != hipSuccess) return VO_ERR_HIP;
            }
            return have;
        }
        case VO_DBG_VOTES: {
            if (index < 0 || index >= c->last_pairs) return VO_ERR_INVALID;
            if (out && bytes >= 16 && hipMemcpy(out, (const int*)c->d_vote.p + 4L * index, 16, hipMemcpyDeviceToHost) != hipSuccess) return VO_ERR_HIP;
            return 16;
        }
    }
    return VO_ERR_INVALID;
}

}  // extern "C"
