/*
 * vo_mi355.h -- C ABI of libvo_mi355.so, the MI355X (gfx950) visual-odometry front-end.
 *
 * This is the drop-in boundary for the hot path of gilbertgonz/visual_slam
 * (SURVEY.md section 8(b)).  The reference has no native FFI of its own: every entry point
 * below replaces one cv2 call made from /root/reference/libs/vo_utils.py, and is what a
 * ctypes binding inside that file would bind.
 *
 * Conventions: plain C, no exceptions cross the boundary.  Every function returns
 * VO_OK (0) or a negative VO_ERR_* code; vo_last_error(ctx) gives the message of the
 * last failure on that context.  The caller allocates every output with the capacity
 * it passes in.  "host" pointers are ordinary process memory; "dev" pointers are HIP
 * device memory on the context's GPU (e.g. torch.Tensor.data_ptr()).  One context per
 * host thread / stream; calls on one context are serialised by the caller.
 * There is NO CPU fallback: without a GPU vo_ctx_create fails.
 */
#ifndef VO_MI355_H
#define VO_MI355_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define VO_OK 0
#define VO_ERR_INVALID (-1)     /* bad argument */
#define VO_ERR_HIP (-2)         /* a HIP runtime call failed */
#define VO_ERR_CAPACITY (-3)    /* an output or an internal list would overflow */
#define VO_ERR_TOO_FEW (-4)     /* fewer than 5 matches / fewer than 2 train descriptors */
#define VO_ERR_NO_MODEL (-5)    /* RANSAC found no essential matrix */

typedef struct vo_ctx vo_ctx;

/* cv2.ORB_create(3000) parameters, libs/vo_utils.py:23 (defaults = OpenCV's). */
typedef struct {
    int32_t nfeatures;      /* 3000 */
    float   scale_factor;   /* 1.2f */
    int32_t nlevels;        /* 8 (1..8 supported) */
    int32_t edge_threshold; /* 31 (fixed) */
    int32_t patch_size;     /* 31 (fixed) */
    int32_t fast_threshold; /* 20 */
    /* Variants of recalled OpenCV 4.8.0 behaviour that no fixture of the reference pins (SURVEY.md
     * App. A "[L]" items): each is switchable on its own, defaults first. */
    int32_t blur_mode;      /* A.1.7: 0 float sepFilter2D kernel (default), 1 int32 kernel x256 */
    int32_t fma_atan;       /* A.1.6: 1 fastAtan2's polynomial contracts to fma (AVX2+FMA3 dispatch, default), 0 not */
    int32_t fma_blur;       /* A.1.7: 1 the float blur's a*b+c contract to fma (default), 0 not */
    int32_t trig_mode;      /* A.1.8: 0 (float)cos((double)angle) (default), 1 glibc cosf/sinf (the float overloads) */
    /* Keypoints the workspace holds per frame; 0 = nfeatures + 1024.  retainBest keeps every keypoint that ties with the
     * n-th best, so cv2 can return more than nfeatures; a frame beyond the capacity is reported VO_ERR_CAPACITY (never
     * truncated), and a caller that wants cv2's behaviour for such frames creates a context with a larger capacity and
     * retries, as the drop-in class does (visual_slam_amd/vo_utils.py).  At most 65535. */
    int32_t keypoint_capacity;
} vo_orb_params;

/* cv::KeyPoint without class_id (always -1). */
typedef struct {
    float x, y, size, angle, response;
    int32_t octave;
} vo_keypoint;

/* One frame pair through get_matches + get_pose (libs/vo_utils.py:83-110,136-145). */
typedef struct {
    double  T[16];     /* form_transf(R, t), row-major 4x4 */
    int32_t n_kp1;     /* keypoints in the previous frame (query) */
    int32_t n_kp2;     /* keypoints in the current frame (train) */
    int32_t n_match;   /* matches that pass the ratio test */
    int32_t n_inlier;  /* RANSAC inliers of the returned E */
    int32_t n_good;    /* recoverPose cheirality count */
    int32_t status;    /* VO_OK or VO_ERR_* for this pair */
} vo_pair_result;

void vo_default_orb_params(vo_orb_params* p);

/* Replaces VisualOdometry.__init__'s cv2 object construction (libs/vo_utils.py:23-33):
 * creates the device context and its HBM workspace for images up to max_w x max_h.
 * chunk_pairs: frame pairs whose detect+match stages run in one launch wave (the chunk's
 * pyramids are sized to stay resident in the 256 MiB Infinity Cache; 2*chunk_pairs frames).
 * max_pairs: frame pairs whose pose stage (RANSAC + recoverPose) runs in one launch wave. */
int vo_ctx_create(int device, int max_w, int max_h, int chunk_pairs, int max_pairs,
                  const vo_orb_params* params, vo_ctx** out);
void vo_ctx_destroy(vo_ctx* ctx);
const char* vo_last_error(const vo_ctx* ctx);

/* Replaces orb.detectAndCompute(img, None), libs/vo_utils.py:94-95.
 * img: host, 8-bit gray, h rows of `stride` bytes.  kps/desc: host, capacity `cap`
 * keypoints (desc is [cap][32]).  *n receives the number of keypoints. */
int vo_orb_detect_compute(vo_ctx* ctx, const uint8_t* img, int w, int h, int stride,
                          vo_keypoint* kps, uint8_t* desc, int cap, int* n);

/* Replaces knnMatch(f1, f2, k=2) + the ratio loop, libs/vo_utils.py:98-104, with the
 * north-star's exact BFMatcher(NORM_HAMMING).  q,t: host [n][32].  Outputs (host, capacity
 * cap): for each kept match, ascending queryIdx: queryIdx, trainIdx, best and second-best
 * distance.  ratio is the Python double (0.45): keep iff (double)d1 < ratio*(double)d2. */
int vo_match_knn2_ratio(vo_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, double ratio,
                        int32_t* qidx, int32_t* tidx, int32_t* d1, int32_t* d2, int cap, int* n);

/* Same matcher without the ratio filter: all nq rows (idx1,d1,idx2,d2), host arrays [nq]. */
int vo_match_knn2(vo_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt,
                  int32_t* idx1, int32_t* d1, int32_t* idx2, int32_t* d2);

/* The matcher for callers that cannot allocate (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  It has NO counterpart in the reference: it is vo_match_knn2_ratio
 * with device pointers in and out, and with the kernels' workspace (staged descriptors, per-segment
 * best/second keys, compacted rows, counters: 459,008 bytes) in a file-scope device array instead of
 * the context's buffers; its address on the context's GPU is resolved once, in vo_ctx_create.
 * q_dev [nq][32], t_dev [nt][32]: device memory.  matches_dev: device int32 [cap][4] = queryIdx,
 * trainIdx, d1, d2 of the kept matches in ascending queryIdx; *n_dev (device int32) receives their
 * number, rows at or beyond cap are not written.  nq, nt <= VO_NOALLOC_MAX_DESC, else VO_ERR_CAPACITY.
 * The work is enqueued on `stream` (hipStream_t; NULL = the default stream) and the call returns
 * without synchronising; it calls no hipMalloc / hipFree / allocating routine.  All calls on one GPU
 * share that GPU's array: calls that overlap on one GPU (two streams, two threads) race. */
#define VO_NOALLOC_MAX_DESC 4096
int vo_match_knn2_ratio_noalloc(vo_ctx* ctx, const uint8_t* q_dev, int nq, const uint8_t* t_dev, int nt, double ratio,
                                int32_t* matches_dev, int cap, int32_t* n_dev, void* stream);

/* Replaces cv2.findEssentialMat(q1, q2, K, cv2.RANSAC, prob, thr, maxIters),
 * libs/vo_utils.py:141.  q1,q2: host float32 [n][2]; K row-major; E row-major; mask [n]. */
int vo_find_essential_ransac(vo_ctx* ctx, const float* q1, const float* q2, int n, const double K[9],
                             double prob, double thr, int max_iters,
                             double E[9], uint8_t* mask, int* n_inliers);

/* Replaces cv2.recoverPose(E, q1, q2, K, <R out>), libs/vo_utils.py:143: no input mask,
 * distanceThresh 50 in the reference's call. */
int vo_recover_pose(vo_ctx* ctx, const double E[9], const float* q1, const float* q2, int n, const double K[9],
                    double dist_thresh, double R[9], double t[3], int* n_good);

/* Replaces cv2.triangulatePoints(P1, P2, pts1, pts2), libs/vo_utils.py:160, batched over
 * n points: x1,x2 host float32 [n][2]; X4 host float32 [4][n] (homogeneous, as cv2). */
int vo_triangulate(vo_ctx* ctx, const double P1[12], const double P2[12], const float* x1, const float* x2,
                   int n, float* X4);

/* Replaces the whole of get_matches (libs/vo_utils.py:83-110: imread'ed frames -> detectAndCompute x2 -> knnMatch ->
 * ratio loop -> q1, q2) in one call: imgA = previous frame, imgB = current frame, host, 8-bit gray.  q1,q2: host
 * float32 [cap][2], the matched pixel positions in ascending queryIdx; *n their number.  Fewer than 2 keypoints in
 * the current frame -> VO_ERR_TOO_FEW (the reference's `for m,n in matches` raises there). */
int vo_get_matches(vo_ctx* ctx, const uint8_t* imgA, const uint8_t* imgB, int w, int h, int stride, double ratio,
                   float* q1, float* q2, int cap, int* n);

/* Replaces the whole of get_pose (libs/vo_utils.py:136-145: findEssentialMat(q1, q2, K, prob 0.999, threshold 1.0) ->
 * recoverPose(E, q1, q2, K) -> form_transf(R, t)) in one call: q1, q2 host float32 [n][2] as get_matches returns them,
 * T = row-major 4x4 [R t; 0 1].  n_inliers (RANSAC) and n_good (cheirality) may be NULL.  Same results as
 * vo_find_essential_ransac + vo_recover_pose with those arguments; VO_ERR_TOO_FEW / VO_ERR_NO_MODEL as there. */
int vo_get_pose(vo_ctx* ctx, const float* q1, const float* q2, int n, const double K[9], double T[16], int* n_inliers, int* n_good);

/* The whole per-pair path for a batch of B independent frame pairs whose images are
 * already resident in device memory: imgsA/imgsB are dev pointers to [B][h][stride]
 * uint8 (previous / current frames).  out: HOST array [B].  Detect+match run in chunks of
 * chunk_pairs, the pose stage over up to max_pairs pairs at once.  stream: hipStream_t or NULL. */
int vo_pairs_run_device(vo_ctx* ctx, const uint8_t* imgsA_dev, const uint8_t* imgsB_dev, int B,
                        int w, int h, int stride, const double K[9], double ratio,
                        vo_pair_result* out_host, void* stream);

/* The same call in two halves, for callers that make it repeatedly on batches of at most max_pairs pairs (a rank's share
 * of every step of a sharded job): vo_pairs_submit_device returns as soon as the work is enqueued, vo_pairs_wait blocks
 * until the OLDEST submission in flight is complete and fills the out_host it was submitted with.  Up to two submissions
 * may be in flight: the pose stage of one (latency bound, about 3 ms whatever the batch) then overlaps detect + match of
 * the next instead of ending every call with an idle GPU.  The images must stay valid until the matching wait.  Results
 * are those of vo_pairs_run_device.  While a submission is in flight every other entry point of the context except
 * submit / wait returns VO_ERR_INVALID (they share its buffers); after the last wait, vo_pairs_triangulate and
 * vo_debug_read refer to the submission waited for last. */
int vo_pairs_submit_device(vo_ctx* ctx, const uint8_t* imgsA_dev, const uint8_t* imgsB_dev, int B,
                           int w, int h, int stride, const double K[9], double ratio,
                           vo_pair_result* out_host, void* stream);
int vo_pairs_wait(vo_ctx* ctx);
/* vo_sequence_run_device in the same two halves (F - 1 <= max_pairs pairs per submission); vo_pairs_wait completes it. */
int vo_sequence_submit_device(vo_ctx* ctx, const uint8_t* imgs_dev, int F, int w, int h, int stride,
                              const double K[9], double ratio, vo_pair_result* out_host, void* stream);

/* Same with host images: each chunk's H2D copy is enqueued on the stream that runs the chunk, so copies overlap the other
 * chunks' kernels.  Page-locked arrays (hipHostMalloc / hipHostRegister / torch pin_memory) copy asynchronously at PCIe
 * rate; pageable ones are staged by the calling thread. */
int vo_pairs_run(vo_ctx* ctx, const uint8_t* imgsA, const uint8_t* imgsB, int B,
                 int w, int h, int stride, const double K[9], double ratio, vo_pair_result* out);

/* A sequence: F consecutive frames [F][h][stride] (device memory), pair i = (frame i, frame i+1)
 * for i in [0, F-1), i.e. the (prev, cur) pairs the per-frame loop of vo.py:48-190 hands to
 * get_matches / get_pose.  Each frame is extracted once per chunk instead of twice
 * (libs/vo_utils.py:94-95 re-extracts the previous frame on every call); the results are
 * identical to vo_pairs_run_device on the same pairs.  out: HOST array [F-1]. */
int vo_sequence_run_device(vo_ctx* ctx, const uint8_t* imgs_dev, int F, int w, int h, int stride,
                           const double K[9], double ratio, vo_pair_result* out_host, void* stream);

/* Row T for a whole wave of pairs: replaces the per-match loop of vo.py:91-93 (2 x N calls of
 * VisualOdometry.triangulate, libs/vo_utils.py:148-164, per frame).  After a vo_pairs_run* / vo_sequence_run* call of
 * at most max_pairs pairs, the ratio-test matches (q1, q2) of those pairs are still on the device; this triangulates
 * ALL matches of the first P pairs in one launch.  P1, P2: host float64 [P][12] row-major projection matrices of
 * every pair, K @ prev_pose[:3] and K @ cur_pose[:3] as libs/vo_utils.py:153-157 forms them (the caller chains the
 * poses, vo.py:80; visual_slam_amd/trajectory.py does).  Outputs (host): X float32 [cap][3], the points already
 * divided by their 4th coordinate in float32 like libs/vo_utils.py:161; q1, q2 float32 [cap][2] (either may be
 * NULL); offsets int64 [P+1]: pair p owns rows offsets[p] .. offsets[p+1]-1, in ascending queryIdx.  A pair whose
 * status in that call's vo_pair_result is not VO_OK (no pose to chain) owns no rows. */
int vo_pairs_triangulate(vo_ctx* ctx, int P, const double* P1, const double* P2, float* X, float* q1, float* q2,
                         long cap, int64_t* offsets);

/* Same with host images (adds the H2D copy). */
int vo_sequence_run(vo_ctx* ctx, const uint8_t* imgs, int F, int w, int h, int stride,
                    const double K[9], double ratio, vo_pair_result* out);

/* ---- Bundle-adjustment residual / Jacobian pass (BASELINE.json configs[4]).
 * Replaces the finite-difference evaluation of VisualOdometry.bundle_adjustment_residuals
 * (libs/vo_utils.py:218-245 -> calc_reprojection_error :184-216 -> cv2.projectPoints, D = 0) that
 * scipy.optimize.least_squares performs at vo.py:118, generalised to P poses.
 * A plan holds the observation structure (static over solver iterations): observation o sees
 * point obs_point[o] from pose obs_pose[o].  poses [P][6] = (rvec, tvec) world->camera, points
 * [N][3], uv [M][2], all float64.  Outputs (any may be NULL): r [M][2] = projected - observed,
 * norm [M] = the reference's scalar residual, Hpp [P][6][6] = sum Jp^T Jp, bp [P][6] = sum Jp^T r,
 * Hll [N][3][3], bl [N][3], Hpl [M][6][3] = Jp^T Jl.  Reductions are atomic-free and
 * deterministic.  With observations sharded over GPUs, Hpp/bp are the rank's partial sums: one
 * all-reduce of the [P][42] block completes them (visual_slam_amd/ba.py). */
typedef struct vo_ba_plan vo_ba_plan;
int vo_ba_plan_create(int device, int n_poses, int n_points, int n_obs, const int32_t* obs_pose,
                      const int32_t* obs_point, vo_ba_plan** out);
void vo_ba_plan_destroy(vo_ba_plan* plan);
/* device pointers; stream: hipStream_t or NULL (NULL = the plan's stream, synchronised on return) */
int vo_ba_pass_device(vo_ba_plan* plan, const double* poses_dev, const double* points_dev, const double* uv_dev,
                      const double K[9], double* r_dev, double* norm_dev, double* Hpp_dev, double* bp_dev,
                      double* Hll_dev, double* bl_dev, double* Hpl_dev, void* stream);
/* Diagnostic, as vo_set_kernel_timing for the detect kernels: with timing on, vo_ba_pass_device brackets each of its two
 * launches with HIP events on the launch stream and waits for them; vo_ba_get_kernel_timing returns the accumulated
 * milliseconds and launch counts of k_ba_rot and k_ba_pass (ms[2], launches[2]) since timing was switched on. */
int vo_ba_set_kernel_timing(vo_ba_plan* plan, int on);
int vo_ba_get_kernel_timing(vo_ba_plan* plan, double ms[2], int launches[2]);
/* host pointers (adds the copies) */
int vo_ba_pass(vo_ba_plan* plan, const double* poses, const double* points, const double* uv, const double K[9],
               double* r, double* norm, double* Hpp, double* bp, double* Hll, double* bl, double* Hpl);

/* Stage read-back for parity tests: copies an internal device buffer of the last
 * vo_orb_detect_compute / vo_pairs_run* chunk to host.  `what`: see VO_DBG_*.
 * Returns the number of bytes the buffer holds (copying at most `bytes`). */
#define VO_DBG_PYR_LEVEL 1      /* frame, level -> tight level image (w_l*h_l bytes) */
#define VO_DBG_BLUR_LEVEL 2     /* frame, level -> tight blurred level (complete in contexts created with max_pairs = 1; batch
                                   contexts blur only the 64 x 64 tiles that a descriptor patch reaches) */
#define VO_DBG_FAST_LIST 3      /* frame, level -> int32 [n][3] x,y,score (raster order) */
#define VO_DBG_SEL1_LIST 4      /* frame, level -> int32 [n][3] after retainBest(2*quota) */
#define VO_DBG_SEL1_HARRIS 5    /* frame, level -> float [n] Harris of the SEL1 list */
#define VO_DBG_KEYPOINTS 6      /* frame -> vo_keypoint [n] */
#define VO_DBG_DESCRIPTORS 7    /* frame -> uint8 [n][32] */
#define VO_DBG_MATCHES 8        /* pair -> int32 [n][4] queryIdx, trainIdx, d1, d2 after ratio */
#define VO_DBG_E_MASK 9         /* pair -> double E[9] followed by uint8 mask[n_match] */
#define VO_DBG_VOTES 10         /* pair -> int32 [4]: recoverPose's cheirality counts of [R1|t], [R2|t], [R1|-t], [R2|-t] */
long vo_debug_read(vo_ctx* ctx, int what, int index, int level, void* out, long bytes);
/* Parity hook for the five-point minimal solver of vo_find_essential_ransac (no counterpart in the reference's call
 * surface: cv::findEssentialMat runs it internally, modules/calib3d/src/five-point.cpp EMEstimatorCallback::runKernel
 * and cv::solvePoly in OpenCV 4.x).  Host pointers, n independent cases, one GPU lane each.
 *   mode 0: in = n x 11 polynomial coefficients (ascending powers); out = n x 10 roots as (re, im) pairs
 *   mode 1: in = n x (x1[5][2], x2[5][2]) normalised correspondences; out = n x 91: the model count, then up to ten
 *           row-major 3x3 essential matrices (unused slots 0)
 *   modes 2, 3: the same two through the 16-lanes-per-case form of the solver that small batches use */
int vo_debug_minimal_solver(vo_ctx* ctx, int mode, const double* in, int n, double* out);
/* retainBest permutes its list in place, so the FAST and SEL1 lists only exist if the detect
 * chain is stopped early: 0 = run everything (default), 1 = stop after the FAST lists,
 * 2 = stop after retainBest #1 + Harris. */
int vo_set_debug_stop(vo_ctx* ctx, int stage);

/* Per-kernel timing of the vo_pairs_run_device calls made while it is enabled, measured with
 * HIP events recorded around every launch on the stream the kernels are launched on
 * (diagnostic mode: it serialises the launches).  ms[k] = summed duration, launches[k] =
 * number of launches of kernel k since timing was enabled.  VO_K_RANSAC / VO_K_RECOVER
 * bracket the whole multi-kernel stage. */
#define VO_N_KERNELS 14
#define VO_K_PYRAMID 0
#define VO_K_FAST_TILES 1
#define VO_K_FAST_SCAN 2
#define VO_K_FAST_EMIT 3
#define VO_K_SELECT1 4
#define VO_K_HARRIS 5
#define VO_K_SELECT2 6
#define VO_K_ORIENT 7
#define VO_K_BLUR 8
#define VO_K_RBRIEF 9
#define VO_K_MATCH 10
#define VO_K_RATIO 11
#define VO_K_RANSAC 12
#define VO_K_RECOVER 13
int vo_set_kernel_timing(vo_ctx* ctx, int enable);
int vo_get_kernel_timing(vo_ctx* ctx, float ms[VO_N_KERNELS], int32_t launches[VO_N_KERNELS]);

#ifdef __cplusplus
}
#endif
#endif
