/*
 * vo_oracle.h -- C ABI of the CPU parity oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under visual_slam_amd/ (the product) may include,
 * link, import or call this.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / the CPU number reported beside
 * the GPU number.
 *
 * PARITY UNPINNED: the arithmetic of the reference's hot path lives in a third-party
 * dependency that is absent from /root/reference and from this image
 * (opencv-python==4.8.0.76, /root/reference/requirements.txt:2), and the reference
 * holds no tests, golden vectors or fixtures.  This file
 * restates the published OpenCV 4.8.0 algorithms behind the calls at
 *   /root/reference/libs/vo_utils.py:23   cv2.ORB_create(3000)
 *   /root/reference/libs/vo_utils.py:94   orb.detectAndCompute(img, None)
 *   /root/reference/libs/vo_utils.py:98   knnMatch(f1, f2, k=2)  (north-star: BFMatcher NORM_HAMMING)
 *   /root/reference/libs/vo_utils.py:103  ratio test
 *   /root/reference/libs/vo_utils.py:141  cv2.findEssentialMat(q1,q2,K,RANSAC,0.999,1.0,None)
 *   /root/reference/libs/vo_utils.py:143  cv2.recoverPose(E,q1,q2,K,mask)
 *   /root/reference/libs/vo_utils.py:160  cv2.triangulatePoints(P1,P2,pts1,pts2)
 * and is pinned only by known-answer tests built from the mathematical definitions
 * (tests/test_oracle_*.py).
 */
#ifndef VO_ORACLE_H
#define VO_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    float x, y;      /* pt, full-resolution pixel units */
    float size;      /* patchSize * scale(octave) */
    float angle;     /* degrees [0,360) */
    float response;  /* Harris response */
    int32_t octave;  /* pyramid level */
} orc_keypoint;      /* class_id is always -1 */

typedef struct {
    int32_t nfeatures;       /* 3000 */
    float   scale_factor;    /* 1.2f */
    int32_t nlevels;         /* 8 */
    int32_t edge_threshold;  /* 31 */
    int32_t patch_size;      /* 31 */
    int32_t fast_threshold;  /* 20 */
    int32_t blur_mode;       /* 0: sepFilter2D float kernel (OpenCV >= 4.5 path, default)
                                1: int32 kernel [18,34,49,55,49,34,18]>>16 */
    int32_t fma_atan;        /* 1: fastAtan2 contracts a*b+c to fma, as the AVX2+FMA3-dispatched
                                   mathfuncs_core of an x86 build does (default); 0: separate mul, add */
    int32_t fma_blur;        /* the same choice for the float sepFilter2D blur (another translation unit) */
    int32_t trig_mode;       /* descriptor rotation: 0 (float)cos((double)angle) (default);
                                1 the float overloads = this host's libm cosf / sinf */
} orc_orb_params;

void orc_default_orb_params(orc_orb_params* p);

/* level geometry: lw/lh/scale/quota each [nlevels] */
void orc_level_geometry(int w, int h, const orc_orb_params* p, int* lw, int* lh, float* scale, int* quota);

/* Optional stage dumps; any pointer may be NULL.  Pyramid buffers are tight
 * (row pitch == level width), levels concatenated 0..nlevels-1. */
typedef struct {
    uint8_t* pyr;         /* unblurred levels */
    uint8_t* blurred;     /* blurred levels */
    int32_t* fast_count;  /* [nlevels] FAST+NMS+border-filtered keypoints per level */
    int32_t* fast_xys;    /* [fast_cap][3] x,y,score, level-major, raster order inside a level */
    int32_t  fast_cap;
    int32_t* sel1_count;  /* [nlevels] after retainBest(2*quota) */
    int32_t* sel1_xys;    /* [sel1_cap][3] x,y,score in the order retainBest left them */
    float*   sel1_harris; /* [sel1_cap] Harris response of those */
    int32_t  sel1_cap;
    int32_t* sel2_count;  /* [nlevels] after retainBest(quota) on Harris */
} orc_orb_debug;

/* ORB::detectAndCompute(image, noArray()).  Returns 0 or <0 on error.  *n = keypoints. */
int orc_orb_detect_compute(const uint8_t* img, int w, int h, int stride,
                           const orc_orb_params* p, orc_orb_debug* dbg,
                           orc_keypoint* kps, uint8_t* desc, int cap, int* n);

/* BFMatcher(NORM_HAMMING).knnMatch(q, t, k=2): per query the two best (distance, trainIdx)
 * in lexicographic order.  Requires nt >= 2. */
int orc_match_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt,
                   int32_t* idx1, int32_t* d1, int32_t* idx2, int32_t* d2);

/* Lowe ratio test of libs/vo_utils.py:101-104 in Python-double arithmetic:
 * keep query i iff (double)d1 < ratio*(double)d2.  Writes kept query indices ascending. */
int orc_ratio_filter(const int32_t* d1, const int32_t* d2, int nq, double ratio, int32_t* keep, int* nkeep);

/* cv::findEssentialMat(q1,q2,K,RANSAC,prob,thr,maxIters).  E row-major.  mask [n].
 * Returns 0 ok, -1 if no model (fewer than 5 points or no consensus). iters_run optional. */
int orc_find_essential_ransac(const float* q1, const float* q2, int n, const double K[9],
                              double prob, double thr, int max_iters,
                              double E[9], uint8_t* mask, int* n_inliers, int* iters_run);

/* five-point minimal solver on already-normalised points (x,y pairs).  Ebuf [10*9]. */
int orc_five_point(const double* x1, const double* x2, double* Ebuf);
/* cv::solvePoly on a degree-10 coefficient vector (ascending powers); roots as (re, im) pairs */
int orc_solve_poly10(const double* coeffs, double* roots_re_im);

/* Rounding variants of the pose side (process-wide; defaults 0, 0 = what the HIP solver implements): hypot_mode 1 = std::hypot in
 * the Jacobi rotations, ainv_mode 1 = LU inverse + matrix product for the 10 x 10 elimination (see vo_oracle.cpp). */
void orc_set_pose_variant(int hypot_mode, int ainv_mode);

/* cv::recoverPose(E, q1, q2, K, R, t) -- no input mask, distanceThresh given. */
int orc_recover_pose(const double E[9], const float* q1, const float* q2, int n, const double K[9],
                     double dist_thresh, double R[9], double t[3], int* n_good);
/* the same, also returning the cheirality counts of the four candidates [R1|t], [R2|t], [R1|-t], [R2|-t] (test hook) */
int orc_recover_pose_votes(const double E[9], const float* q1, const float* q2, int n, const double K[9],
                           double dist_thresh, double R[9], double t[3], int* n_good, int votes[4]);

/* cv::triangulatePoints for float32 points (2 floats per point, n points): X4 is [4][n]
 * float32, as cv2 returns for float32 inputs. */
int orc_triangulate_f32(const double P1[12], const double P2[12], const float* x1, const float* x2,
                        int n, float* X4);

/* whole pair: detect x2, match, ratio, pose.  T = [[R,t],[0,1]] row-major.  counts[4] =
 * n_kp1, n_kp2, n_match, n_inlier.  Returns 0 or <0 (-2: fewer than 5 matches / no model). */
int orc_pair_run(const uint8_t* imgA, const uint8_t* imgB, int w, int h, int stride,
                 const orc_orb_params* p, const double K[9], double ratio,
                 double T[16], int32_t counts[4]);

/* libstdc++ KeyPointsFilter::retainBest replay on bare keys; writes the surviving
 * original indices in final order. (exposed to test the GPU's parallel replay) */
int orc_retain_best(const float* keys, int n, int n_points, int32_t* out_idx, int* n_out);

#ifdef __cplusplus
}
#endif
#endif
