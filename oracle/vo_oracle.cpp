/*
 * vo_oracle.cpp -- CPU parity oracle (TEST INFRASTRUCTURE ONLY, see vo_oracle.h).
 *
 * PARITY UNPINNED (no cv2 in this image, no fixtures in the reference): a plain scalar
 * restatement of the OpenCV 4.8.0 algorithms the reference's hot path executes.  Each
 * function cites the reference call site it stands behind (file:line under
 * /root/reference) and the SURVEY.md appendix paragraph it restates.
 *
 * Build: g++ -O2 -ffp-contract=off -fno-fast-math (oracle/Makefile).  Contraction is off so
 * that every a*b+c below is two IEEE roundings unless fma()/fmaf() is spelled out.
 */
#include "vo_oracle.h"
#include "../include/vo_orb_pattern_31.h"

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

/* Pose-side rounding variants (pin-readiness: tests/test_zz_live_cv2.py enumerates them against a live cv2; the defaults are
 * what the HIP solver implements).  Process-wide switches, set through orc_set_pose_variant before calling; not meant to
 * be flipped while other threads run the solver.
 *   hypot_mode 0: the Jacobi rotation's gamma = sqrt(p*p + beta*beta) (default); 1: std::hypot(p, beta), as OpenCV's
 *                 JacobiSVDImpl_ writes it
 *   ainv_mode  0: A[:, :10]^-1 A[:, 10:] by Gauss-Jordan on the augmented 10 x 20 matrix (default);
 *              1: what `A.colRange(0, 10).inv() * A.colRange(10, 20)` does in OpenCV: Mat::inv(DECOMP_LU) = LU with partial
 *                 pivoting applied to an identity right-hand side and back-substituted (hal::LU64f), then a matrix product
 *                 whose every entry is a row x column sum in ascending k */
#include <atomic>
static std::atomic<int> g_pose_hypot_mode{0}, g_pose_ainv_mode{0};

namespace {

/* ---------------------------------------------------------------- rounding helpers */
/* cvRound = round-half-to-even (SSE cvtss2si / cvtsd2si under the default MXCSR). */
inline int cvRoundF(float v) { return (int)nearbyintf(v); }
inline int cvRoundD(double v) { return (int)nearbyint(v); }
inline int cvFloorD(double v) { int i = (int)v; return i - (i > v); }

/* ---------------------------------------------------------------- geometry (A.1.1) */
struct Geometry {
    int nlevels;
    std::vector<int> w, h, quota;
    std::vector<float> scale;
    std::vector<size_t> ofs;  /* offset of each tight level in a concatenated buffer */
    size_t total;
};

Geometry make_geometry(int cols, int rows, const orc_orb_params& p) {
    Geometry g;
    g.nlevels = p.nlevels;
    const double scaleFactor = (double)p.scale_factor; /* ORB_Impl stores the float as double */
    g.total = 0;
    for (int l = 0; l < p.nlevels; l++) {
        float scale = (float)std::pow(scaleFactor, (double)l); /* firstLevel = 0 */
        float inv = 1.f / scale;
        g.scale.push_back(scale);
        g.w.push_back(cvRoundF(cols * inv));
        g.h.push_back(cvRoundF(rows * inv));
        g.ofs.push_back(g.total);
        g.total += (size_t)g.w.back() * g.h.back();
    }
    /* per-level quota */
    g.quota.resize(p.nlevels);
    float factor = (float)(1.0 / scaleFactor);
    float nd = p.nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)p.nlevels));
    int sum = 0;
    for (int l = 0; l < p.nlevels - 1; l++) {
        g.quota[l] = cvRoundF(nd);
        sum += g.quota[l];
        nd *= factor;
    }
    g.quota[p.nlevels - 1] = std::max(p.nfeatures - sum, 0);
    return g;
}

/* ---------------------------------------------------------------- pyramid (A.1.2)
 * resize(prev, cur, size, 0, 0, INTER_LINEAR_EXACT) for 8-bit single channel:
 * ufixedpoint16 (8 fractional bits) coefficients, horizontal pass to 8.8, vertical pass
 * (c0*H0 + c1*H1 + 32768) >> 16. */
struct LinCoef { int ofs; int c0, c1; };

void linear_coeffs(int src, int dst, std::vector<LinCoef>& out, int& dmin, int& dmax) {
    double inv_scale = (double)dst / src;
    double scale = 1.0 / inv_scale;
    out.assign(dst, LinCoef{0, 256, 0});
    dmin = 0;
    dmax = dst;
    for (int d = 0; d < dst; d++) {
        double f = scale * ((double)d + 0.5) - 0.5;
        int i = cvFloorD(f);
        if (i >= 0 && src > 1) {
            if (i < src - 1) {
                int c1 = cvRoundD((f - (double)i) * 256.0);
                out[d] = LinCoef{i, 256 - c1, c1};
            } else {
                out[d] = LinCoef{src - 1, 256, 0};
                dmax = std::min(dmax, d);
            }
        } else {
            dmin = std::max(dmin, d + 1);
        }
    }
}

void resize_linear_exact(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride) {
    std::vector<LinCoef> cx, cy;
    int xmin, xmax, ymin, ymax;
    linear_coeffs(sw, dw, cx, xmin, xmax);
    linear_coeffs(sh, dh, cy, ymin, ymax);
    std::vector<uint16_t> H((size_t)sh * dw);
    for (int y = 0; y < sh; y++) {
        const uint8_t* s = src + (size_t)y * sstride;
        uint16_t* hrow = &H[(size_t)y * dw];
        int x = 0;
        for (; x < xmin; x++) hrow[x] = (uint16_t)(s[0] << 8);
        for (; x < xmax; x++) hrow[x] = (uint16_t)(cx[x].c0 * s[cx[x].ofs] + cx[x].c1 * s[cx[x].ofs + 1]);
        for (; x < dw; x++) hrow[x] = (uint16_t)(s[sw - 1] << 8);
    }
    for (int y = 0; y < dh; y++) {
        uint8_t* d = dst + (size_t)y * dstride;
        if (y < ymin) {
            for (int x = 0; x < dw; x++) d[x] = (uint8_t)((H[x] + 128) >> 8);
        } else if (y >= ymax) {
            const uint16_t* h0 = &H[(size_t)(sh - 1) * dw];
            for (int x = 0; x < dw; x++) d[x] = (uint8_t)((h0[x] + 128) >> 8);
        } else {
            const uint16_t* h0 = &H[(size_t)cy[y].ofs * dw];
            const uint16_t* h1 = h0 + dw;
            uint32_t c0 = cy[y].c0, c1 = cy[y].c1;
            for (int x = 0; x < dw; x++) d[x] = (uint8_t)((c0 * h0[x] + c1 * h1[x] + 32768u) >> 16);
        }
    }
}

/* ---------------------------------------------------------------- FAST-9/16 (A.1.3) */
const int kCircle[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                            {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

/* returns 0 for a non-corner, else the cornerScore (>= threshold, fits a byte) */
int fast_score(const uint8_t* p, int stride, int threshold) {
    int v = p[0];
    int d[25];
    for (int k = 0; k < 25; k++) d[k] = v - p[kCircle[k & 15][0] + kCircle[k & 15][1] * stride];
    /* corner test: 9 contiguous strictly darker than v-t or strictly brighter than v+t */
    bool corner = false;
    int run_d = 0, run_b = 0;
    for (int k = 0; k < 25 && !corner; k++) {
        run_d = d[k] > threshold ? run_d + 1 : 0;  /* neighbour darker: p < v - t */
        run_b = d[k] < -threshold ? run_b + 1 : 0; /* neighbour brighter: p > v + t */
        if (run_d > 8 || run_b > 8) corner = true;
    }
    if (!corner) return 0;
    /* cornerScore<16>: largest threshold for which the pixel is still a corner, minus one */
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = std::min(d[k + 1], d[k + 2]);
        a = std::min(a, d[k + 3]);
        if (a <= a0) continue;
        a = std::min(a, d[k + 4]);
        a = std::min(a, d[k + 5]);
        a = std::min(a, d[k + 6]);
        a = std::min(a, d[k + 7]);
        a = std::min(a, d[k + 8]);
        a0 = std::max(a0, std::min(a, d[k]));
        a0 = std::max(a0, std::min(a, d[k + 9]));
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = std::max(d[k + 1], d[k + 2]);
        b = std::max(b, d[k + 3]);
        b = std::max(b, d[k + 4]);
        b = std::max(b, d[k + 5]);
        if (b >= b0) continue;
        b = std::max(b, d[k + 6]);
        b = std::max(b, d[k + 7]);
        b = std::max(b, d[k + 8]);
        b0 = std::min(b0, std::max(b, d[k]));
        b0 = std::min(b0, std::max(b, d[k + 9]));
    }
    return -b0 - 1;
}

struct KP {
    float x, y, response, angle;
    int octave;
};

/* FAST(img, kps, threshold, nonmaxSuppression=true) in raster order, followed by
 * KeyPointsFilter::runByImageBorder(kps, size, edge)  (A.1.3, A.1.4) */
void fast_detect(const uint8_t* img, int w, int h, int stride, int threshold, int edge, std::vector<KP>& out) {
    out.clear();
    if (w < 7 || h < 7) return;
    std::vector<uint8_t> score((size_t)w * h, 0);
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) score[(size_t)y * w + x] = (uint8_t)fast_score(img + (size_t)y * stride + x, stride, threshold);
    const bool border_ok = (w > 2 * edge) && (h > 2 * edge);
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            int s = score[(size_t)y * w + x];
            if (!s) continue;
            const uint8_t* r = &score[(size_t)y * w + x];
            if (s > r[-1] && s > r[1] && s > r[-w - 1] && s > r[-w] && s > r[-w + 1] && s > r[w - 1] && s > r[w] && s > r[w + 1]) {
                if (border_ok && x >= edge && x < w - edge && y >= edge && y < h - edge) out.push_back(KP{(float)x, (float)y, (float)s, -1.f, 0});
            }
        }
}

/* ---------------------------------------------------------------- retainBest (A.1.5)
 * KeyPointsFilter::retainBest: std::nth_element + std::partition from libstdc++ -- the
 * permutation they leave is part of the result (keypoint order). */
struct KeyIdx {
    float response;
    int idx;
};
template <class T>
void retain_best(std::vector<T>& v, int n_points) {
    if (n_points >= 0 && v.size() > (size_t)n_points) {
        if (n_points == 0) {
            v.clear();
            return;
        }
        std::nth_element(v.begin(), v.begin() + n_points - 1, v.end(), [](const T& a, const T& b) { return a.response > b.response; });
        float amb = v[n_points - 1].response;
        auto new_end = std::partition(v.begin() + n_points, v.end(), [amb](const T& a) { return a.response >= amb; });
        v.resize(new_end - v.begin());
    }
}

/* ---------------------------------------------------------------- Harris (A.1.5) */
float harris_response(const uint8_t* img, int stride, int x0, int y0) {
    const int blockSize = 7, r = blockSize / 2;
    const float harris_k = 0.04f;
    float scale = 1.f / ((1 << 2) * blockSize * 255.f);
    float scale_sq_sq = scale * scale * scale * scale;
    const uint8_t* ptr0 = img + (size_t)(y0 - r) * stride + (x0 - r);
    int a = 0, b = 0, c = 0;
    for (int i = 0; i < blockSize; i++)
        for (int j = 0; j < blockSize; j++) {
            const uint8_t* ptr = ptr0 + i * stride + j;
            int Ix = (ptr[1] - ptr[-1]) * 2 + (ptr[-stride + 1] - ptr[-stride - 1]) + (ptr[stride + 1] - ptr[stride - 1]);
            int Iy = (ptr[stride] - ptr[-stride]) * 2 + (ptr[stride - 1] - ptr[-stride - 1]) + (ptr[stride + 1] - ptr[-stride + 1]);
            a += Ix * Ix;
            b += Iy * Iy;
            c += Ix * Iy;
        }
    return ((float)a * b - (float)c * c - harris_k * ((float)a + b) * ((float)a + b)) * scale_sq_sq;
}

/* ---------------------------------------------------------------- IC angle (A.1.6) */
const int kUmax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};

float fast_atan2_deg(float y, float x, bool use_fma) {
    const float p1 = 0.9997878412794807f * (float)(180 / M_PI);
    const float p3 = -0.3258083974640975f * (float)(180 / M_PI);
    const float p5 = 0.1555786518463281f * (float)(180 / M_PI);
    const float p7 = -0.04432655554792128f * (float)(180 / M_PI);
    float ax = std::fabs(x), ay = std::fabs(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        if (use_fma)
            a = fmaf(fmaf(fmaf(p7, c2, p5), c2, p3), c2, p1) * c;
        else
            a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        if (use_fma)
            a = fmaf(-fmaf(fmaf(fmaf(p7, c2, p5), c2, p3), c2, p1), c, 90.f);
        else
            a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

float ic_angle(const uint8_t* img, int stride, int x0, int y0, bool use_fma) {
    const uint8_t* center = img + (size_t)y0 * stride + x0;
    int m_01 = 0, m_10 = 0;
    for (int u = -15; u <= 15; ++u) m_10 += u * center[u];
    for (int v = 1; v <= 15; ++v) {
        int v_sum = 0, d = kUmax[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * stride], val_minus = center[u - v * stride];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return fast_atan2_deg((float)m_01, (float)m_10, use_fma);
}

/* ---------------------------------------------------------------- blur (A.1.7) */
inline int reflect101(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i;
}

void gaussian_kernel7_sigma2(float k[7]) {
    /* getGaussianKernel(7, 2, CV_32F): exp(-(x^2)/(2 sigma^2)) normalised in double, cast to float */
    double t[3], sum = 0;
    const double scale2X = -0.125 / (2.0 * 2.0);
    for (int i = 0, x = 1 - 7; i < 3; i++, x += 2) {
        t[i] = std::exp((double)(x * x) * scale2X);
        sum += t[i];
    }
    sum *= 2;
    sum += 1.0;
    double mul1 = 1.0 / sum;
    for (int i = 0; i < 3; i++) k[i] = k[6 - i] = (float)(t[i] * mul1);
    k[3] = (float)(1.0 * mul1);
}

/* mode 0: sepFilter2D with the float kernel (row filter u8->f32 left to right, symmetric
 * column filter f32->u8 with saturate_cast = cvRound).  mode 1: int32 kernel x256. */
void gaussian_blur7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride, int mode, bool use_fma) {
    if (mode == 1) {
        static const int k[7] = {18, 34, 49, 55, 49, 34, 18};
        std::vector<int> rows((size_t)w * h);
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                int s = 0;
                for (int i = 0; i < 7; i++) s += k[i] * src[(size_t)y * sstride + reflect101(x + i - 3, w)];
                rows[(size_t)y * w + x] = s;
            }
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                int s = 0;
                for (int i = 0; i < 7; i++) s += k[i] * rows[(size_t)reflect101(y + i - 3, h) * w + x];
                s = (s + 32768) >> 16;
                dst[(size_t)y * dstride + x] = (uint8_t)std::min(std::max(s, 0), 255);
            }
        return;
    }
    float k[7];
    gaussian_kernel7_sigma2(k);
    std::vector<float> rows((size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint8_t* s = src + (size_t)y * sstride;
            float acc = k[0] * (float)s[reflect101(x - 3, w)];
            for (int i = 1; i < 7; i++) {
                float pv = (float)s[reflect101(x + i - 3, w)];
                acc = use_fma ? fmaf(k[i], pv, acc) : acc + k[i] * pv;
            }
            rows[(size_t)y * w + x] = acc;
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float acc = k[3] * rows[(size_t)y * w + x] + 0.f;
            for (int i = 1; i <= 3; i++) {
                float t = rows[(size_t)reflect101(y + i, h) * w + x] + rows[(size_t)reflect101(y - i, h) * w + x];
                acc = use_fma ? fmaf(k[3 + i], t, acc) : acc + k[3 + i] * t;
            }
            int iv = cvRoundF(acc);
            dst[(size_t)y * dstride + x] = (uint8_t)std::min(std::max(iv, 0), 255);
        }
}

/* ---------------------------------------------------------------- rBRIEF (A.1.8) */
void orb_descriptor(const uint8_t* img, int stride, int cx, int cy, float angle_deg, uint8_t desc[32], int trig_mode) {
    float angle = angle_deg;
    angle *= (float)(M_PI / 180.f);
    float a, b;
    if (trig_mode == 1) { a = cosf(angle); b = sinf(angle); }      /* the float overloads: this host's libm itself */
    else { a = (float)std::cos((double)angle); b = (float)std::sin((double)angle); }
    const uint8_t* center = img + (size_t)cy * stride + cx;
    const signed char* pat = VO_ORB_PATTERN_31;
    for (int i = 0; i < 32; i++) {
        int val = 0;
        for (int j = 0; j < 8; j++) {
            int t[2];
            for (int s = 0; s < 2; s++) {
                const signed char* pp = pat + ((i * 16 + j * 2 + s) * 2);
                float px = (float)pp[0], py = (float)pp[1];
                float x = px * a - py * b;
                float y = px * b + py * a;
                int ix = cvRoundF(x), iy = cvRoundF(y);
                t[s] = center[iy * stride + ix];
            }
            val |= (t[0] < t[1]) << j;
        }
        desc[i] = (uint8_t)val;
    }
}

/* ---------------------------------------------------------------- cv::RNG */
struct CvRNG {
    uint64_t state;
    explicit CvRNG(uint64_t s) : state(s ? s : 0xffffffffu) {}
    unsigned next() {
        state = (uint64_t)(unsigned)state * 4164903690u + (unsigned)(state >> 32);
        return (unsigned)state;
    }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

/* ---------------------------------------------------------------- JacobiSVDImpl_<double>
 * One-sided Jacobi on the rows of At (n rows of length m); Vt n x n accumulates the
 * rotations; rows n..n1-1 of At are completed to an orthonormal set with RNG(0x12345678)
 * exactly as OpenCV does for FULL_UV.  hypot() is written sqrt(p*p+beta*beta). */
void jacobi_svd(double* At, int astep, double* W, double* Vt, int vstep, int m, int n, int n1) {
    const double minval = DBL_MIN, eps = DBL_EPSILON * 10;
    int i, j, k, iter, max_iter = std::max(m, 30);
    double c, s, sd;
    for (i = 0; i < n; i++) {
        for (k = 0, sd = 0; k < m; k++) {
            double t = At[i * astep + k];
            sd += t * t;
        }
        W[i] = sd;
        if (Vt) {
            for (k = 0; k < n; k++) Vt[i * vstep + k] = 0;
            Vt[i * vstep + i] = 1;
        }
    }
    for (iter = 0; iter < max_iter; iter++) {
        bool changed = false;
        for (i = 0; i < n - 1; i++)
            for (j = i + 1; j < n; j++) {
                double *Ai = At + i * astep, *Aj = At + j * astep;
                double a = W[i], p = 0, b = W[j];
                for (k = 0; k < m; k++) p += Ai[k] * Aj[k];
                if (std::fabs(p) <= eps * std::sqrt(a * b)) continue;
                p *= 2;
                double beta = a - b, gamma = g_pose_hypot_mode.load() ? std::hypot(p, beta) : std::sqrt(p * p + beta * beta);
                if (beta < 0) {
                    double delta = (gamma - beta) * 0.5;
                    s = std::sqrt(delta / gamma);
                    c = p / (gamma * s * 2);
                } else {
                    c = std::sqrt((gamma + beta) / (gamma * 2));
                    s = p / (gamma * c * 2);
                }
                a = b = 0;
                for (k = 0; k < m; k++) {
                    double t0 = c * Ai[k] + s * Aj[k];
                    double t1 = -s * Ai[k] + c * Aj[k];
                    Ai[k] = t0;
                    Aj[k] = t1;
                    a += t0 * t0;
                    b += t1 * t1;
                }
                W[i] = a;
                W[j] = b;
                changed = true;
                if (Vt) {
                    double *Vi = Vt + i * vstep, *Vj = Vt + j * vstep;
                    for (k = 0; k < n; k++) {
                        double t0 = c * Vi[k] + s * Vj[k];
                        double t1 = -s * Vi[k] + c * Vj[k];
                        Vi[k] = t0;
                        Vj[k] = t1;
                    }
                }
            }
        if (!changed) break;
    }
    for (i = 0; i < n; i++) {
        for (k = 0, sd = 0; k < m; k++) {
            double t = At[i * astep + k];
            sd += t * t;
        }
        W[i] = std::sqrt(sd);
    }
    for (i = 0; i < n - 1; i++) {
        j = i;
        for (k = i + 1; k < n; k++)
            if (W[j] < W[k]) j = k;
        if (i != j) {
            std::swap(W[i], W[j]);
            if (Vt) {
                for (k = 0; k < m; k++) std::swap(At[i * astep + k], At[j * astep + k]);
                for (k = 0; k < n; k++) std::swap(Vt[i * vstep + k], Vt[j * vstep + k]);
            }
        }
    }
    if (!Vt) return;
    CvRNG rng(0x12345678);
    for (i = 0; i < n1; i++) {
        sd = i < n ? W[i] : 0;
        for (int ii = 0; ii < 100 && sd <= minval; ii++) {
            const double val0 = 1. / m;
            for (k = 0; k < m; k++) {
                double val = (rng.next() & 256) != 0 ? val0 : -val0;
                At[i * astep + k] = val;
            }
            for (iter = 0; iter < 2; iter++) {
                for (j = 0; j < i; j++) {
                    sd = 0;
                    for (k = 0; k < m; k++) sd += At[i * astep + k] * At[j * astep + k];
                    double asum = 0;
                    for (k = 0; k < m; k++) {
                        double t = At[i * astep + k] - sd * At[j * astep + k];
                        At[i * astep + k] = t;
                        asum += std::fabs(t);
                    }
                    asum = asum > eps * 100 ? 1 / asum : 0;
                    for (k = 0; k < m; k++) At[i * astep + k] *= asum;
                }
            }
            sd = 0;
            for (k = 0; k < m; k++) {
                double t = At[i * astep + k];
                sd += t * t;
            }
            sd = std::sqrt(sd);
        }
        s = sd > minval ? 1 / sd : 0.;
        for (k = 0; k < m; k++) At[i * astep + k] *= s;
    }
}

/* SVD::compute(A[m x m]) for square A: returns U (m x m), W, Vt (m x m), all row-major. */
template <int M>
void svd_square(const double* A, double* W, double* U, double* Vt) {
    double At[M * M], V[M * M];
    for (int i = 0; i < M; i++)
        for (int j = 0; j < M; j++) At[i * M + j] = A[j * M + i];
    jacobi_svd(At, M, W, V, M, M, M, M);
    for (int i = 0; i < M; i++)
        for (int j = 0; j < M; j++) {
            U[i * M + j] = At[j * M + i];
            Vt[i * M + j] = V[i * M + j];
        }
}

/* ---------------------------------------------------------------- 5-point (A.3.3)
 * Polynomials in (x,y,z) of total degree <= 3 held as c[i][j][k] (exponents of x,y,z). */
struct Poly3 {
    double c[4][4][4];
    Poly3() { std::memset(c, 0, sizeof(c)); }
};
Poly3 pmul(const Poly3& a, int da, const Poly3& b, int db) {
    Poly3 r;
    for (int i = 0; i <= da; i++)
        for (int j = 0; i + j <= da; j++)
            for (int k = 0; i + j + k <= da; k++) {
                double av = a.c[i][j][k];
                if (av == 0) continue;
                for (int p = 0; p <= db; p++)
                    for (int q = 0; p + q <= db; q++)
                        for (int s = 0; p + q + s <= db; s++) r.c[i + p][j + q][k + s] += av * b.c[p][q][s];
            }
    return r;
}
Poly3 padd(const Poly3& a, const Poly3& b, double sb) {
    Poly3 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            for (int k = 0; k < 4; k++) r.c[i][j][k] = a.c[i][j][k] + sb * b.c[i][j][k];
    return r;
}
/* column (monomial) order of the 10x20 constraint matrix */
const int kMono[20][3] = {{3, 0, 0}, {0, 3, 0}, {2, 1, 0}, {1, 2, 0}, {2, 0, 1}, {2, 0, 0}, {0, 2, 1}, {0, 2, 0}, {1, 1, 1}, {1, 1, 0},
                          {1, 0, 2}, {1, 0, 1}, {1, 0, 0}, {0, 1, 2}, {0, 1, 1}, {0, 1, 0}, {0, 0, 3}, {0, 0, 2}, {0, 0, 1}, {0, 0, 0}};

/* complex helpers with OpenCV's Complex<double> operator semantics */
struct Cx {
    double re, im;
};
inline Cx cmul(Cx a, Cx b) { return Cx{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
inline Cx cadd(Cx a, Cx b) { return Cx{a.re + b.re, a.im + b.im}; }
inline Cx csub(Cx a, Cx b) { return Cx{a.re - b.re, a.im - b.im}; }
inline Cx cdiv(Cx a, Cx b) {
    double t = 1. / (b.re * b.re + b.im * b.im);
    return Cx{(a.re * b.re + a.im * b.im) * t, (-a.re * b.im + a.im * b.re) * t};
}
inline double cabs_(Cx a) { return std::sqrt(a.re * a.re + a.im * a.im); }

/* cv::solvePoly(coeffs, roots, maxIters=300): Durand-Kerner with in-place (Gauss-Seidel)
 * updates from start values (1+i)^k.  coeffs[i] multiplies z^i.  The repeated-root
 * branch of OpenCV (num_same_root > 1) is taken only when two iterates coincide exactly;
 * it is restated for even multiplicities and left as the plain step otherwise. */
int solve_poly(const double* coeffs_in, int n0, Cx* roots) {
    int n = n0;
    Cx coeffs[16];
    for (int i = 0; i <= n; i++) coeffs[i] = Cx{coeffs_in[i], 0};
    for (; n > 1; n--)
        if (std::fabs(coeffs[n].re) + std::fabs(coeffs[n].im) > DBL_EPSILON) break;
    Cx p{1, 0}, r{1, 1};
    for (int i = 0; i < n; i++) {
        roots[i] = p;
        p = cmul(p, r);
    }
    const int maxIters = 300;
    for (int iter = 0; iter < maxIters; iter++) {
        double maxDiff = 0;
        for (int i = 0; i < n; i++) {
            p = roots[i];
            Cx num = coeffs[n], denom = coeffs[n];
            int num_same_root = 1;
            for (int j = 0; j < n; j++) {
                num = cadd(cmul(num, p), coeffs[n - j - 1]);
                if (j != i) {
                    Cx d = csub(p, roots[j]);
                    if (d.re != 0 || d.im != 0)
                        denom = cmul(denom, d);
                    else
                        num_same_root++;
                }
            }
            num = cdiv(num, denom);
            if (num_same_root > 1 && (num_same_root % 2) == 0) {
                double old_re = num.re, old_im = num.im;
                int times = num_same_root / 2;
                for (int j = 0; j < times; j++) {
                    num.re = old_re * old_re + old_im * old_im;
                    num.re = std::sqrt(num.re);
                    num.re += old_re;
                    num.im = num.re - old_re;
                    num.re /= 2;
                    num.re = std::sqrt(num.re);
                    num.im /= 2;
                    num.im = std::sqrt(num.im);
                    if (old_re < 0) num.im = -num.im;
                }
            }
            roots[i] = csub(p, num);
            maxDiff = std::max(maxDiff, cabs_(num));
        }
        if (maxDiff <= 0) break;
    }
    const double verySmallEps = 1e-100;
    for (int i = 0; i < n; i++)
        if (std::fabs(roots[i].im) < verySmallEps) roots[i].im = 0;
    for (; n < n0; n++) roots[n] = roots[n - 1];
    return n0;
}

/* EMEstimatorCallback::runKernel: x1,x2 = 5 normalised correspondences (x,y pairs).
 * Returns the number of real solutions; E (row-major, unit Frobenius norm, x2^T E x1 = 0)
 * written consecutively to Ebuf. */
int five_point(const double* x1, const double* x2, double* Ebuf) {
    /* design matrix rows: coefficient of E(r,c) (row-major) in x2^T E x1 */
    double At[9 * 9];
    std::memset(At, 0, sizeof(At));
    for (int i = 0; i < 5; i++) {
        double u1 = x1[2 * i], v1 = x1[2 * i + 1], u2 = x2[2 * i], v2 = x2[2 * i + 1];
        double* q = At + i * 9;
        q[0] = u2 * u1; q[1] = u2 * v1; q[2] = u2;
        q[3] = v2 * u1; q[4] = v2 * v1; q[5] = v2;
        q[6] = u1;      q[7] = v1;      q[8] = 1.0;
    }
    double W[5], V[5 * 5];
    jacobi_svd(At, 9, W, V, 5, 9, 5, 9); /* SVD::compute(Q, W, U, Vt, FULL_UV) on the 5x9 matrix */
    const double* EE[4] = {At + 5 * 9, At + 6 * 9, At + 7 * 9, At + 8 * 9}; /* null-space basis */

    /* E(x,y,z) = x*E0 + y*E1 + z*E2 + E3 as nine degree-1 polynomials */
    Poly3 Ep[3][3];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            Ep[r][c].c[1][0][0] = EE[0][r * 3 + c];
            Ep[r][c].c[0][1][0] = EE[1][r * 3 + c];
            Ep[r][c].c[0][0][1] = EE[2][r * 3 + c];
            Ep[r][c].c[0][0][0] = EE[3][r * 3 + c];
        }
    /* EEt = E E^T (degree 2) */
    Poly3 EEt[3][3];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            Poly3 s;
            for (int k = 0; k < 3; k++) s = padd(s, pmul(Ep[r][k], 1, Ep[c][k], 1), 1.0);
            EEt[r][c] = s;
        }
    Poly3 tr = padd(padd(EEt[0][0], EEt[1][1], 1.0), EEt[2][2], 1.0);
    Poly3 eq[10];
    /* det(E) */
    {
        Poly3 m0 = padd(pmul(Ep[1][1], 1, Ep[2][2], 1), pmul(Ep[1][2], 1, Ep[2][1], 1), -1.0);
        Poly3 m1 = padd(pmul(Ep[1][0], 1, Ep[2][2], 1), pmul(Ep[1][2], 1, Ep[2][0], 1), -1.0);
        Poly3 m2 = padd(pmul(Ep[1][0], 1, Ep[2][1], 1), pmul(Ep[1][1], 1, Ep[2][0], 1), -1.0);
        Poly3 d = pmul(Ep[0][0], 1, m0, 2);
        d = padd(d, pmul(Ep[0][1], 1, m1, 2), -1.0);
        d = padd(d, pmul(Ep[0][2], 1, m2, 2), 1.0);
        eq[0] = d;
    }
    /* 2 E E^T E - tr(E E^T) E */
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            Poly3 s;
            for (int k = 0; k < 3; k++) s = padd(s, pmul(EEt[r][k], 2, Ep[k][c], 1), 2.0);
            s = padd(s, pmul(tr, 2, Ep[r][c], 1), -1.0);
            eq[1 + r * 3 + c] = s;
        }
    double A[10][20];
    for (int e = 0; e < 10; e++)
        for (int m = 0; m < 20; m++) A[e][m] = eq[e].c[kMono[m][0]][kMono[m][1]][kMono[m][2]];
    if (g_pose_ainv_mode.load() == 1) {
        /* hal::LU64f (LUImpl<double>, eps = DBL_EPSILON * 100) on a copy of the left block with B = I, then the product */
        double L[10][10], X[10][10], R[10][10];
        for (int i = 0; i < 10; i++)
            for (int j = 0; j < 10; j++) { L[i][j] = A[i][j]; X[i][j] = i == j ? 1.0 : 0.0; }
        bool ok = true;
        for (int i = 0; i < 10 && ok; i++) {
            int k = i;
            for (int j = i + 1; j < 10; j++)
                if (std::fabs(L[j][i]) > std::fabs(L[k][i])) k = j;
            if (std::fabs(L[k][i]) < DBL_EPSILON * 100) { ok = false; break; }
            if (k != i) {
                for (int j = i; j < 10; j++) std::swap(L[i][j], L[k][j]);
                for (int j = 0; j < 10; j++) std::swap(X[i][j], X[k][j]);
            }
            const double d = -1 / L[i][i];
            for (int j = i + 1; j < 10; j++) {
                const double alpha = L[j][i] * d;
                for (int kk = i + 1; kk < 10; kk++) L[j][kk] += alpha * L[i][kk];
                for (int kk = 0; kk < 10; kk++) X[j][kk] += alpha * X[i][kk];
            }
        }
        if (ok) {
            for (int i = 9; i >= 0; i--)
                for (int j = 0; j < 10; j++) {
                    double sum = X[i][j];
                    for (int kk = i + 1; kk < 10; kk++) sum -= L[i][kk] * X[kk][j];
                    X[i][j] = sum / L[i][i];
                }
        } else {
            for (int i = 0; i < 10; i++)
                for (int j = 0; j < 10; j++) X[i][j] = 0;      /* Mat::inv of a singular matrix: the zero matrix */
        }
        for (int i = 0; i < 10; i++)
            for (int j = 0; j < 10; j++) {
                double sum = 0;
                for (int kk = 0; kk < 10; kk++) sum += X[i][kk] * A[kk][10 + j];
                R[i][j] = sum;
            }
        for (int i = 0; i < 10; i++)
            for (int j = 0; j < 10; j++) A[i][10 + j] = R[i][j];
    } else
    /* A = inv(A[:, :10]) * A[:, 10:]  by Gauss-Jordan with partial pivoting */
    for (int col = 0; col < 10; col++) {
        int piv = col;
        double best = std::fabs(A[col][col]);
        for (int r = col + 1; r < 10; r++)
            if (std::fabs(A[r][col]) > best) { best = std::fabs(A[r][col]); piv = r; }
        if (best < DBL_MIN) return 0;
        if (piv != col)
            for (int m = 0; m < 20; m++) std::swap(A[piv][m], A[col][m]);
        double inv = 1.0 / A[col][col];
        for (int m = col; m < 20; m++) A[col][m] *= inv;
        for (int r = 0; r < 10; r++) {
            if (r == col) continue;
            double f = A[r][col];
            if (f == 0) continue;
            for (int m = col; m < 20; m++) A[r][m] -= f * A[col][m];
        }
    }
    /* B(z): rows (x^2 z) - z (x^2), (y^2 z) - z (y^2), (xyz) - z (xy).  Row layout:
     * [x z^3, x z^2, x z, x | y z^3, y z^2, y z, y | z^4, z^3, z^2, z, 1] */
    double b[3][13];
    for (int i = 0; i < 3; i++) {
        const double* a1 = &A[i * 2 + 4][10];
        const double* a2 = &A[i * 2 + 5][10];
        double row1[13] = {0}, row2[13] = {0};
        for (int k = 0; k < 3; k++) { row1[1 + k] = a1[k]; row1[5 + k] = a1[3 + k]; }
        for (int k = 0; k < 4; k++) row1[9 + k] = a1[6 + k];
        for (int k = 0; k < 3; k++) { row2[k] = a2[k]; row2[4 + k] = a2[3 + k]; }
        for (int k = 0; k < 4; k++) row2[8 + k] = a2[6 + k];
        for (int k = 0; k < 13; k++) b[i][k] = row1[k] - row2[k];
    }
    /* det B(z): univariate polynomials, index = power of z */
    auto col_poly = [&](int row, int col, double* out /*[5]*/) {
        for (int k = 0; k < 5; k++) out[k] = 0;
        if (col < 2)
            for (int k = 0; k < 4; k++) out[3 - k] = b[row][col * 4 + k];
        else
            for (int k = 0; k < 5; k++) out[4 - k] = b[row][8 + k];
    };
    auto conv = [](const double* a, int da, const double* bb, int db, double* out) {
        for (int k = 0; k <= da + db; k++) out[k] = 0;
        for (int i = 0; i <= da; i++)
            for (int j = 0; j <= db; j++) out[i + j] += a[i] * bb[j];
    };
    double P[3][3][5];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) col_poly(r, c, P[r][c]);
    double cdet[11];
    for (int k = 0; k <= 10; k++) cdet[k] = 0;
    {
        /* expansion along column 2 (degree 4) of the 3x3: sum_r (-1)^r P[r][2] * minor_r */
        for (int r = 0; r < 3; r++) {
            int r0 = (r == 0) ? 1 : 0, r1 = (r == 2) ? 1 : 2;
            double m_a[7], m_b[7], minor[7], term[11];
            conv(P[r0][0], 3, P[r1][1], 3, m_a);
            conv(P[r0][1], 3, P[r1][0], 3, m_b);
            for (int k = 0; k < 7; k++) minor[k] = m_a[k] - m_b[k];
            conv(P[r][2], 4, minor, 6, term);
            double sgn = (r == 1) ? -1.0 : 1.0;
            for (int k = 0; k <= 10; k++) cdet[k] += sgn * term[k];
        }
    }
    Cx roots[16];
    solve_poly(cdet, 10, roots);
    int count = 0;
    for (int i = 0; i < 10; i++) {
        if (std::fabs(roots[i].im) > 1e-10) continue;
        double z1 = roots[i].re, z2 = z1 * z1, z3 = z2 * z1, z4 = z3 * z1;
        double bz[9];
        for (int j = 0; j < 3; j++) {
            const double* br = b[j];
            bz[j * 3 + 0] = br[0] * z3 + br[1] * z2 + br[2] * z1 + br[3];
            bz[j * 3 + 1] = br[4] * z3 + br[5] * z2 + br[6] * z1 + br[7];
            bz[j * 3 + 2] = br[8] * z4 + br[9] * z3 + br[10] * z2 + br[11] * z1 + br[12];
        }
        double w3[3], U3[9], Vt3[9];
        svd_square<3>(bz, w3, U3, Vt3); /* SVD::solveZ: last row of Vt */
        const double* xy1 = Vt3 + 6;
        if (std::fabs(xy1[2]) < 1e-10) continue;
        double xs = xy1[0] / xy1[2], ys = xy1[1] / xy1[2], zs = z1;
        double* E = Ebuf + count * 9;
        double nrm = 0;
        for (int k = 0; k < 9; k++) {
            E[k] = EE[0][k] * xs + EE[1][k] * ys + EE[2][k] * zs + EE[3][k];
            nrm += E[k] * E[k];
        }
        nrm = std::sqrt(nrm);
        for (int k = 0; k < 9; k++) E[k] /= nrm;
        count++;
    }
    return count;
}

/* normalisation used by findEssentialMat / recoverPose: (col - c) / f evaluated as the
 * MatExpr col*(1/f) + (-c*(1/f)) in double (A.3.1) */
void normalize_points(const float* q, int n, double f_x, double f_y, double c_x, double c_y, std::vector<double>& out) {
    out.resize((size_t)n * 2);
    double ax = 1. / f_x, bx = -c_x * (1. / f_x), ay = 1. / f_y, by = -c_y * (1. / f_y);
    for (int i = 0; i < n; i++) {
        out[2 * i] = (double)q[2 * i] * ax + bx;
        out[2 * i + 1] = (double)q[2 * i + 1] * ay + by;
    }
}

int ransac_update_num_iters(double p, double ep, int modelPoints, int maxIters) {
    p = std::max(p, 0.);
    p = std::min(p, 1.);
    ep = std::max(ep, 0.);
    ep = std::min(ep, 1.);
    double num = std::max(1. - p, DBL_MIN);
    double denom = 1. - std::pow(1. - ep, modelPoints);
    if (denom < DBL_MIN) return 0;
    num = std::log(num);
    denom = std::log(denom);
    return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : cvRoundD(num / denom);
}

/* Sampson error as float32 (EMEstimatorCallback::computeError) */
inline float sampson_err(const double* E, double u1, double v1, double u2, double v2) {
    double Ex1[3] = {E[0] * u1 + E[1] * v1 + E[2], E[3] * u1 + E[4] * v1 + E[5], E[6] * u1 + E[7] * v1 + E[8]};
    double Etx2[2] = {E[0] * u2 + E[3] * v2 + E[6], E[1] * u2 + E[4] * v2 + E[7]};
    double x2tEx1 = u2 * Ex1[0] + v2 * Ex1[1] + Ex1[2];
    double a = Ex1[0] * Ex1[0], b = Ex1[1] * Ex1[1], c = Etx2[0] * Etx2[0], d = Etx2[1] * Etx2[1];
    return (float)(x2tEx1 * x2tEx1 / (a + b + c + d));
}

/* icvTriangulatePoints for one point, double in / double out */
void triangulate_one(const double* P1, const double* P2, double x1, double y1, double x2, double y2, double X[4]) {
    double A[16];
    for (int k = 0; k < 4; k++) {
        A[0 * 4 + k] = x1 * P1[8 + k] - P1[0 + k];
        A[1 * 4 + k] = y1 * P1[8 + k] - P1[4 + k];
        A[2 * 4 + k] = x2 * P2[8 + k] - P2[0 + k];
        A[3 * 4 + k] = y2 * P2[8 + k] - P2[4 + k];
    }
    double W[4], U[16], Vt[16];
    svd_square<4>(A, W, U, Vt);
    for (int k = 0; k < 4; k++) X[k] = Vt[12 + k];
}

double det3(const double* M) {
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}
void mat3mul(const double* A, const double* B, double* C) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += A[i * 3 + k] * B[k * 3 + j];
            C[i * 3 + j] = s;
        }
}

}  // namespace

/* ================================================================== C ABI */
extern "C" {

void orc_default_orb_params(orc_orb_params* p) {
    p->nfeatures = 3000; /* libs/vo_utils.py:23 */
    p->scale_factor = 1.2f;
    p->nlevels = 8;
    p->edge_threshold = 31;
    p->patch_size = 31;
    p->fast_threshold = 20;
    p->blur_mode = 0;
    p->fma_atan = 1;
    p->fma_blur = 1;
    p->trig_mode = 0;
}

void orc_level_geometry(int w, int h, const orc_orb_params* p, int* lw, int* lh, float* scale, int* quota) {
    Geometry g = make_geometry(w, h, *p);
    for (int l = 0; l < p->nlevels; l++) {
        if (lw) lw[l] = g.w[l];
        if (lh) lh[l] = g.h[l];
        if (scale) scale[l] = g.scale[l];
        if (quota) quota[l] = g.quota[l];
    }
}

/* libs/vo_utils.py:94-95 -> ORB_Impl::detectAndCompute (A.1) */
int orc_orb_detect_compute(const uint8_t* img, int w, int h, int stride, const orc_orb_params* p, orc_orb_debug* dbg,
                           orc_keypoint* kps, uint8_t* desc, int cap, int* n) {
    if (!img || !p || !n || w <= 0 || h <= 0 || p->patch_size != 31 || p->nlevels < 1 || p->nlevels > 16) return -1;
    Geometry g = make_geometry(w, h, *p);
    std::vector<uint8_t> pyr(g.total), blurred(g.total);
    for (int y = 0; y < h; y++) std::memcpy(&pyr[(size_t)y * w], img + (size_t)y * stride, w);
    for (int l = 1; l < g.nlevels; l++)
        resize_linear_exact(&pyr[g.ofs[l - 1]], g.w[l - 1], g.h[l - 1], g.w[l - 1], &pyr[g.ofs[l]], g.w[l], g.h[l], g.w[l]);

    std::vector<KP> all;
    std::vector<int> counters(g.nlevels, 0);
    int fast_total = 0, sel1_total = 0;
    for (int l = 0; l < g.nlevels; l++) {
        std::vector<KP> kp;
        fast_detect(&pyr[g.ofs[l]], g.w[l], g.h[l], g.w[l], p->fast_threshold, p->edge_threshold, kp);
        if (dbg && dbg->fast_count) dbg->fast_count[l] = (int)kp.size();
        if (dbg && dbg->fast_xys)
            for (size_t i = 0; i < kp.size(); i++, fast_total++)
                if (fast_total < dbg->fast_cap) {
                    dbg->fast_xys[fast_total * 3 + 0] = (int)kp[i].x;
                    dbg->fast_xys[fast_total * 3 + 1] = (int)kp[i].y;
                    dbg->fast_xys[fast_total * 3 + 2] = (int)kp[i].response;
                }
        retain_best(kp, 2 * g.quota[l]); /* HARRIS_SCORE: keep 2x first */
        for (auto& k : kp) k.octave = l;
        counters[l] = (int)kp.size();
        if (dbg && dbg->sel1_count) dbg->sel1_count[l] = counters[l];
        all.insert(all.end(), kp.begin(), kp.end());
    }
    /* Harris responses, then per-level retainBest(quota) in the order the first cull left */
    std::vector<KP> kept;
    int offset = 0;
    for (int l = 0; l < g.nlevels; l++) {
        std::vector<KP> kp(all.begin() + offset, all.begin() + offset + counters[l]);
        offset += counters[l];
        for (auto& k : kp) {
            float fast_resp = k.response;
            k.response = harris_response(&pyr[g.ofs[l]], g.w[l], cvRoundF(k.x), cvRoundF(k.y));
            if (dbg && dbg->sel1_xys && sel1_total < dbg->sel1_cap) {
                dbg->sel1_xys[sel1_total * 3 + 0] = (int)k.x;
                dbg->sel1_xys[sel1_total * 3 + 1] = (int)k.y;
                dbg->sel1_xys[sel1_total * 3 + 2] = (int)fast_resp;
                if (dbg->sel1_harris) dbg->sel1_harris[sel1_total] = k.response;
            }
            sel1_total++;
        }
        retain_best(kp, g.quota[l]);
        if (dbg && dbg->sel2_count) dbg->sel2_count[l] = (int)kp.size();
        kept.insert(kept.end(), kp.begin(), kp.end());
    }
    for (auto& k : kept) k.angle = ic_angle(&pyr[g.ofs[k.octave]], g.w[k.octave], cvRoundF(k.x), cvRoundF(k.y), p->fma_atan != 0);
    /* blur every level, then describe */
    for (int l = 0; l < g.nlevels; l++)
        gaussian_blur7(&pyr[g.ofs[l]], g.w[l], g.h[l], g.w[l], &blurred[g.ofs[l]], g.w[l], p->blur_mode, p->fma_blur != 0);
    if (dbg && dbg->pyr) std::memcpy(dbg->pyr, pyr.data(), g.total);
    if (dbg && dbg->blurred) std::memcpy(dbg->blurred, blurred.data(), g.total);
    *n = (int)kept.size();
    if ((int)kept.size() > cap) return -3;
    for (size_t i = 0; i < kept.size(); i++) {
        const KP& k = kept[i];
        float sc = g.scale[k.octave];
        orc_keypoint o;
        o.x = k.x * sc; /* pt *= layerScale */
        o.y = k.y * sc;
        o.size = p->patch_size * sc;
        o.angle = k.angle;
        o.response = k.response;
        o.octave = k.octave;
        if (kps) kps[i] = o;
        if (desc) {
            float s = 1.f / sc;
            int cx = cvRoundF(o.x * s), cy = cvRoundF(o.y * s);
            orb_descriptor(&blurred[g.ofs[k.octave]], g.w[k.octave], cx, cy, k.angle, desc + i * 32, p->trig_mode);
        }
    }
    return 0;
}

/* north-star matcher: BFMatcher(NORM_HAMMING).knnMatch(k=2)  (A.2.1; stands in for
 * libs/vo_utils.py:98) */
int orc_match_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx1, int32_t* d1, int32_t* idx2, int32_t* d2) {
    if (nt < 2) return -1;
    for (int i = 0; i < nq; i++) {
        int bd[2] = {INT_MAX, INT_MAX}, bi[2] = {-1, -1};
        const uint64_t* a = (const uint64_t*)(q + (size_t)i * 32);
        uint64_t a0, a1, a2, a3;
        std::memcpy(&a0, a, 8); std::memcpy(&a1, a + 1, 8); std::memcpy(&a2, a + 2, 8); std::memcpy(&a3, a + 3, 8);
        for (int j = 0; j < nt; j++) {
            uint64_t b0, b1, b2, b3;
            const uint8_t* bp = t + (size_t)j * 32;
            std::memcpy(&b0, bp, 8); std::memcpy(&b1, bp + 8, 8); std::memcpy(&b2, bp + 16, 8); std::memcpy(&b3, bp + 24, 8);
            int d = __builtin_popcountll(a0 ^ b0) + __builtin_popcountll(a1 ^ b1) + __builtin_popcountll(a2 ^ b2) + __builtin_popcountll(a3 ^ b3);
            if (d < bd[1]) {
                if (d < bd[0]) {
                    bd[1] = bd[0]; bi[1] = bi[0];
                    bd[0] = d; bi[0] = j;
                } else {
                    bd[1] = d; bi[1] = j;
                }
            }
        }
        idx1[i] = bi[0]; d1[i] = bd[0]; idx2[i] = bi[1]; d2[i] = bd[1];
    }
    return 0;
}

/* libs/vo_utils.py:101-104 */
int orc_ratio_filter(const int32_t* d1, const int32_t* d2, int nq, double ratio, int32_t* keep, int* nkeep) {
    int k = 0;
    for (int i = 0; i < nq; i++)
        if ((double)(float)d1[i] < ratio * (double)(float)d2[i]) keep[k++] = i;
    *nkeep = k;
    return 0;
}

int orc_five_point(const double* x1, const double* x2, double* Ebuf) { return five_point(x1, x2, Ebuf); }
int orc_solve_poly10(const double* coeffs, double* roots_re_im) {
    Cx r[16];
    solve_poly(coeffs, 10, r);
    for (int i = 0; i < 10; i++) { roots_re_im[2 * i] = r[i].re; roots_re_im[2 * i + 1] = r[i].im; }
    return 10;
}

/* libs/vo_utils.py:141 -> findEssentialMat -> RANSACPointSetRegistrator::run (A.3) */
int orc_find_essential_ransac(const float* q1, const float* q2, int n, const double K[9], double prob, double thr, int max_iters,
                              double E[9], uint8_t* mask, int* n_inliers, int* iters_run) {
    const int modelPoints = 5;
    if (iters_run) *iters_run = 0;
    if (n_inliers) *n_inliers = 0;
    if (n < modelPoints) return -1;
    double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    std::vector<double> m1, m2;
    normalize_points(q1, n, fx, fy, cx, cy, m1);
    normalize_points(q2, n, fx, fy, cx, cy, m2);
    double threshold = thr / ((fx + fy) / 2);
    const float t = (float)(threshold * threshold);
    int niters = std::max(max_iters, 1);
    double models[10 * 9];
    if (n == modelPoints) {
        int nm = five_point(m1.data(), m2.data(), models);
        if (nm <= 0) return -1;
        /* bestModel is the whole (3*nm) x 3 stack; cv2 returns it as is.  Keep the first. */
        std::memcpy(E, models, 9 * sizeof(double));
        if (mask) std::memset(mask, 1, n);
        if (n_inliers) *n_inliers = n;
        return 0;
    }
    CvRNG rng((uint64_t)-1);
    std::vector<uint8_t> cur(n), best(n, 0);
    int maxGood = 0, iter;
    double bestE[9];
    for (iter = 0; iter < niters; iter++) {
        int idx[5];
        for (int i = 0; i < modelPoints; i++) {
            int idx_i;
            for (idx_i = rng.uniform(0, n); std::find(idx, idx + i, idx_i) != idx + i; idx_i = rng.uniform(0, n)) {}
            idx[i] = idx_i;
        }
        double s1[10], s2[10];
        for (int i = 0; i < 5; i++) {
            s1[2 * i] = m1[2 * idx[i]]; s1[2 * i + 1] = m1[2 * idx[i] + 1];
            s2[2 * i] = m2[2 * idx[i]]; s2[2 * i + 1] = m2[2 * idx[i] + 1];
        }
        int nmodels = five_point(s1, s2, models);
        if (nmodels <= 0) continue;
        for (int mi = 0; mi < nmodels; mi++) {
            const double* Em = models + mi * 9;
            int good = 0;
            for (int i = 0; i < n; i++) {
                float err = sampson_err(Em, m1[2 * i], m1[2 * i + 1], m2[2 * i], m2[2 * i + 1]);
                int f = err <= t;
                cur[i] = (uint8_t)f;
                good += f;
            }
            if (good > std::max(maxGood, modelPoints - 1)) {
                std::swap(cur, best);
                std::memcpy(bestE, Em, sizeof(bestE));
                maxGood = good;
                niters = ransac_update_num_iters(prob, (double)(n - good) / n, modelPoints, niters);
            }
        }
    }
    if (iters_run) *iters_run = iter;
    if (maxGood <= 0) return -1;
    std::memcpy(E, bestE, sizeof(bestE));
    if (mask) std::memcpy(mask, best.data(), n);
    if (n_inliers) *n_inliers = maxGood;
    return 0;
}

/* libs/vo_utils.py:143 -> recoverPose(E, points1, points2, cameraMatrix, R, t) (A.4) */
static int recover_pose_impl(const double E[9], const float* q1, const float* q2, int n, const double K[9], double dist_thresh,
                             double R[9], double t[3], int* n_good, int votes[4]) {
    double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    std::vector<double> m1, m2;
    normalize_points(q1, n, fx, fy, cx, cy, m1);
    normalize_points(q2, n, fx, fy, cx, cy, m2);
    /* decomposeEssentialMat */
    double D[3], U[9], Vt[9];
    svd_square<3>(E, D, U, Vt);
    if (det3(U) < 0) for (int i = 0; i < 9; i++) U[i] *= -1.;
    if (det3(Vt) < 0) for (int i = 0; i < 9; i++) Vt[i] *= -1.;
    const double Wm[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1}, Wt[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1};
    double tmp[9], R1[9], R2[9], tv[3];
    mat3mul(U, Wm, tmp); mat3mul(tmp, Vt, R1);
    mat3mul(U, Wt, tmp); mat3mul(tmp, Vt, R2);
    for (int i = 0; i < 3; i++) tv[i] = U[i * 3 + 2];
    const double P0[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    int good[4];
    for (int c = 0; c < 4; c++) {
        const double* Rc = (c & 1) ? R2 : R1;
        double sg = (c >= 2) ? -1.0 : 1.0;
        double P[12];
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) P[i * 4 + j] = Rc[i * 3 + j];
            P[i * 4 + 3] = sg * tv[i];
        }
        int cnt = 0;
        for (int i = 0; i < n; i++) {
            double Q[4];
            triangulate_one(P0, P, m1[2 * i], m1[2 * i + 1], m2[2 * i], m2[2 * i + 1], Q);
            bool m = (Q[2] * Q[3]) > 0;
            Q[0] /= Q[3]; Q[1] /= Q[3]; Q[2] /= Q[3]; Q[3] /= Q[3];
            m = m && (Q[2] < dist_thresh);
            double z = P[8] * Q[0] + P[9] * Q[1] + P[10] * Q[2] + P[11] * Q[3];
            m = m && (z > 0) && (z < dist_thresh);
            cnt += m ? 1 : 0;
        }
        good[c] = cnt;
    }
    int sel;
    if (good[0] >= good[1] && good[0] >= good[2] && good[0] >= good[3]) sel = 0;
    else if (good[1] >= good[0] && good[1] >= good[2] && good[1] >= good[3]) sel = 1;
    else if (good[2] >= good[0] && good[2] >= good[1] && good[2] >= good[3]) sel = 2;
    else sel = 3;
    std::memcpy(R, (sel & 1) ? R2 : R1, 9 * sizeof(double));
    for (int i = 0; i < 3; i++) t[i] = (sel >= 2 ? -1.0 : 1.0) * tv[i];
    if (n_good) *n_good = good[sel];
    if (votes) std::memcpy(votes, good, sizeof(good));
    return 0;
}

int orc_recover_pose(const double E[9], const float* q1, const float* q2, int n, const double K[9], double dist_thresh,
                     double R[9], double t[3], int* n_good) {
    return recover_pose_impl(E, q1, q2, n, K, dist_thresh, R, t, n_good, nullptr);
}

/* the same call, also returning the four candidates' counts (test hook: the GPU derives two of them) */
int orc_recover_pose_votes(const double E[9], const float* q1, const float* q2, int n, const double K[9], double dist_thresh,
                           double R[9], double t[3], int* n_good, int votes[4]) {
    return recover_pose_impl(E, q1, q2, n, K, dist_thresh, R, t, n_good, votes);
}

/* libs/vo_utils.py:160 -> triangulatePoints (A.5) */
int orc_triangulate_f32(const double P1[12], const double P2[12], const float* x1, const float* x2, int n, float* X4) {
    for (int i = 0; i < n; i++) {
        double X[4];
        triangulate_one(P1, P2, (double)x1[2 * i], (double)x1[2 * i + 1], (double)x2[2 * i], (double)x2[2 * i + 1], X);
        for (int k = 0; k < 4; k++) X4[(size_t)k * n + i] = (float)X[k];
    }
    return 0;
}

/* libs/vo_utils.py:83-110 + 136-145 for one frame pair */
int orc_pair_run(const uint8_t* imgA, const uint8_t* imgB, int w, int h, int stride, const orc_orb_params* p, const double K[9],
                 double ratio, double T[16], int32_t counts[4]) {
    int cap = p->nfeatures * 2 + 4096;
    std::vector<orc_keypoint> k1(cap), k2(cap);
    std::vector<uint8_t> f1((size_t)cap * 32), f2((size_t)cap * 32);
    int n1 = 0, n2 = 0;
    int rc = orc_orb_detect_compute(imgA, w, h, stride, p, nullptr, k1.data(), f1.data(), cap, &n1);
    if (rc) return rc;
    rc = orc_orb_detect_compute(imgB, w, h, stride, p, nullptr, k2.data(), f2.data(), cap, &n2);
    if (rc) return rc;
    counts[0] = n1; counts[1] = n2; counts[2] = 0; counts[3] = 0;
    if (n2 < 2) return -2;
    std::vector<int32_t> i1(n1), d1(n1), i2(n1), d2(n1), keep(n1);
    orc_match_knn2(f1.data(), n1, f2.data(), n2, i1.data(), d1.data(), i2.data(), d2.data());
    int nk = 0;
    orc_ratio_filter(d1.data(), d2.data(), n1, ratio, keep.data(), &nk);
    counts[2] = nk;
    std::vector<float> q1((size_t)nk * 2), q2((size_t)nk * 2);
    for (int i = 0; i < nk; i++) {
        q1[2 * i] = k1[keep[i]].x; q1[2 * i + 1] = k1[keep[i]].y;
        q2[2 * i] = k2[i1[keep[i]]].x; q2[2 * i + 1] = k2[i1[keep[i]]].y;
    }
    double E[9], R[9], t[3];
    std::vector<uint8_t> mask(nk > 0 ? nk : 1);
    int ninl = 0;
    if (orc_find_essential_ransac(q1.data(), q2.data(), nk, K, 0.999, 1.0, 1000, E, mask.data(), &ninl, nullptr)) return -2;
    counts[3] = ninl;
    int ngood = 0;
    orc_recover_pose(E, q1.data(), q2.data(), nk, K, 50.0, R, t, &ngood);
    for (int i = 0; i < 16; i++) T[i] = 0;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) T[i * 4 + j] = R[i * 3 + j];
        T[i * 4 + 3] = t[i];
    }
    T[15] = 1;
    return 0;
}

int orc_retain_best(const float* keys, int n, int n_points, int32_t* out_idx, int* n_out) {
    std::vector<KeyIdx> v(n);
    for (int i = 0; i < n; i++) v[i] = KeyIdx{keys[i], i};
    retain_best(v, n_points);
    for (size_t i = 0; i < v.size(); i++) out_idx[i] = v[i].idx;
    *n_out = (int)v.size();
    return 0;
}

void orc_set_pose_variant(int hypot_mode, int ainv_mode) {
    g_pose_hypot_mode.store(hypot_mode ? 1 : 0);
    g_pose_ainv_mode.store(ainv_mode ? 1 : 0);
}

}  // extern "C"
