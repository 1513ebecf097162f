// vo_pose.hip -- five-point RANSAC, recoverPose and DLT triangulation on MI355X (gfx950).
//
// Stands behind cv2.findEssentialMat(q1,q2,K,RANSAC,0.999,1.0,None), cv2.recoverPose and
// cv2.triangulatePoints at /root/reference/libs/vo_utils.py:141,143,160.
//
// RANSAC keeps OpenCV's sequential semantics while scoring hypotheses in parallel:
//   k_ransac_sample  one lane per pair replays cv::RNG(2^64-1) and draws the 5-tuples of
//                    the next VO_HYP_ROUND iterations (they depend only on the match count);
//   k_ransac_solve   one lane per (pair, hypothesis): Nister five-point solver in fp64
//                    (Jacobi SVD null space, 10x20 elimination, degree-10 Durand-Kerner);
//   k_ransac_score   one wavefront per (pair, hypothesis): float32 Sampson error of every
//                    match against every model, inlier counts;
//   k_ransac_scan    one lane per pair walks the (iteration, model) count table in order:
//                    first strictly better model wins, adaptive niters (RANSACUpdateNumIters).
// Rounds repeat (host-checked) until every pair's iteration counter reaches its niters.
// All arithmetic is fp64 (fp32 only where OpenCV casts), compiled with -ffp-contract=off.
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <utility>

#include "vo_internal.h"

// ------------------------------------------------------------------ cv::RNG
struct DevRNG {
    unsigned long long state;
    __device__ unsigned next() {
        state = (unsigned long long)(unsigned)state * 4164903690u + (unsigned)(state >> 32);
        return (unsigned)state;
    }
    __device__ int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

// ------------------------------------------------------------------ JacobiSVDImpl_<double>
// One-sided Jacobi on the N rows (length M) of At; Vt (N x N) accumulates the rotations;
// rows N..N1-1 are completed with RNG(0x12345678) as OpenCV does for FULL_UV.
template <int M, int N, int N1>
__device__ void jacobi_svd(double* At, double* W, double* Vt) {
    const double minval = DBL_MIN, eps = DBL_EPSILON * 10;
    const int max_iter = M > 30 ? M : 30;
    double c, s, sd;
    for (int i = 0; i < N; i++) {
        sd = 0;
        for (int k = 0; k < M; k++) { double t = At[i * M + k]; sd += t * t; }
        W[i] = sd;
        for (int k = 0; k < N; k++) Vt[i * N + k] = 0;
        Vt[i * N + i] = 1;
    }
    for (int iter = 0; iter < max_iter; iter++) {
        bool changed = false;
#pragma unroll
        for (int i = 0; i < N - 1; i++)
#pragma unroll
            for (int j = i + 1; j < N; j++) {
                double *Ai = At + i * M, *Aj = At + j * M;
                double a = W[i], p = 0, b = W[j];
#pragma unroll
                for (int k = 0; k < M; k++) p += Ai[k] * Aj[k];
                if (fabs(p) <= eps * sqrt(a * b)) continue;
                p *= 2;
                double beta = a - b, gamma = sqrt(p * p + beta * beta);
                if (beta < 0) {
                    double delta = (gamma - beta) * 0.5;
                    s = sqrt(delta / gamma);
                    c = p / (gamma * s * 2);
                } else {
                    c = sqrt((gamma + beta) / (gamma * 2));
                    s = p / (gamma * c * 2);
                }
                a = b = 0;
#pragma unroll
                for (int k = 0; k < M; k++) {
                    double t0 = c * Ai[k] + s * Aj[k];
                    double t1 = -s * Ai[k] + c * Aj[k];
                    Ai[k] = t0; Aj[k] = t1;
                    a += t0 * t0; b += t1 * t1;
                }
                W[i] = a; W[j] = b;
                changed = true;
                double *Vi = Vt + i * N, *Vj = Vt + j * N;
#pragma unroll
                for (int k = 0; k < N; k++) {
                    double t0 = c * Vi[k] + s * Vj[k];
                    double t1 = -s * Vi[k] + c * Vj[k];
                    Vi[k] = t0; Vj[k] = t1;
                }
            }
        if (!changed) break;
    }
    for (int i = 0; i < N; i++) {
        sd = 0;
        for (int k = 0; k < M; k++) { double t = At[i * M + k]; sd += t * t; }
        W[i] = sqrt(sd);
    }
    for (int i = 0; i < N - 1; i++) {
        int j = i;
        for (int k = i + 1; k < N; k++)
            if (W[j] < W[k]) j = k;
        if (i != j) {
            double t = W[i]; W[i] = W[j]; W[j] = t;
            for (int k = 0; k < M; k++) { t = At[i * M + k]; At[i * M + k] = At[j * M + k]; At[j * M + k] = t; }
            for (int k = 0; k < N; k++) { t = Vt[i * N + k]; Vt[i * N + k] = Vt[j * N + k]; Vt[j * N + k] = t; }
        }
    }
    DevRNG rng{0x12345678ull};
    for (int i = 0; i < N1; i++) {
        sd = i < N ? W[i] : 0;
        for (int ii = 0; ii < 100 && sd <= minval; ii++) {
            const double val0 = 1. / M;
            for (int k = 0; k < M; k++) At[i * M + k] = (rng.next() & 256) != 0 ? val0 : -val0;
            for (int iter = 0; iter < 2; iter++)
                for (int j = 0; j < i; j++) {
                    sd = 0;
                    for (int k = 0; k < M; k++) sd += At[i * M + k] * At[j * M + k];
                    double asum = 0;
                    for (int k = 0; k < M; k++) {
                        double t = At[i * M + k] - sd * At[j * M + k];
                        At[i * M + k] = t;
                        asum += fabs(t);
                    }
                    asum = asum > eps * 100 ? 1 / asum : 0;
                    for (int k = 0; k < M; k++) At[i * M + k] *= asum;
                }
            sd = 0;
            for (int k = 0; k < M; k++) { double t = At[i * M + k]; sd += t * t; }
            sd = sqrt(sd);
        }
        s = sd > minval ? 1 / sd : 0.;
        for (int k = 0; k < M; k++) At[i * M + k] *= s;
    }
}

// SVD::compute of a square MxM matrix: U, W, Vt row-major
template <int M>
__device__ void svd_square(const double* A, double* W, double* U, double* Vt) {
    double At[M * M];
    for (int i = 0; i < M; i++)
        for (int j = 0; j < M; j++) At[i * M + j] = A[j * M + i];
    jacobi_svd<M, M, M>(At, W, Vt);
    for (int i = 0; i < M; i++)
        for (int j = 0; j < M; j++) U[i * M + j] = At[j * M + i];
}

// ------------------------------------------------------------------ polynomials in (x,y,z), degree <= 3
// Compact forms: a polynomial of degree <= D holds exactly its (D+1)(D+2)(D+3)/6 coefficients (4 / 10 / 20), in the order
// the nested loops "for i, for j, for k with i + j + k <= D" visit x^i y^j z^k.  Every loop over monomials is a compile-time
// loop (sfor), so every coefficient access is a constant index and the polynomials live in registers.  (Until round 3 a
// polynomial was a double[4][4][4] walked with run-time bounds: 29 of them, 15 KB of private memory per lane, and 239 MB of
// scratch writes per launch of 1024 hypotheses.)  The arithmetic is the one of the loops it replaces, term by term and in
// the same order; the loops skipped a term whose left coefficient was exactly 0, which only differs from adding it when
// the right coefficient is not finite (0 * inf): the null-space vectors these are built from have unit norm.
template <int... Is, class F>
__device__ __forceinline__ void sfor_impl(std::integer_sequence<int, Is...>, F&& f) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) { sfor_impl(std::make_integer_sequence<int, N>{}, f); }

struct Mono { int i, j, k; };
constexpr int poly_size(int D) { return (D + 1) * (D + 2) * (D + 3) / 6; }
template <int D>
constexpr Mono mono_of(int n) {
    int m = 0;
    for (int i = 0; i <= D; i++)
        for (int j = 0; i + j <= D; j++)
            for (int k = 0; i + j + k <= D; k++) {
                if (m == n) return Mono{i, j, k};
                m++;
            }
    return Mono{-1, -1, -1};
}
template <int D>
constexpr int mono_idx(int i, int j, int k) {
    int m = 0;
    for (int a = 0; a <= D; a++)
        for (int b = 0; a + b <= D; b++)
            for (int c = 0; a + b + c <= D; c++) {
                if (a == i && b == j && c == k) return m;
                m++;
            }
    return -1;
}
template <int D>
struct Poly { static constexpr int N = poly_size(D); double c[N]; };

template <int D>
__device__ __forceinline__ void pzero(Poly<D>& r) { sfor<Poly<D>::N>([&](auto I) { r.c[I] = 0; }); }
// r = a * b
template <int DA, int DB>
__device__ __forceinline__ void pmul(Poly<DA + DB>& r, const Poly<DA>& a, const Poly<DB>& b) {
    pzero(r);
    sfor<Poly<DA>::N>([&](auto I) {
        constexpr Mono ma = mono_of<DA>(I);
        const double av = a.c[I];
        sfor<Poly<DB>::N>([&](auto J) {
            constexpr Mono mb = mono_of<DB>(J);
            constexpr int o = mono_idx<DA + DB>(ma.i + mb.i, ma.j + mb.j, ma.k + mb.k);
            r.c[o] += av * b.c[J];
        });
    });
}
// r = a + sb * b (r may be a)
template <int D>
__device__ __forceinline__ void padd(Poly<D>& r, const Poly<D>& a, const Poly<D>& b, double sb) {
    sfor<Poly<D>::N>([&](auto I) { r.c[I] = a.c[I] + sb * b.c[I]; });
}
// the linear polynomial E0 x + E1 y + E2 z + E3 of one entry of the null-space basis (row-major 3x3 matrices, 9 apart)
__device__ __forceinline__ Poly<1> plin(double e0, double e1, double e2, double e3) {
    Poly<1> r;
    r.c[mono_idx<1>(1, 0, 0)] = e0; r.c[mono_idx<1>(0, 1, 0)] = e1; r.c[mono_idx<1>(0, 0, 1)] = e2; r.c[mono_idx<1>(0, 0, 0)] = e3;
    return r;
}
// the 20 monomials in the column order of the 10 x 20 constraint matrix
constexpr int c_mono[20][3] = {{3, 0, 0}, {0, 3, 0}, {2, 1, 0}, {1, 2, 0}, {2, 0, 1}, {2, 0, 0}, {0, 2, 1}, {0, 2, 0}, {1, 1, 1}, {1, 1, 0},
                               {1, 0, 2}, {1, 0, 1}, {1, 0, 0}, {0, 1, 2}, {0, 1, 1}, {0, 1, 0}, {0, 0, 3}, {0, 0, 2}, {0, 0, 1}, {0, 0, 0}};
// q's coefficients as a row of that matrix
template <class ROW>
__device__ __forceinline__ void poly_to_row(const Poly<3>& q, ROW&& row) {
    sfor<20>([&](auto M) { row[M] = q.c[mono_idx<3>(c_mono[M][0], c_mono[M][1], c_mono[M][2])]; });
}
// det(E) = 0 as a cubic: E rows given as linear polynomials
__device__ __forceinline__ void poly_det(Poly<3>& d, const Poly<1> (&E)[3][3]) {
    Poly<2> tmp, tmp2, m0, m1, m2;
    Poly<3> t3;
    pmul<1, 1>(tmp, E[1][1], E[2][2]); pmul<1, 1>(tmp2, E[1][2], E[2][1]); padd(m0, tmp, tmp2, -1.0);
    pmul<1, 1>(tmp, E[1][0], E[2][2]); pmul<1, 1>(tmp2, E[1][2], E[2][0]); padd(m1, tmp, tmp2, -1.0);
    pmul<1, 1>(tmp, E[1][0], E[2][1]); pmul<1, 1>(tmp2, E[1][1], E[2][0]); padd(m2, tmp, tmp2, -1.0);
    pmul<1, 2>(d, E[0][0], m0);
    pmul<1, 2>(t3, E[0][1], m1); padd(d, d, t3, -1.0);
    pmul<1, 2>(t3, E[0][2], m2); padd(d, d, t3, 1.0);
}
// entry (r, c') of E E^T: sum_k Er[k] * Ec[k] over two rows of E
__device__ __forceinline__ void poly_eet(Poly<2>& s, const Poly<1> (&Er)[3], const Poly<1> (&Ec)[3]) {
    Poly<2> tmp;
    pzero(s);
    sfor<3>([&](auto K) {
        pmul<1, 1>(tmp, Er[K], Ec[K]);
        padd(s, s, tmp, 1.0);
    });
}
// entry (r, c) of 2 E E^T E - trace(E E^T) E: EEr = row r of E E^T, Ecol = column c of E, Erc = E[r][c]
__device__ __forceinline__ void poly_trace_eq(Poly<3>& q, const Poly<2> (&EEr)[3], const Poly<1> (&Ecol)[3], const Poly<2>& tr, const Poly<1>& Erc) {
    Poly<3> tmp;
    pzero(q);
    sfor<3>([&](auto K) {
        pmul<2, 1>(tmp, EEr[K], Ecol[K]);
        padd(q, q, tmp, 2.0);
    });
    pmul<2, 1>(tmp, tr, Erc);
    padd(q, q, tmp, -1.0);
}

struct Cx { double re, im; };
__device__ __forceinline__ Cx cmul(Cx a, Cx b) { return Cx{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ Cx cadd(Cx a, Cx b) { return Cx{a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ Cx csub(Cx a, Cx b) { return Cx{a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ Cx cdiv(Cx a, Cx b) {
    double t = 1. / (b.re * b.re + b.im * b.im);
    return Cx{(a.re * b.re + a.im * b.im) * t, (-a.re * b.im + a.im * b.re) * t};
}

#define DK_T 64  // threads per solver workgroup
#ifndef DK_SWEEPS
#define DK_SWEEPS 300  // cv::solvePoly's maxIters at the reference's call (timing experiments lower it)
#endif
#ifndef VO_WIDE_SOLVE_MAX_HYP
#ifndef VO_WIDE_SOLVE_ASYNC_MAX_HYP
#define VO_WIDE_SOLVE_ASYNC_MAX_HYP 512
#endif
#define VO_WIDE_SOLVE_MAX_HYP 2048   // hypotheses per round up to which the 16-lanes-per-hypothesis solver runs (measured crossover: 1.73 against 2.05 ms at 2048, 2.65 against 2.07 ms at 4096)
#endif
// -DFP_CLOCKS (experiment builds, tools/exp_solver_clocks.py): the solver stamps the cycle counter at its phase boundaries into
// the unused tail of the model buffer, which the debug hook then returns
#ifdef FP_CLOCKS
#define FP_CLOCKS_ON 1
#define FP_TICK(i) Ebuf[80 + (i)] = (double)__builtin_readcyclecounter()
#else
#define FP_CLOCKS_ON 0
#define FP_TICK(i)
#endif
#ifndef DK_SWEEP_MODE
#define DK_SWEEP_MODE 1   // 0: round 1's branchy root updates (A/B runs); 1: straight-line updates with a rare redo
#endif

// cv::solvePoly(coeffs, roots, 300): Durand-Kerner, in-place (Gauss-Seidel) updates, start
// (1+i)^k.  Degree-10 fast path: roots and coefficients live in registers (every index is a
// compile-time constant after unrolling), so a sweep is two independent fp64 chains per root
// (Horner numerator, product denominator) with no memory traffic.  The arithmetic and its
// order are exactly those of the generic loop below.
__device__ __forceinline__ void dk_update_root(const double (&coef)[11], double (&rre)[10], double (&rim)[10], const int i, double& maxDiff) {
    const Cx p{rre[i], rim[i]};
    Cx num{coef[10], 0}, denom{coef[10], 0};
    int num_same_root = 1;
#pragma unroll
    for (int j = 0; j < 10; j++) {
        num = cadd(cmul(num, p), Cx{coef[9 - j], 0});
        if (j != i) {
            const Cx d = csub(p, Cx{rre[j], rim[j]});
            if (d.re != 0 || d.im != 0) denom = cmul(denom, d);
            else num_same_root++;
        }
    }
    num = cdiv(num, denom);
    if (num_same_root > 1 && (num_same_root % 2) == 0) {
        const double old_re = num.re, old_im = num.im;
        const int times = num_same_root / 2;
        for (int j = 0; j < times; j++) {
            num.re = old_re * old_re + old_im * old_im;
            num.re = sqrt(num.re);
            num.re += old_re;
            num.im = num.re - old_re;
            num.re /= 2;
            num.re = sqrt(num.re);
            num.im /= 2;
            num.im = sqrt(num.im);
            if (old_re < 0) num.im = -num.im;
        }
    }
    rre[i] = p.re - num.re;
    rim[i] = p.im - num.im;
    maxDiff = fmax(maxDiff, sqrt(num.re * num.re + num.im * num.im));
}

// One root's update from arrays in memory, index at run time: the rare path of dk_update_root_fast below.
__device__ __noinline__ void dk_update_root_mem(const double* coef, double* rre, double* rim, int i, double* maxSq) {
    const Cx p{rre[i], rim[i]};
    Cx num{coef[10], 0}, denom{coef[10], 0};
    int num_same_root = 1;
    for (int j = 0; j < 10; j++) {
        num = cadd(cmul(num, p), Cx{coef[9 - j], 0});
        if (j != i) {
            const Cx d = csub(p, Cx{rre[j], rim[j]});
            if (d.re != 0 || d.im != 0) denom = cmul(denom, d);
            else num_same_root++;
        }
    }
    num = cdiv(num, denom);
    if (num_same_root > 1 && (num_same_root % 2) == 0) {
        const double old_re = num.re, old_im = num.im;
        const int times = num_same_root / 2;
        for (int j = 0; j < times; j++) {
            num.re = old_re * old_re + old_im * old_im;
            num.re = sqrt(num.re);
            num.re += old_re;
            num.im = num.re - old_re;
            num.re /= 2;
            num.re = sqrt(num.re);
            num.im /= 2;
            num.im = sqrt(num.im);
            if (old_re < 0) num.im = -num.im;
        }
    }
    rre[i] = p.re - num.re;
    rim[i] = p.im - num.im;
    *maxSq = fmax(*maxSq, num.re * num.re + num.im * num.im);
}

// dk_update_root without its 9 "is this factor exactly zero" branches per root (they cost more instructions than the
// arithmetic they guard and cut the update into ten basic blocks).  A factor (0, 0) leaves the product (+-0, +-0) or NaN
// whatever follows, so a product that is neither proves that no factor was zero, and then this straight-line code has
// done exactly what the branchy loop does (num_same_root stays 1).  Otherwise (coincident roots, or a product that
// underflowed) the update is redone by the generic code above.  maxSq tracks |step|^2, not |step|: the sweep loop only
// asks whether the largest step is > 0, and sqrt(x) > 0 exactly when x > 0.
__device__ __forceinline__ void dk_update_root_fast(const double (&coef)[11], double (&rre)[10], double (&rim)[10], const int i, double& maxSq) {
    const Cx p{rre[i], rim[i]};
    Cx num{coef[10], 0}, denom{coef[10], 0};
#pragma unroll
    for (int j = 0; j < 10; j++) {
        num = cadd(cmul(num, p), Cx{coef[9 - j], 0});
        if (j != i) denom = cmul(denom, csub(p, Cx{rre[j], rim[j]}));
    }
    if (__builtin_expect((denom.re != 0 || denom.im != 0) && denom.re == denom.re && denom.im == denom.im, 1)) {
        num = cdiv(num, denom);
        rre[i] = p.re - num.re;
        rim[i] = p.im - num.im;
        maxSq = fmax(maxSq, num.re * num.re + num.im * num.im);
    } else {
        double t[31];
        for (int k = 0; k < 10; k++) { t[k] = rre[k]; t[10 + k] = rim[k]; }
        for (int k = 0; k <= 10; k++) t[20 + k] = coef[k];
        double m = maxSq;
        dk_update_root_mem(t + 20, t, t + 10, i, &m);
        rre[i] = t[i]; rim[i] = t[10 + i];
        maxSq = m;
    }
}

// generic degree (leading coefficients ~ 0): same algorithm on local arrays
__device__ __noinline__ void solve_poly_generic(const double* coeffs_in, int n, double* rre, double* rim) {
    Cx p{1, 0};
    const Cx r{1, 1};
    for (int i = 0; i < n; i++) {
        rre[i] = p.re; rim[i] = p.im;
        p = cmul(p, r);
    }
    const double lead = coeffs_in[n];
    for (int iter = 0; iter < 300; iter++) {
        double maxDiff = 0;
        for (int i = 0; i < n; i++) {
            p = Cx{rre[i], rim[i]};
            Cx num{lead, 0}, denom{lead, 0};
            int num_same_root = 1;
            for (int j = 0; j < n; j++) {
                num = cadd(cmul(num, p), Cx{coeffs_in[n - j - 1], 0});
                if (j != i) {
                    Cx d = csub(p, Cx{rre[j], rim[j]});
                    if (d.re != 0 || d.im != 0) denom = cmul(denom, d);
                    else num_same_root++;
                }
            }
            num = cdiv(num, denom);
            if (num_same_root > 1 && (num_same_root % 2) == 0) {
                const double old_re = num.re, old_im = num.im;
                const int times = num_same_root / 2;
                for (int j = 0; j < times; j++) {
                    num.re = old_re * old_re + old_im * old_im;
                    num.re = sqrt(num.re);
                    num.re += old_re;
                    num.im = num.re - old_re;
                    num.re /= 2;
                    num.re = sqrt(num.re);
                    num.im /= 2;
                    num.im = sqrt(num.im);
                    if (old_re < 0) num.im = -num.im;
                }
            }
            rre[i] = p.re - num.re;
            rim[i] = p.im - num.im;
            maxDiff = fmax(maxDiff, sqrt(num.re * num.re + num.im * num.im));
        }
        if (maxDiff <= 0) break;
    }
}

__device__ void solve_poly10(const double* coeffs_in, double* out_re, double* out_im) {
    int n = 10;
    for (; n > 1; n--)
        if (fabs(coeffs_in[n]) > DBL_EPSILON) break;
    if (n == 10) {
        double coef[11], rre[10], rim[10];
#pragma unroll
        for (int i = 0; i <= 10; i++) coef[i] = coeffs_in[i];
        {
            Cx p{1, 0};
            const Cx r{1, 1};
#pragma unroll
            for (int i = 0; i < 10; i++) {
                rre[i] = p.re; rim[i] = p.im;
                p = cmul(p, r);
            }
        }
#if DK_SWEEP_MODE == 0
        for (int iter = 0; iter < DK_SWEEPS; iter++) {
            double maxDiff = 0;
            dk_update_root(coef, rre, rim, 0, maxDiff);
            dk_update_root(coef, rre, rim, 1, maxDiff);
            dk_update_root(coef, rre, rim, 2, maxDiff);
            dk_update_root(coef, rre, rim, 3, maxDiff);
            dk_update_root(coef, rre, rim, 4, maxDiff);
            dk_update_root(coef, rre, rim, 5, maxDiff);
            dk_update_root(coef, rre, rim, 6, maxDiff);
            dk_update_root(coef, rre, rim, 7, maxDiff);
            dk_update_root(coef, rre, rim, 8, maxDiff);
            dk_update_root(coef, rre, rim, 9, maxDiff);
            if (maxDiff <= 0) break;
        }
#else
        for (int iter = 0; iter < DK_SWEEPS; iter++) {
            double maxSq = 0;
            dk_update_root_fast(coef, rre, rim, 0, maxSq);
            dk_update_root_fast(coef, rre, rim, 1, maxSq);
            dk_update_root_fast(coef, rre, rim, 2, maxSq);
            dk_update_root_fast(coef, rre, rim, 3, maxSq);
            dk_update_root_fast(coef, rre, rim, 4, maxSq);
            dk_update_root_fast(coef, rre, rim, 5, maxSq);
            dk_update_root_fast(coef, rre, rim, 6, maxSq);
            dk_update_root_fast(coef, rre, rim, 7, maxSq);
            dk_update_root_fast(coef, rre, rim, 8, maxSq);
            dk_update_root_fast(coef, rre, rim, 9, maxSq);
            if (maxSq <= 0) break;
        }
#endif
#pragma unroll
        for (int i = 0; i < 10; i++) { out_re[i] = rre[i]; out_im[i] = rim[i]; }
    } else {
        solve_poly_generic(coeffs_in, n, out_re, out_im);
    }
    for (int i = 0; i < n; i++)
        if (fabs(out_im[i]) < 1e-100) out_im[i] = 0;
    for (; n < 10; n++) { out_re[n] = out_re[n - 1]; out_im[n] = out_im[n - 1]; }
}

// ---- the same root finder spread over the 16 lanes of a DPP row, for calls with few hypotheses (one pair at a time):
// lane k < 10 of the row owns root k, and every lane keeps a copy of all ten.  A sweep is then
//   Horner numerators: ten lanes side by side;
//   for i = 0..9: lane i multiplies the factors with the OLD roots j > i into its denominator, divides, and its new root is
//     broadcast along the row (DPP row_newbcast: no LDS, no barrier); the lanes k > i multiply (p_k - new_i) into theirs.
// Every chain's operations and their order are those of the one-lane sweep (new roots j < k ascending, then old roots
// j > k ascending), so the roots are bit identical; about 1.1 k dependent instructions per sweep instead of 1.95 k.
// All 16 lanes of a row must call it together with the same coefficients (rows are independent of each other).
template <int I>
__device__ __forceinline__ double row_bcast(double v) {
    // row_newbcast:I; every lane reads an active lane of its own row, so bound_ctrl only saves the "old value" move
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x150 + I, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x150 + I, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ double row_dpp(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// msq / bad are this lane's own step (|q|^2, denominator unusable); they are reduced over the row once per sweep
template <int I>
__device__ __forceinline__ void dkw_step(const int k, Cx& p, const Cx& num, Cx& den, double (&cre)[10], double (&cim)[10], double& msq, int& bad) {
    Cx np_ = p;
    if (k == I) {
#pragma unroll
        for (int j = I + 1; j < 10; j++) den = cmul(den, csub(p, Cx{cre[j], cim[j]}));
        bad = ((den.re != 0 || den.im != 0) && den.re == den.re && den.im == den.im) ? 0 : 1;   // see dk_update_root_fast
        const Cx q = cdiv(num, den);
        np_ = Cx{p.re - q.re, p.im - q.im};
        msq = q.re * q.re + q.im * q.im;
        p = np_;
    }
    const double nre = row_bcast<I>(np_.re), nim = row_bcast<I>(np_.im);
    cre[I] = nre; cim[I] = nim;
    if (k > I) den = cmul(den, csub(p, Cx{nre, nim}));
}

__device__ void solve_poly10_wide(const double* coeffs_in, double* out_re, double* out_im, const int sub) {
    int n = 10;
    for (; n > 1; n--)
        if (fabs(coeffs_in[n]) > DBL_EPSILON) break;
    if (n != 10) { solve_poly10(coeffs_in, out_re, out_im); return; }   // lower degree: every lane for itself (identical results)
    const int k = sub < 10 ? sub : 9;              // lanes 10..15 of the row shadow lane 9
    double coef[11], cre[10], cim[10];
#pragma unroll
    for (int i = 0; i <= 10; i++) coef[i] = coeffs_in[i];
    Cx p{1, 0};
    {
        Cx q{1, 0};
        const Cx r{1, 1};
#pragma unroll
        for (int i = 0; i < 10; i++) {
            cre[i] = q.re; cim[i] = q.im;
            if (i == k) p = q;
            q = cmul(q, r);
        }
    }
    bool redo = false;
    for (int iter = 0; iter < DK_SWEEPS; iter++) {
        double msq = 0;
        int bad = 0;
        Cx num{coef[10], 0}, den{coef[10], 0};
#pragma unroll
        for (int j = 0; j < 10; j++) num = cadd(cmul(num, p), Cx{coef[9 - j], 0});
        dkw_step<0>(k, p, num, den, cre, cim, msq, bad);
        dkw_step<1>(k, p, num, den, cre, cim, msq, bad);
        dkw_step<2>(k, p, num, den, cre, cim, msq, bad);
        dkw_step<3>(k, p, num, den, cre, cim, msq, bad);
        dkw_step<4>(k, p, num, den, cre, cim, msq, bad);
        dkw_step<5>(k, p, num, den, cre, cim, msq, bad);
        dkw_step<6>(k, p, num, den, cre, cim, msq, bad);
        dkw_step<7>(k, p, num, den, cre, cim, msq, bad);
        dkw_step<8>(k, p, num, den, cre, cim, msq, bad);
        dkw_step<9>(k, p, num, den, cre, cim, msq, bad);
        // row-wide "largest step > 0" and "some denominator unusable".  The one-lane loop tests fmax over the roots of |q|^2
        // against 0, with fmax dropping NaNs: that is "some |q|^2 > 0".
        int any = (msq > 0 ? 1 : 0) | (bad << 1);
        any |= __builtin_amdgcn_mov_dpp(any, 0x121, 0xf, 0xf, true);   // row_ror:1
        any |= __builtin_amdgcn_mov_dpp(any, 0x122, 0xf, 0xf, true);   // row_ror:2
        any |= __builtin_amdgcn_mov_dpp(any, 0x124, 0xf, 0xf, true);   // row_ror:4
        any |= __builtin_amdgcn_mov_dpp(any, 0x128, 0xf, 0xf, true);   // row_ror:8
        if (__builtin_expect(any & 2, 0)) { redo = true; break; }
        if (!(any & 1)) break;
    }
    if (redo) {
        // a denominator came out zero or NaN somewhere (coincident iterates, over/underflow): every lane solves the polynomial
        // again by itself with the one-lane code, which handles those cases root by root
        solve_poly10(coeffs_in, out_re, out_im);
        return;
    }
#pragma unroll
    for (int i = 0; i < 10; i++) { out_re[i] = cre[i]; out_im[i] = cim[i]; }
    for (int i = 0; i < 10; i++)
        if (fabs(out_im[i]) < 1e-100) out_im[i] = 0;
}

// ---- row-wide helpers for the 16-lanes-per-hypothesis solver (all 16 lanes of a DPP row active)
template <int N> __device__ __forceinline__ int row_ror(int v) { return __builtin_amdgcn_mov_dpp(v, 0x120 + N, 0xf, 0xf, true); }
template <int N> __device__ __forceinline__ double row_ror(double v) { return row_dpp<0x120 + N>(v); }
// the smallest value of the row, in every lane
__device__ __forceinline__ int row_min(int v) {
    v = min(v, row_ror<1>(v)); v = min(v, row_ror<2>(v)); v = min(v, row_ror<4>(v)); v = min(v, row_ror<8>(v));
    return v;
}
__device__ __forceinline__ int row_or(int v) {
    v |= row_ror<1>(v); v |= row_ror<2>(v); v |= row_ror<4>(v); v |= row_ror<8>(v);
    return v;
}
// v of lane `src` (a lane index in the wavefront, the same in every lane of the row)
__device__ __forceinline__ double lane_read(int src, double v) {
    const int lo = __builtin_amdgcn_ds_bpermute(src << 2, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(src << 2, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

// EMEstimatorCallback::runKernel.  x1,x2: 5 normalised correspondences.  Writes up to 10
// models (row-major, unit Frobenius norm, x2^T E x1 = 0) to Ebuf; returns their number.
// WIDE = false: one lane per call (sub is ignored).  WIDE = true: the 16 lanes of a DPP row call it together with the same
// correspondences, sub = 0..15 the lane's place in the row: the null space and E E^T are computed by every lane for itself
// (identical results), and then lane k owns constraint k -- it expands that equation, holds that matrix row through the
// elimination (pivot search and pivot row travel along the row by DPP / ds_bpermute), owns root k of the tenth-degree polynomial
// and back-substitutes it.  Every number is produced by the same operations in the same order as in the one-lane form.
template <bool WIDE>
__device__ int five_point(const double* x1, const double* x2, double* Ebuf, const int sub = -1) {
    FP_TICK(0);
    double At[9 * 9];
    for (int i = 0; i < 81; i++) At[i] = 0;
    for (int i = 0; i < 5; i++) {
        const double u1 = x1[2 * i], v1 = x1[2 * i + 1], u2 = x2[2 * i], v2 = x2[2 * i + 1];
        double* q = At + i * 9;
        q[0] = u2 * u1; q[1] = u2 * v1; q[2] = u2;
        q[3] = v2 * u1; q[4] = v2 * v1; q[5] = v2;
        q[6] = u1;      q[7] = v1;      q[8] = 1.0;
    }
    double W[5], V[25];
    jacobi_svd<9, 5, 9>(At, W, V);
    const double* EE0 = At + 5 * 9;  // 4 null-space vectors, consecutive rows
    FP_TICK(1);

    Poly<1> Ep[3][3];
    sfor<9>([&](auto I) { Ep[I / 3][I % 3] = plin(EE0[0 * 9 + I], EE0[1 * 9 + I], EE0[2 * 9 + I], EE0[3 * 9 + I]); });
    double A[10][20];          // one-lane form: the 10 x 20 constraint matrix
    double Afin[6][10];        // 16-lane form: rows 4..9, columns 10..19 after the elimination (all that is read afterwards)
    if (!WIDE) {
        Poly<2> EEt[3][3], tr;
        sfor<9>([&](auto I) { poly_eet(EEt[I / 3][I % 3], Ep[I / 3], Ep[I % 3]); });
        padd(tr, EEt[0][0], EEt[1][1], 1.0);
        padd(tr, tr, EEt[2][2], 1.0);
        {
            Poly<3> d;
            poly_det(d, Ep);
            poly_to_row(d, A[0]);
        }
        sfor<9>([&](auto I) {
            constexpr int r = I / 3, c = I % 3;
            const Poly<1> Ecol[3] = {Ep[0][c], Ep[1][c], Ep[2][c]};
            Poly<3> q;
            poly_trace_eq(q, EEt[r], Ecol, tr, Ep[r][c]);
            poly_to_row(q, A[1 + I]);
        });
        FP_TICK(2);
        // Gauss-Jordan with partial pivoting.  A lives in private memory, where a dependent access costs a round trip of several
        // hundred cycles, and the loop as OpenCV writes it is one such round trip per element (300 us of the solver's 420 us outside the
        // root finder).  Here every loop bound is a compile-time constant after unrolling (only the pivot row's index is data) and a
        // column step is five batches of independent loads: the column, the pivot row and row `col`, then the other nine rows three at
        // a time.  The arithmetic per element is unchanged; "skip the row if f == 0" is a select (f * pr could be NaN or change a
        // zero's sign); columns left of `col` are not touched (they are never read again).
    #pragma unroll
        for (int col = 0; col < 10; col++) {
            double cv[10];
    #pragma unroll
            for (int r = 0; r < 10; r++) cv[r] = A[r][col];
            int piv = col;
            double best = fabs(cv[col]);
    #pragma unroll
            for (int r = col + 1; r < 10; r++) {
                const double v = fabs(cv[r]);
                if (v > best) { best = v; piv = r; }
            }
            if (best < DBL_MIN) return 0;
            double pr[20], oc[20];
    #pragma unroll
            for (int m = col; m < 20; m++) { pr[m] = A[piv][m]; oc[m] = A[col][m]; }
    #pragma unroll
            for (int m = col; m < 20; m++) A[piv][m] = oc[m];          // the swap (a plain rewrite when piv == col)
            const double inv = 1.0 / pr[col];
    #pragma unroll
            for (int m = col; m < 20; m++) { pr[m] *= inv; A[col][m] = pr[m]; }
    #pragma unroll
            for (int r0 = 0; r0 < 10; r0 += 3) {
                double row[3][20];
    #pragma unroll
                for (int q = 0; q < 3; q++) {
                    const int r = r0 + q;
                    if (r < 10 && r != col) {
    #pragma unroll
                        for (int m = col; m < 20; m++) row[q][m] = A[r][m];
                    }
                }
    #pragma unroll
                for (int q = 0; q < 3; q++) {
                    const int r = r0 + q;
                    if (r < 10 && r != col) {
                        const double f = (r > col && r == piv) ? cv[col] : cv[r];   // row piv now holds what row col held
                        const bool skip = f == 0;
    #pragma unroll
                        for (int m = col; m < 20; m++) {
                            const double t = row[q][m] - f * pr[m];
                            A[r][m] = skip ? row[q][m] : t;
                        }
                    }
                }
            }
        }
    } else {
        const int k = sub < 10 ? sub : 9;          // lanes 10..15 shadow lane 9
        double row[20];
        {
            // lane k > 0 expands constraint (r, c) = ((k - 1) / 3, (k - 1) % 3): it needs row r of E E^T (made here from row r of
            // E, picked by selects: no lane-dependent indexing) and the diagonal of E E^T for the trace; lane 0 expands det(E)
            const int kk = k > 0 ? k - 1 : 0;
            const int r = kk / 3, c = kk - 3 * r;
            auto pick = [&](const Poly<1>& p0, const Poly<1>& p1, const Poly<1>& p2, int w) {
                Poly<1> o;
                sfor<4>([&](auto I) { o.c[I] = w == 0 ? p0.c[I] : (w == 1 ? p1.c[I] : p2.c[I]); });
                return o;
            };
            const Poly<1> Er[3] = {pick(Ep[0][0], Ep[1][0], Ep[2][0], r), pick(Ep[0][1], Ep[1][1], Ep[2][1], r), pick(Ep[0][2], Ep[1][2], Ep[2][2], r)};
            const Poly<1> Ecol[3] = {pick(Ep[0][0], Ep[0][1], Ep[0][2], c), pick(Ep[1][0], Ep[1][1], Ep[1][2], c), pick(Ep[2][0], Ep[2][1], Ep[2][2], c)};
            Poly<2> EEr[3], d0, d1, d2, tr;
            sfor<3>([&](auto K) { poly_eet(EEr[K], Er, Ep[K]); });
            poly_eet(d0, Ep[0], Ep[0]); poly_eet(d1, Ep[1], Ep[1]); poly_eet(d2, Ep[2], Ep[2]);
            padd(tr, d0, d1, 1.0);
            padd(tr, tr, d2, 1.0);
            Poly<3> q;
            poly_trace_eq(q, EEr, Ecol, tr, pick(Er[0], Er[1], Er[2], c));
            if (k == 0) poly_det(q, Ep);
            poly_to_row(q, row);
        }
        FP_TICK(2);
        // Gauss-Jordan with the matrix row `lid` in this lane (a row swap exchanges the two lanes' lids, not their data)
        int lid = k;
        const int me = (int)__lane_id();
#pragma unroll
        for (int col = 0; col < 10; col++) {
            // pivot = the first row >= col with the largest |A[r][col]|, as the serial scan finds it (it never moves to a NaN,
            // and never away from a NaN in row col)
            const double a = fabs(row[col]);
            double best = (lid >= col && a == a) ? a : -1.0;
            int piv = lid;
#define VO_ARGMAX_STEP(N)                                                                         \
            {                                                                                     \
                const double ov = row_ror<N>(best);                                               \
                const int oi = row_ror<N>(piv);                                                   \
                const bool take = ov > best || (ov == best && oi < piv);                          \
                best = take ? ov : best;                                                          \
                piv = take ? oi : piv;                                                            \
            }
            VO_ARGMAX_STEP(1) VO_ARGMAX_STEP(2) VO_ARGMAX_STEP(4) VO_ARGMAX_STEP(8)
#undef VO_ARGMAX_STEP
            const bool nan_at_col = row_or((lid == col && a != a) ? 1 : 0) != 0;
            if (nan_at_col) piv = col;
            else if (best < DBL_MIN) return 0;
            const int src = row_min(lid == piv ? me : 127);
            double pr[20];
#pragma unroll
            for (int m = col; m < 20; m++) pr[m] = lane_read(src, row[m]);
            if (lid == col) lid = piv;
            else if (lid == piv) lid = col;
            const double inv = 1.0 / pr[col];
#pragma unroll
            for (int m = col; m < 20; m++) pr[m] *= inv;
            const double f = row[col];
            const bool is_pivot = lid == col, skip = f == 0;
#pragma unroll
            for (int m = col; m < 20; m++) {
                const double t = row[m] - f * pr[m];
                row[m] = is_pivot ? pr[m] : (skip ? row[m] : t);
            }
        }
#pragma unroll
        for (int e = 4; e < 10; e++) {
            const int src = row_min(lid == e ? me : 127);
#pragma unroll
            for (int j = 0; j < 10; j++) Afin[e - 4][j] = lane_read(src, row[10 + j]);
        }
    }
    FP_TICK(3);
    double b[3][13];
    for (int i = 0; i < 3; i++) {
        const double* a1 = WIDE ? &Afin[i * 2][0] : &A[i * 2 + 4][10];
        const double* a2 = WIDE ? &Afin[i * 2 + 1][0] : &A[i * 2 + 5][10];
        double row1[13], row2[13];
        for (int k = 0; k < 13; k++) { row1[k] = 0; row2[k] = 0; }
        for (int k = 0; k < 3; k++) { row1[1 + k] = a1[k]; row1[5 + k] = a1[3 + k]; }
        for (int k = 0; k < 4; k++) row1[9 + k] = a1[6 + k];
        for (int k = 0; k < 3; k++) { row2[k] = a2[k]; row2[4 + k] = a2[3 + k]; }
        for (int k = 0; k < 4; k++) row2[8 + k] = a2[6 + k];
        for (int k = 0; k < 13; k++) b[i][k] = row1[k] - row2[k];
    }
    double P[3][3][5];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            for (int k = 0; k < 5; k++) P[r][c][k] = 0;
            if (c < 2) for (int k = 0; k < 4; k++) P[r][c][3 - k] = b[r][c * 4 + k];
            else for (int k = 0; k < 5; k++) P[r][c][4 - k] = b[r][8 + k];
        }
    double cdet[11];
    for (int k = 0; k <= 10; k++) cdet[k] = 0;
    for (int r = 0; r < 3; r++) {
        const int r0 = (r == 0) ? 1 : 0, r1 = (r == 2) ? 1 : 2;
        double m_a[7], m_b[7], minor[7], term[11];
        for (int k = 0; k < 7; k++) { m_a[k] = 0; m_b[k] = 0; }
        for (int i = 0; i <= 3; i++)
            for (int j = 0; j <= 3; j++) { m_a[i + j] += P[r0][0][i] * P[r1][1][j]; }
        for (int i = 0; i <= 3; i++)
            for (int j = 0; j <= 3; j++) { m_b[i + j] += P[r0][1][i] * P[r1][0][j]; }
        for (int k = 0; k < 7; k++) minor[k] = m_a[k] - m_b[k];
        for (int k = 0; k <= 10; k++) term[k] = 0;
        for (int i = 0; i <= 4; i++)
            for (int j = 0; j <= 6; j++) term[i + j] += P[r][2][i] * minor[j];
        const double sgn = (r == 1) ? -1.0 : 1.0;
        for (int k = 0; k <= 10; k++) cdet[k] += sgn * term[k];
    }
    FP_TICK(4);
    double rre[10], rim[10];
    if (!WIDE) solve_poly10(cdet, rre, rim);
    else solve_poly10_wide(cdet, rre, rim, sub);
    FP_TICK(5);
    // the model of one root: back-substitution through the 3x3 null space; false if the root is complex or the null vector has no
    // finite dehomogenisation
    auto model_of_root = [&](const double zre, const double zim, double* E) -> bool {
        if (fabs(zim) > 1e-10) return false;
        const double z1 = zre, z2 = z1 * z1, z3 = z2 * z1, z4 = z3 * z1;
        double bz[9];
        for (int j = 0; j < 3; j++) {
            const double* br = b[j];
            bz[j * 3 + 0] = br[0] * z3 + br[1] * z2 + br[2] * z1 + br[3];
            bz[j * 3 + 1] = br[4] * z3 + br[5] * z2 + br[6] * z1 + br[7];
            bz[j * 3 + 2] = br[8] * z4 + br[9] * z3 + br[10] * z2 + br[11] * z1 + br[12];
        }
        double w3[3], U3[9], Vt3[9];
        svd_square<3>(bz, w3, U3, Vt3);
        const double* xy1 = Vt3 + 6;
        if (fabs(xy1[2]) < 1e-10) return false;
        const double xs = xy1[0] / xy1[2], ys = xy1[1] / xy1[2], zs = z1;
        double nrm = 0;
        for (int k = 0; k < 9; k++) {
            E[k] = EE0[0 * 9 + k] * xs + EE0[1 * 9 + k] * ys + EE0[2 * 9 + k] * zs + EE0[3 * 9 + k];
            nrm += E[k] * E[k];
        }
        nrm = sqrt(nrm);
        for (int k = 0; k < 9; k++) E[k] /= nrm;
        return true;
    };
    // Ebuf may be global memory (the solve kernels hand in the hypothesis' slot of the model table): a model is finished in
    // registers and stored once
    int count = 0;
    if (!WIDE) {
#pragma unroll
        for (int i = 0; i < 10; i++) {
            double Em[9];
            if (model_of_root(rre[i], rim[i], Em)) {
#pragma unroll
                for (int j = 0; j < 9; j++) Ebuf[count * 9 + j] = Em[j];
                count++;
            }
        }
    } else {
        // 16 lanes: lane k < 10 takes root k, then the models are collected in root order along the row (lanes 10..15 shadow lane 9)
        const int k = sub < 10 ? sub : 9;
        double Em[9];
        for (int j = 0; j < 9; j++) Em[j] = 0;
        const int mine = model_of_root(rre[k], rim[k], Em) ? 1 : 0;
#define FP_COLLECT(I)                                                                                       \
        if (__builtin_amdgcn_mov_dpp(mine, 0x150 + I, 0xf, 0xf, true)) {                                    \
            for (int j = 0; j < 9; j++) {                                                                   \
                const double ev = row_bcast<I>(Em[j]);                                                      \
                if (sub == 0) Ebuf[count * 9 + j] = ev;                                                     \
            }                                                                                               \
            count++;                                                                                        \
        }
        FP_COLLECT(0) FP_COLLECT(1) FP_COLLECT(2) FP_COLLECT(3) FP_COLLECT(4)
        FP_COLLECT(5) FP_COLLECT(6) FP_COLLECT(7) FP_COLLECT(8) FP_COLLECT(9)
#undef FP_COLLECT
    }
    FP_TICK(6);
    return count;
}

// (col - c)/f as the MatExpr col*(1/f) + (-c*(1/f))
struct NormK { double ax, bx, ay, by; };
__device__ __forceinline__ void norm_pt(const NormK& nk, float2 q, double& x, double& y) {
    x = (double)q.x * nk.ax + nk.bx;
    y = (double)q.y * nk.ay + nk.by;
}

__device__ __forceinline__ float sampson_err(const double* E, double u1, double v1, double u2, double v2) {
    const double e0 = E[0] * u1 + E[1] * v1 + E[2], e1 = E[3] * u1 + E[4] * v1 + E[5], e2 = E[6] * u1 + E[7] * v1 + E[8];
    const double f0 = E[0] * u2 + E[3] * v2 + E[6], f1 = E[1] * u2 + E[4] * v2 + E[7];
    const double x2tEx1 = u2 * e0 + v2 * e1 + e2;
    const double a = e0 * e0, b = e1 * e1, c = f0 * f0, d = f1 * f1;
    return (float)(x2tEx1 * x2tEx1 / (a + b + c + d));
}

__device__ int ransac_update_num_iters(double p, double ep, int modelPoints, int maxIters) {
    p = fmax(p, 0.); p = fmin(p, 1.);
    ep = fmax(ep, 0.); ep = fmin(ep, 1.);
    double num = fmax(1. - p, DBL_MIN);
    double denom = 1. - pow(1. - ep, (double)modelPoints);
    if (denom < DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : (int)rint(num / denom);
}

// ------------------------------------------------------------------ RANSAC kernels
__global__ void k_ransac_init(PoseWork* pw, const int* __restrict__ nmatch, const int* __restrict__ pair_status, int P, int max_iters) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    PoseWork& w = pw[p];
    w.rng = 0xFFFFFFFFFFFFFFFFull;
    w.niters = max_iters > 1 ? max_iters : 1;
    w.iter = 0;
    w.max_good = 0;
    w.n_good = 0;
    const int n = nmatch[p];
    w.status = pair_status[p];
    w.done = 0;
    if (w.status != VO_OK) w.done = 1;
    else if (n < 5) { w.status = VO_ERR_TOO_FEW; w.done = 1; }
    for (int i = 0; i < 9; i++) w.E[i] = 0;
}

// draw the 5-tuples of iterations [iter, iter+nh) of every active pair (nh <= VO_HYP_ROUND)
__global__ void k_ransac_sample(PoseWork* pw, const int* __restrict__ nmatch, int P, int nh, int* __restrict__ samples) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    PoseWork& w = pw[p];
    if (w.done) return;
    const int n = nmatch[p];
    DevRNG rng{w.rng};
    int* sp = samples + (long)p * VO_HYP_ROUND * 5;
    for (int h = 0; h < nh; h++) {
        int idx[5];
        if (n == 5) {
            for (int i = 0; i < 5; i++) idx[i] = i;
        } else {
            for (int i = 0; i < 5; i++) {
                int v;
                bool dup;
                do {
                    v = rng.uniform(0, n);
                    dup = false;
                    for (int j = 0; j < i; j++) dup |= (idx[j] == v);
                } while (dup);
                idx[i] = v;
            }
        }
        for (int i = 0; i < 5; i++) sp[h * 5 + i] = idx[i];
    }
    w.rng = rng.state;
}

__global__ __launch_bounds__(DK_T) void k_ransac_solve(const PoseWork* __restrict__ pw, const int* __restrict__ nmatch, int P, int nh, int kp_cap,
                                                       const float2* __restrict__ q1, const float2* __restrict__ q2, NormK nk,
                                                       const int* __restrict__ samples, double* __restrict__ models, int* __restrict__ nmodels) {
    // nh (a power of two <= DK_T) hypotheses per pair: a wavefront serves DK_T / nh pairs
    const int g = blockIdx.x * DK_T + threadIdx.x;
    const int p = g / nh, h = g % nh;
    if (p >= P) return;
    const PoseWork& w = pw[p];
    if (w.done) return;
    const long ho = (long)p * VO_HYP_ROUND + h;
    if (w.iter + h >= w.niters) { nmodels[ho] = 0; return; }   // beyond the current budget: never consumed
    const int* sp = samples + ho * 5;
    double x1[10], x2[10];
    for (int i = 0; i < 5; i++) {
        const long o = (long)p * kp_cap + sp[i];
        norm_pt(nk, q1[o], x1[2 * i], x1[2 * i + 1]);
        norm_pt(nk, q2[o], x2[2 * i], x2[2 * i + 1]);
    }
    nmodels[ho] = five_point<false>(x1, x2, models + ho * 90);
}

// The same with 16 lanes (one DPP row) per hypothesis, for rounds with few hypotheses in total: the serial chain of a
// hypothesis is shorter (solve_poly10_wide), at 16 times the lanes.  Bit-identical models.
__global__ __launch_bounds__(DK_T) void k_ransac_solve_wide(const PoseWork* __restrict__ pw, const int* __restrict__ nmatch, int P, int nh, int kp_cap,
                                                            const float2* __restrict__ q1, const float2* __restrict__ q2, NormK nk,
                                                            const int* __restrict__ samples, double* __restrict__ models, int* __restrict__ nmodels) {
    const int g = (blockIdx.x * DK_T + threadIdx.x) >> 4, sub = threadIdx.x & 15;
    const int p = g / nh, h = g % nh;
    if (p >= P) return;                                        // row-uniform exits
    const PoseWork& w = pw[p];
    if (w.done) return;
    const long ho = (long)p * VO_HYP_ROUND + h;
    if (w.iter + h >= w.niters) { if (sub == 0) nmodels[ho] = 0; return; }
    const int* sp = samples + ho * 5;
    double x1[10], x2[10];
    for (int i = 0; i < 5; i++) {
        const long o = (long)p * kp_cap + sp[i];
        norm_pt(nk, q1[o], x1[2 * i], x1[2 * i + 1]);
        norm_pt(nk, q2[o], x2[2 * i], x2[2 * i + 1]);
    }
    const int nm = five_point<true>(x1, x2, models + ho * 90, sub);
    if (sub == 0) nmodels[ho] = nm;
}

// one wavefront per (pair, hypothesis): inlier count of each of its models
__global__ __launch_bounds__(256) void k_ransac_score(const PoseWork* __restrict__ pw, const int* __restrict__ nmatch, int P, int nh, int kp_cap,
                                                      const float2* __restrict__ q1, const float2* __restrict__ q2, NormK nk, float thr2,
                                                      const double* __restrict__ models, const int* __restrict__ nmodels, int* __restrict__ counts) {
    const int p = blockIdx.y, h = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (pw[p].done || h >= nh) return;
    const long ho = (long)p * VO_HYP_ROUND + h;
    const int nm = nmodels[ho];
    if (nm <= 0) return;
    const int n = nmatch[p];
    int cnt[10];
    for (int m = 0; m < 10; m++) cnt[m] = 0;
    const double* mo = models + ho * 90;
    for (int i = lane; i < n; i += 64) {
        const long o = (long)p * kp_cap + i;
        double u1, v1, u2, v2;
        norm_pt(nk, q1[o], u1, v1);
        norm_pt(nk, q2[o], u2, v2);
        for (int m = 0; m < nm; m++) cnt[m] += sampson_err(mo + m * 9, u1, v1, u2, v2) <= thr2 ? 1 : 0;
    }
    for (int m = 0; m < nm; m++) {
        int c = cnt[m];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
        if (lane == 0) counts[ho * 10 + m] = c;
    }
}

// sequential semantics of RANSACPointSetRegistrator::run over this round's hypotheses
__global__ void k_ransac_scan(PoseWork* pw, const int* __restrict__ nmatch, int P, int nh, double prob, const double* __restrict__ models,
                              const int* __restrict__ nmodels, const int* __restrict__ counts, int* __restrict__ active) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    PoseWork& w = pw[p];
    if (w.done) return;
    const int n = nmatch[p];
    int niters = w.niters, maxGood = w.max_good, iter = w.iter;
    if (n == 5) {
        // count == modelPoints: the kernel runs once, all points are inliers
        const long ho = (long)p * VO_HYP_ROUND;
        if (nmodels[ho] <= 0) { w.status = VO_ERR_NO_MODEL; }
        else { for (int i = 0; i < 9; i++) w.E[i] = models[ho * 90 + i]; w.max_good = 5; }
        w.done = 1;
        return;
    }
    int h = 0;
    for (; h < nh && iter < niters; h++, iter++) {
        const long ho = (long)p * VO_HYP_ROUND + h;
        const int nm = nmodels[ho];
        for (int m = 0; m < nm; m++) {
            const int good = counts[ho * 10 + m];
            if (good > max(maxGood, 4)) {
                for (int i = 0; i < 9; i++) w.E[i] = models[ho * 90 + m * 9 + i];
                maxGood = good;
                niters = ransac_update_num_iters(prob, (double)(n - good) / n, 5, niters);
            }
        }
    }
    w.niters = niters; w.max_good = maxGood; w.iter = iter;
    if (!(iter < niters)) {
        w.done = 1;
        if (maxGood <= 0) w.status = VO_ERR_NO_MODEL;
    } else {
        if (active) atomicAdd(active, 1);
    }
}

// inlier mask of the winning model (same arithmetic as the scoring pass)
__global__ __launch_bounds__(256) void k_ransac_mask(const PoseWork* __restrict__ pw, const int* __restrict__ nmatch, int kp_cap,
                                                     const float2* __restrict__ q1, const float2* __restrict__ q2, NormK nk, float thr2,
                                                     uint8_t* __restrict__ mask) {
    const int p = blockIdx.y;
    const int n = nmatch[p];
    const PoseWork& w = pw[p];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long o = (long)p * kp_cap + i;
    uint8_t f = 0;
    if (w.status == VO_OK) {
        if (n == 5) f = 1;
        else {
            double u1, v1, u2, v2;
            norm_pt(nk, q1[o], u1, v1);
            norm_pt(nk, q2[o], u2, v2);
            f = sampson_err(w.E, u1, v1, u2, v2) <= thr2 ? 1 : 0;
        }
    }
    mask[o] = f;
}

// ------------------------------------------------------------------ recoverPose
__device__ double det3(const double* M) {
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}
__device__ void mat3mul(const double* A, const double* B, double* C) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += A[i * 3 + k] * B[k * 3 + j];
            C[i * 3 + j] = s;
        }
}

// decomposeEssentialMat per pair
__global__ void k_decompose(PoseWork* pw, int P, int* __restrict__ vote) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    PoseWork& w = pw[p];
    for (int c = 0; c < 4; c++) vote[p * 4 + c] = 0;
    if (w.status != VO_OK) return;
    double D[3], U[9], Vt[9], E[9];
    for (int i = 0; i < 9; i++) E[i] = w.E[i];
    svd_square<3>(E, D, U, Vt);
    if (det3(U) < 0) for (int i = 0; i < 9; i++) U[i] *= -1.;
    if (det3(Vt) < 0) for (int i = 0; i < 9; i++) Vt[i] *= -1.;
    const double Wm[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1}, Wt[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1};
    double tmp[9], R1[9], R2[9];
    mat3mul(U, Wm, tmp); mat3mul(tmp, Vt, R1);
    mat3mul(U, Wt, tmp); mat3mul(tmp, Vt, R2);
    for (int i = 0; i < 9; i++) { w.R1[i] = R1[i]; w.R2[i] = R2[i]; }
    for (int i = 0; i < 3; i++) w.tv[i] = U[i * 3 + 2];
}

__device__ void triangulate_one(const double* P1, const double* P2, double x1, double y1, double x2, double y2, double X[4]) {
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

// cheirality vote: one lane per (pair, match, rotation candidate); the lane votes for (R, +t) AND for (R, -t).
// recoverPose triangulates every match with each of the four candidates [R1|t], [R2|t], [R1|-t], [R2|-t].  Negating t negates
// the fourth COLUMN of the DLT matrix A (rows 0, 1 hold a zero there), i.e. A' = A diag(1,1,1,-1), and the one-sided Jacobi
// SVD is sign-symmetric step by step: with rows At'_i = e_i At_i (e_i = +-1) the dot product p, hence s (beta >= 0) or c
// (beta < 0), only changes sign, the rotated rows come out as +- the unprimed ones (in the beta < 0 branch the two signs swap
// rows), the row norms W and every convergence test are the same, and IEEE negation is exact.  So the last right singular
// vector is X' = +-(X0, X1, X2, -X3) bit for bit, and the primed tests follow from the unprimed quantities:
// X2' X3' = -(X2 X3), Q' = -Q (Q3' = Q3), z' = -z.  Half of the 4 x 4 SVDs (round 3: 13.8 M -> 6.9 M instructions per 32 pairs);
// tests/test_gpu_pose.py compares all four counts with a CPU restatement that triangulates the four candidates separately.
__global__ __launch_bounds__(64) void k_cheirality(const PoseWork* __restrict__ pw, const int* __restrict__ nmatch, int kp_cap,
                                                    const float2* __restrict__ q1, const float2* __restrict__ q2, NormK nk, double dist_thresh,
                                                    int* __restrict__ vote) {
    const int p = blockIdx.y, c = blockIdx.z;   // c = 0: R1, 1: R2
    const PoseWork& w = pw[p];
    if (w.status != VO_OK) return;
    const int n = nmatch[p];
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (blockIdx.x * 64 >= n) return;
    int m = 0, m2 = 0;
    if (i < n) {
        const double* Rc = (c & 1) ? w.R2 : w.R1;
        double Pm[12];
        for (int r = 0; r < 3; r++) {
            for (int j = 0; j < 3; j++) Pm[r * 4 + j] = Rc[r * 3 + j];
            Pm[r * 4 + 3] = w.tv[r];
        }
        const double P0[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
        const long o = (long)p * kp_cap + i;
        double u1, v1, u2, v2, Q[4];
        norm_pt(nk, q1[o], u1, v1);
        norm_pt(nk, q2[o], u2, v2);
        triangulate_one(P0, Pm, u1, v1, u2, v2, Q);
        const double x23 = Q[2] * Q[3];
        Q[0] /= Q[3]; Q[1] /= Q[3]; Q[2] /= Q[3]; Q[3] /= Q[3];
        const double z = Pm[8] * Q[0] + Pm[9] * Q[1] + Pm[10] * Q[2] + Pm[11] * Q[3];
        m = (x23 > 0) && (Q[2] < dist_thresh) && (z > 0) && (z < dist_thresh);
        m2 = (-x23 > 0) && (-Q[2] < dist_thresh) && (-z > 0) && (-z < dist_thresh);
    }
    const unsigned long long b = __ballot(m), b2 = __ballot(m2);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(&vote[p * 4 + c], (int)__popcll(b));
    if ((threadIdx.x & 63) == 0 && b2) atomicAdd(&vote[p * 4 + c + 2], (int)__popcll(b2));
}

__global__ void k_pose_pick(PoseWork* pw, int P, const int* __restrict__ vote) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    PoseWork& w = pw[p];
    if (w.status != VO_OK) return;
    const int g0 = vote[p * 4], g1 = vote[p * 4 + 1], g2 = vote[p * 4 + 2], g3 = vote[p * 4 + 3];
    int sel;
    if (g0 >= g1 && g0 >= g2 && g0 >= g3) sel = 0;
    else if (g1 >= g0 && g1 >= g2 && g1 >= g3) sel = 1;
    else if (g2 >= g0 && g2 >= g1 && g2 >= g3) sel = 2;
    else sel = 3;
    const double* Rs = (sel & 1) ? w.R2 : w.R1;
    for (int i = 0; i < 9; i++) w.R[i] = Rs[i];
    for (int i = 0; i < 3; i++) w.t[i] = (sel >= 2 ? -1.0 : 1.0) * w.tv[i];
    w.n_good = sel == 0 ? g0 : sel == 1 ? g1 : sel == 2 ? g2 : g3;
}

// ------------------------------------------------------------------ triangulatePoints (float32 in/out)
__global__ __launch_bounds__(256) void k_triangulate(const double* __restrict__ Pm, const float* __restrict__ x1, const float* __restrict__ x2,
                                                     int n, float* __restrict__ X4) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double P1[12], P2[12], X[4];
    for (int k = 0; k < 12; k++) { P1[k] = Pm[k]; P2[k] = Pm[12 + k]; }
    triangulate_one(P1, P2, (double)x1[2 * i], (double)x1[2 * i + 1], (double)x2[2 * i], (double)x2[2 * i + 1], X);
    for (int k = 0; k < 4; k++) X4[(long)k * n + i] = (float)X[k];
}

// All ratio-test matches of a wave of pairs in ONE launch (the loop of vo.py:91-93 calls triangulate twice per match):
// pair p uses its own projection matrices Pm[p] = (P1, P2) = (K @ prev_pose[:3], K @ cur_pose[:3]); the result is
// dehomogenised as the reference does it, in float32 (points_4D / points_4D[3], libs/vo_utils.py:161), and written
// compacted at offsets[p] + i together with the matched pixel positions.
__global__ __launch_bounds__(256) void k_triangulate_pairs(const double* __restrict__ Pm, const long* __restrict__ offsets, int kp_cap,
                                                           const float2* __restrict__ q1, const float2* __restrict__ q2,
                                                           float* __restrict__ X3, float2* __restrict__ q1o, float2* __restrict__ q2o) {
    const int p = blockIdx.y;
    const long o0 = offsets[p];
    const int n = (int)(offsets[p + 1] - o0);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double P1[12], P2[12], X[4];
    for (int k = 0; k < 12; k++) { P1[k] = Pm[p * 24 + k]; P2[k] = Pm[p * 24 + 12 + k]; }
    const float2 a = q1[(long)p * kp_cap + i], b = q2[(long)p * kp_cap + i];
    triangulate_one(P1, P2, (double)a.x, (double)a.y, (double)b.x, (double)b.y, X);
    const float w = (float)X[3];
    X3[(o0 + i) * 3 + 0] = (float)X[0] / w;
    X3[(o0 + i) * 3 + 1] = (float)X[1] / w;
    X3[(o0 + i) * 3 + 2] = (float)X[2] / w;
    if (q1o) q1o[o0 + i] = a;
    if (q2o) q2o[o0 + i] = b;
}

int vo_launch_triangulate_pairs(vo_ctx* c, int P, const double* dPm, const long* doffsets, float* dX3, float2* dq1, float2* dq2, hipStream_t st) {
    hipLaunchKernelGGL(k_triangulate_pairs, dim3((c->kp_cap + 255) / 256, P), dim3(256), 0, st, dPm, doffsets, c->kp_cap, (const float2*)c->q1(),
                       (const float2*)c->q2(), dX3, dq1, dq2);
    HIP_TRY(c, hipGetLastError());
    return VO_OK;
}

// ------------------------------------------------------------------ host
static NormK make_normk(const double K[9]) {
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    NormK nk;
    nk.ax = 1. / fx; nk.bx = -cx * (1. / fx);
    nk.ay = 1. / fy; nk.by = -cy * (1. / fy);
    return nk;
}

int vo_launch_ransac(vo_ctx* c, int P, const double K[9], double prob, double thr, int max_iters, hipStream_t st, bool host_free) {
    if (P < 1 || P > c->max_pairs) return vo_fail(c, VO_ERR_INVALID, "pair count exceeds max_pairs");
    const int kp_cap = c->kp_cap;
    const NormK nk = make_normk(K);
    const double threshold = thr / ((K[0] + K[4]) / 2);
    const float thr2 = (float)(threshold * threshold);
    PoseWork* pw = (PoseWork*)c->d_pose.p;
    const int* nmatch = (const int*)c->nmatch();
    const float2* q1 = (const float2*)c->q1();
    const float2* q2 = (const float2*)c->q2();
    int* samples = (int*)c->d_samples.p;
    double* models = (double*)c->d_hyp.p;
    int* nmodels = (int*)c->d_nmodels.p;
    int* counts = (int*)c->d_counts.p;
    int* active = (int*)c->d_misc.p;
    const int pb = (P + 63) / 64;
    hipLaunchKernelGGL(k_ransac_init, dim3(pb), dim3(64), 0, st, pw, nmatch, (const int*)c->pair_status(), P, max_iters);
    // The first round is small: with the usual inlier ratios the adaptive iteration count drops below it
    // after the first good model, and a small round packs several pairs into one solver wavefront.
    const int first_round = 16;   // a power of two <= VO_HYP_ROUND (k_ransac_solve packs DK_T / nh pairs per wavefront)
    int iters_done = 0;
    for (int round = 0; iters_done < std::max(max_iters, 1); round++) {
        const int nh = round == 0 ? first_round : VO_HYP_ROUND;
        if (!host_free) HIP_TRY(c, hipMemsetAsync(active, 0, sizeof(int), st));
        hipLaunchKernelGGL(k_ransac_sample, dim3(pb), dim3(64), 0, st, pw, nmatch, P, nh, samples);
        // few hypotheses in the round (single pairs, small batches): a DPP row per hypothesis, one wavefront per SIMD at most
        // (the asynchronous entry points overlap the pose stage with the next submission's detect kernels, where the 16-lane form's
        // ten-fold VALU work would cost throughput: they only take it for rounds of <= 512 hypotheses)
        if ((long)P * nh <= (host_free ? VO_WIDE_SOLVE_ASYNC_MAX_HYP : VO_WIDE_SOLVE_MAX_HYP))
            hipLaunchKernelGGL(k_ransac_solve_wide, dim3((P * nh * 16 + DK_T - 1) / DK_T), dim3(DK_T), 0, st, pw, nmatch, P, nh, kp_cap, q1, q2, nk, samples, models, nmodels);
        else
            hipLaunchKernelGGL(k_ransac_solve, dim3((P * nh + DK_T - 1) / DK_T), dim3(DK_T), 0, st, pw, nmatch, P, nh, kp_cap, q1, q2, nk, samples, models, nmodels);
        hipLaunchKernelGGL(k_ransac_score, dim3((nh + 3) / 4, P), dim3(256), 0, st, pw, nmatch, P, nh, kp_cap, q1, q2, nk, thr2, models, nmodels, counts);
        hipLaunchKernelGGL(k_ransac_scan, dim3(pb), dim3(64), 0, st, pw, nmatch, P, nh, prob, models, nmodels, counts, host_free ? nullptr : active);
        iters_done += nh;
        // Host-free schedule (the asynchronous entry point): every round up to max_iters is enqueued; once a pair is done
        // all its wavefronts exit at their first load, so the rounds nobody needs are a few empty launches on the pose
        // stream.  Otherwise the host reads the number of pairs still active after every round and stops early.
        if (host_free) continue;
        int h_active = 0;
        HIP_TRY(c, hipMemcpyAsync(&h_active, active, sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        if (h_active == 0) break;
    }
    hipLaunchKernelGGL(k_ransac_mask, dim3((kp_cap + 255) / 256, P), dim3(256), 0, st, pw, nmatch, kp_cap, q1, q2, nk, thr2, (uint8_t*)c->d_mask.p);
    HIP_TRY(c, hipGetLastError());
    return VO_OK;
}

int vo_launch_recover(vo_ctx* c, int P, const double K[9], double dist_thresh, hipStream_t st) {
    if (P < 1 || P > c->max_pairs) return vo_fail(c, VO_ERR_INVALID, "pair count exceeds max_pairs");
    const int kp_cap = c->kp_cap;
    const NormK nk = make_normk(K);
    PoseWork* pw = (PoseWork*)c->d_pose.p;
    int* vote = (int*)c->d_vote.p;
    const int pb = (P + 63) / 64;
    hipLaunchKernelGGL(k_decompose, dim3(pb), dim3(64), 0, st, pw, P, vote);
    hipLaunchKernelGGL(k_cheirality, dim3((kp_cap + 63) / 64, P, 2), dim3(64), 0, st, pw, (const int*)c->nmatch(), kp_cap,
                       (const float2*)c->q1(), (const float2*)c->q2(), nk, dist_thresh, vote);
    hipLaunchKernelGGL(k_pose_pick, dim3(pb), dim3(64), 0, st, pw, P, vote);
    HIP_TRY(c, hipGetLastError());
    return VO_OK;
}

// Parity hook for the minimal solver (vo_debug_minimal_solver): one lane per case.
// mode 0: in = 11 coefficients (ascending powers), out = 10 roots as (re, im) pairs
// mode 1: in = x1[5][2], x2[5][2] (normalised points), out = model count followed by 10 row-major E
__global__ __launch_bounds__(DK_T) void k_debug_minimal(int mode, const double* __restrict__ in, int n, double* __restrict__ out) {
    const bool wide = mode >= 2;                 // modes 2, 3: 16 lanes (a DPP row) per case
    const int t = wide ? (blockIdx.x * DK_T + threadIdx.x) >> 4 : blockIdx.x * DK_T + threadIdx.x;
    const int sub = wide ? (int)(threadIdx.x & 15) : -1;
    if (t >= n) return;
    if (mode == 2) {
        double c[11], re[10], im[10];
        for (int i = 0; i < 11; i++) c[i] = in[(size_t)t * 11 + i];
        solve_poly10_wide(c, re, im, sub);
        if (sub == 0) for (int i = 0; i < 10; i++) { out[(size_t)t * 20 + 2 * i] = re[i]; out[(size_t)t * 20 + 2 * i + 1] = im[i]; }
    } else if (mode == 3) {
        double x1[10], x2[10], Eb[90];
        for (int i = 0; i < 10; i++) { x1[i] = in[(size_t)t * 20 + i]; x2[i] = in[(size_t)t * 20 + 10 + i]; }
        for (int i = 0; i < 90; i++) Eb[i] = 0;
        const int nm = five_point<true>(x1, x2, Eb, sub);
        if (sub == 0) {
            out[(size_t)t * 91] = nm;
            for (int i = 0; i < 90; i++) out[(size_t)t * 91 + 1 + i] = (i < 9 * nm || (FP_CLOCKS_ON && i >= 80)) ? Eb[i] : 0.0;
        }
    } else if (mode == 0) {
        double c[11], re[10], im[10];
        for (int i = 0; i < 11; i++) c[i] = in[(size_t)t * 11 + i];
        solve_poly10(c, re, im);
        for (int i = 0; i < 10; i++) { out[(size_t)t * 20 + 2 * i] = re[i]; out[(size_t)t * 20 + 2 * i + 1] = im[i]; }
    } else {
        double x1[10], x2[10], Eb[90];
        for (int i = 0; i < 10; i++) { x1[i] = in[(size_t)t * 20 + i]; x2[i] = in[(size_t)t * 20 + 10 + i]; }
        for (int i = 0; i < 90; i++) Eb[i] = 0;
        const int nm = five_point<false>(x1, x2, Eb);
        out[(size_t)t * 91] = nm;
        for (int i = 0; i < 90; i++) out[(size_t)t * 91 + 1 + i] = (i < 9 * nm || (FP_CLOCKS_ON && i >= 80)) ? Eb[i] : 0.0;
    }
}
int vo_launch_debug_minimal(vo_ctx* c, int mode, const double* d_in, int n, double* d_out, hipStream_t st) {
    const long lanes = mode >= 2 ? (long)n * 16 : n;
    hipLaunchKernelGGL(k_debug_minimal, dim3((unsigned)((lanes + DK_T - 1) / DK_T)), dim3(DK_T), 0, st, mode, d_in, n, d_out);
    HIP_TRY(c, hipGetLastError());
    return VO_OK;
}

int vo_launch_triangulate(vo_ctx* c, const double* dP, const float* dx1, const float* dx2, int n, float* dX4, hipStream_t st) {
    hipLaunchKernelGGL(k_triangulate, dim3((n + 255) / 256), dim3(256), 0, st, dP, dx1, dx2, n, dX4);
    HIP_TRY(c, hipGetLastError());
    return VO_OK;
}
