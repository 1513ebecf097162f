/*
 * vo_sincosf.h -- glibc (>= 2.28, "optimized routines") cosf / sinf restated for |x| < 120, usable from device and
 * host code.  Needed for vo_orb_params.trig_mode = 1: if OpenCV's `cos(angle)` on a float binds to
 * the float overload, the rBRIEF rotation uses the C library's cosf/sinf, whose results are NOT the correctly
 * rounded ones (max error 0.56 ulp), so the device has to run the same algorithm to give the same bits:
 * double-precision range reduction by pi/2 (n = round(x * 2/pi), x - n * pi/2) and two degree-8/9 polynomials.
 * tests/test_sincosf_cpu.py compiles this header with gcc and compares it with the host libm over every float
 * in [0, 6.3] in a strided sweep (the full 1.09e9-value sweep was run once: no difference, with and without FMA
 * contraction -- the rounding to float absorbs it).
 */
#ifndef VO_SINCOSF_H
#define VO_SINCOSF_H
#include <stdint.h>
#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define VO_HD __host__ __device__ __forceinline__
#else
#define VO_HD static inline
#endif

VO_HD uint32_t vo_sc_abstop12(float x) {
    union { float f; uint32_t u; } v;
    v.f = x;
    return (v.u >> 20) & 0x7ff;
}

/* n even: sine polynomial; n odd: cosine polynomial (negated coefficients when `neg`) */
VO_HD float vo_sc_poly(double x, double x2, int neg, int n) {
    const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
    if ((n & 1) == 0) {
        const double x3 = x * x2;
        const double s1 = S2 + x2 * S3;
        const double x7 = x3 * x2;
        const double s = x + x3 * S1;
        return (float)(s + x7 * s1);
    }
    const double sg = neg ? -1.0 : 1.0;
    const double C0 = sg * 0x1p0, C1 = sg * -0x1.ffffffd0c621cp-2, C2 = sg * 0x1.55553e1068f19p-5, C3 = sg * -0x1.6c087e89a359dp-10,
                 C4 = sg * 0x1.99343027bf8c3p-16;
    const double x4 = x2 * x2;
    const double c2 = C3 + x2 * C4;
    const double c1 = C0 + x2 * C1;
    const double x6 = x4 * x2;
    const double c = c1 + x4 * C2;
    return (float)(c + x6 * c2);
}

VO_HD double vo_sc_reduce(double x, int* np) {
    const double r = x * 0x1.45F306DC9C883p+23;          /* 2/pi * 2^24: the quadrant ends up in bits 24..31 */
    const int n = ((int32_t)r + 0x800000) >> 24;
    *np = n;
    return x - n * 0x1.921FB54442D18p0;
}

VO_HD float vo_glibc_cosf(float y) {
    double x = y;
    if (vo_sc_abstop12(y) < vo_sc_abstop12(0x1.921FB6p-1f)) {
        if (vo_sc_abstop12(y) < vo_sc_abstop12(0x1p-12f)) return 1.0f;
        return vo_sc_poly(x, x * x, 0, 1);
    }
    int n;
    x = vo_sc_reduce(x, &n);
    const double s = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
    return vo_sc_poly(x * s, x * x, (n & 2) != 0, n ^ 1);
}

VO_HD float vo_glibc_sinf(float y) {
    double x = y;
    if (vo_sc_abstop12(y) < vo_sc_abstop12(0x1.921FB6p-1f)) {
        if (vo_sc_abstop12(y) < vo_sc_abstop12(0x1p-12f)) return y;
        return vo_sc_poly(x, x * x, 0, 0);
    }
    int n;
    x = vo_sc_reduce(x, &n);
    const double s = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
    return vo_sc_poly(x * s, x * x, (n & 2) != 0, n);
}
#endif
