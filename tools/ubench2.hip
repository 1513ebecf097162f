// micro-benchmark 2: issue cost (cycles per wave64 instruction at full occupancy) of the float / convert /
// pack instructions the blur and pyramid kernels are made of.  hipcc --offload-arch=gfx950 -O3 tools/ubench2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed, int iters, float sk) {
    uint32_t a[8], acc[8];
    for (int i = 0; i < 8; i++) { a[i] = seed * (threadIdx.x + i + 1); acc[i] = i + threadIdx.x; }
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 x[4], y[4];
    for (int i = 0; i < 4; i++) { x[i] = f2{(float)a[2 * i], (float)a[2 * i + 1]}; y[i] = f2{(float)i, 1.f}; }
    const f2 m = f2{sk, sk};
    for (int it = 0; it < iters; it++) {
#define OP(i)                                                                                                                   \
    if (MODE == 0) asm volatile("v_cvt_f32_ubyte1_e32 %0, %0" : "+v"(acc[i]));                                                   \
    if (MODE == 1) asm volatile("v_rndne_f32_e32 %0, %0" : "+v"(acc[i]));                                                        \
    if (MODE == 2) asm volatile("v_cvt_i32_f32_e32 %0, %0" : "+v"(acc[i]));                                                      \
    if (MODE == 3) asm volatile("v_med3_i32 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(a[i]));                                          \
    if (MODE == 4) asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(acc[i]) : "v"(a[i]));                                       \
    if (MODE == 5) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(acc[i]) : "v"(a[i]));                                          \
    if (MODE == 6) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(acc[i]) : "s"(sk), "v"(a[i]));                                \
    if (MODE == 7) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(acc[i]) : "v"(a[i]));                                     \
    if (MODE == 8) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(acc[i]) : "s"(sk));                                            \
    if (MODE == 12) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(acc[i]) : "v"(a[i]));                                     \
    if (MODE == 13) asm volatile("v_cvt_f32_ubyte0_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "+v"(acc[i])); \
    if (MODE == 14) asm volatile("v_max_i32_e32 %0, %1, %0" : "+v"(acc[i]) : "v"(a[i]));                                         \
    if (MODE == 15) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(acc[i]) : "v"(a[i]));                                        \
    if (MODE == 16) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(acc[i]) : "v"(a[i]));                                          \
    if (MODE == 17) asm volatile("v_mul_u32_u24_e32 %0, %0, %1" : "+v"(acc[i]) : "v"(a[i]));                                     \
    if (MODE == 18) asm volatile("v_lshrrev_b32_e32 %0, 3, %0" : "+v"(acc[i]));                                                  \
    if (MODE == 19) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(acc[i]) : "v"(a[i]));                                        \
    if (MODE == 20) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(acc[i]) : "v"(a[i]));                                      \
    if (MODE == 21) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(acc[i]) : "v"(a[i]));                                          \
    if (MODE == 22) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(acc[i]) : "v"(a[i]));                                          \
    if (MODE == 23) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(acc[i]) : "v"(a[i]));                                       \
    if (MODE == 24) asm volatile("v_pk_mad_u16 %0, %0, %1, %1" : "+v"(acc[i]) : "v"(a[i]));                                      \
    if (MODE == 25) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(acc[i]) : "v"(a[i]) : "vcc");                        \
    if (MODE == 26) asm volatile("v_cmp_gt_u32_e32 vcc, %0, %1" : : "v"(acc[i]), "v"(a[i]) : "vcc");                             \
    if (MODE == 27) asm volatile("v_mov_b32_e32 %0, %1" : "=v"(acc[i]) : "v"(a[i]));                                             \
    if (MODE == 28) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(acc[i]) : "v"(a[i]));                                         \
    if (MODE == 29) asm volatile("v_max3_u32 %0, %0, %1, %1" : "+v"(acc[i]) : "v"(a[i]));                                        \
    if (MODE == 30) asm volatile("v_min_u32_e32 %0, %1, %0" : "+v"(acc[i]) : "v"(a[i]));                                         \
    if (MODE == 31) asm volatile("v_sub_u32_e32 %0, %1, %0" : "+v"(acc[i]) : "v"(a[i]));
        REP8(OP)
#undef OP
        if (MODE == 9) for (int i = 0; i < 4; i++) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(y[i]) : "v"(x[i]));
        if (MODE == 10) for (int i = 0; i < 4; i++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(y[i]) : "v"(m), "v"(x[i]));
        if (MODE == 11) for (int i = 0; i < 4; i++) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(y[i]) : "v"(m));
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s += acc[i];
    for (int i = 0; i < 4; i++) s += (uint32_t)(y[i].x + y[i].y);
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
static int g_blocks = 256 * 8;
template <int MODE>
void run(const char* name, int n) {
    uint32_t* d;
    hipMalloc(&d, (size_t)g_blocks * 256 * 4);
    const int blocks = g_blocks, iters = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 12345u, iters, 1.0001f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 12345u, iters, 1.0001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD = blocks*4 waves / 1024 SIMDs * iters * n
    const double per_simd = (double)blocks * 4 / 1024.0 * iters * n;
    printf("%-22s %.3f ms  %.2f cycles/instr @2.4GHz\n", name, ms, ms * 1e-3 * 2.4e9 / per_simd);
    hipFree(d);
}
int main(int argc, char** argv) {
    if (argc > 1) g_blocks = atoi(argv[1]);
    printf("blocks %d (%.2f waves/SIMD)\n", g_blocks, g_blocks / 256.0);
    run<0>("cvt_f32_ubyte1", 8); run<13>("cvt_f32_ubyte0_sdwa", 8); run<1>("rndne_f32", 8); run<2>("cvt_i32_f32", 8); run<3>("med3_i32", 8);
    run<4>("lshl_or_b32", 8); run<5>("add_f32", 8); run<6>("fmac_f32 (sgpr)", 8); run<28>("fma_f32 vop3", 8); run<7>("cvt_pk_u8_f32", 8); run<8>("mul_f32 (sgpr)", 8);
    run<9>("pk_add_f32", 4); run<10>("pk_fma_f32", 4); run<11>("pk_mul_f32", 4); run<12>("mad_u32_u24", 8); run<14>("max_i32", 8); run<30>("min_u32", 8);
    run<29>("max3_u32", 8); run<15>("perm_b32", 8); run<16>("mul_lo_u32", 8); run<17>("mul_u32_u24", 8); run<18>("lshrrev_b32", 8); run<31>("sub_u32", 8);
    run<19>("add3_u32", 8); run<20>("and_or_b32", 8); run<21>("pk_max_u16", 8); run<22>("pk_add_u16", 8); run<23>("pk_mul_lo_u16", 8);
    run<24>("pk_mad_u16", 8); run<25>("cndmask_b32", 8); run<26>("cmp_gt_u32", 8); run<27>("mov_b32", 8);
    return 0;
}
