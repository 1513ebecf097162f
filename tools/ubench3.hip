// micro-benchmark 3: issue cost (cycles per wave64 instruction at full occupancy) of the byte-parallel / bit-logic
// instructions a SWAR form of the FAST pretest would be made of (v_lerp_u8 as a per-byte compare, v_alignbit/alignbyte,
// plain logic ops, DPP adds, v_bcnt, v_ffbl).  hipcc --offload-arch=gfx950 -O3 tools/ubench3.hip -o tools/ubench3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed, int iters) {
    uint32_t a[8], acc[8];
    for (int i = 0; i < 8; i++) { a[i] = seed * (threadIdx.x + i + 1); acc[i] = i + threadIdx.x; }
    for (int it = 0; it < iters; it++) {
#define OP(i)                                                                                                            \
    if (MODE == 0) asm volatile("v_lerp_u8 %0, %0, %1, %1" : "+v"(acc[i]) : "v"(a[i]));                                   \
    if (MODE == 1) asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(acc[i]) : "v"(a[i]));                              \
    if (MODE == 2) asm volatile("v_alignbyte_b32 %0, %0, %1, 3" : "+v"(acc[i]) : "v"(a[i]));                              \
    if (MODE == 3) asm volatile("v_and_b32_e32 %0, %1, %0" : "+v"(acc[i]) : "v"(a[i]));                                   \
    if (MODE == 4) asm volatile("v_or_b32_e32 %0, %1, %0" : "+v"(acc[i]) : "v"(a[i]));                                    \
    if (MODE == 5) asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(acc[i]) : "v"(a[i]));                                   \
    if (MODE == 6) asm volatile("v_not_b32_e32 %0, %0" : "+v"(acc[i]));                                                   \
    if (MODE == 7) asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(acc[i]) : "v"(a[i]));                                   \
    if (MODE == 8) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[i]) : "v"(a[i]));                                  \
    if (MODE == 9) asm volatile("v_ffbl_b32_e32 %0, %0" : "+v"(acc[i]));                                                  \
    if (MODE == 10) asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(acc[i]));                                                \
    if (MODE == 11) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(acc[i]) : "v"(a[i]));                                  \
    if (MODE == 12) asm volatile("v_sad_u8 %0, %0, %1, %1" : "+v"(acc[i]) : "v"(a[i]));                                   \
    if (MODE == 13) asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(acc[i]) : "v"(a[i]));                             \
    if (MODE == 14) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(acc[i]) : "v"(a[i]));                             \
    if (MODE == 15) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]));         \
    if (MODE == 16) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(a[i])); \
    if (MODE == 17) asm volatile("v_and_b32_e32 %0, 0x80808080, %0" : "+v"(acc[i]));                                      \
    if (MODE == 18) asm volatile("v_and_b32_e32 %0, %1, %0" : "+v"(acc[i]) : "s"(seed));                                  \
    if (MODE == 19) asm volatile("v_lshlrev_b32_e32 %0, 1, %0" : "+v"(acc[i]));                                           \
    if (MODE == 20) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(acc[i]) : "v"(a[i]));                              \
    if (MODE == 21) asm volatile("v_sub_u32_e32 %0, %1, %0" : "+v"(acc[i]) : "v"(a[i]));                                  \
    if (MODE == 22) asm volatile("v_xad_u32 %0, %0, %1, %1" : "+v"(acc[i]) : "v"(a[i]));                                  \
    if (MODE == 23) asm volatile("v_or3_b32 %0, %0, %1, %1" : "+v"(acc[i]) : "v"(a[i]));                                  \
    if (MODE == 24) asm volatile("v_cmp_ne_u32_e32 vcc, 0, %0\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc" : "+v"(acc[i]) : : "vcc"); \
    if (MODE == 25) asm volatile("v_msad_u8 %0, %0, %1, %1" : "+v"(acc[i]) : "v"(a[i]));                                  \
    if (MODE == 26) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(acc[i]) : "v"(a[i]));                                   \
    if (MODE == 27) asm volatile("v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(acc[i]) : "v"(a[i]));
        REP8(OP)
#undef OP
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s += acc[i];
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
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 12345u, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 12345u, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)blocks * 4 / 1024.0 * iters * n;
    printf("%-22s %.3f ms  %.2f cycles/instr @2.4GHz\n", name, ms, ms * 1e-3 * 2.4e9 / per_simd);
    hipFree(d);
}
int main(int argc, char** argv) {
    if (argc > 1) g_blocks = atoi(argv[1]);
    printf("blocks %d (%.2f waves/SIMD)\n", g_blocks, g_blocks / 256.0);
    run<0>("lerp_u8", 8); run<1>("alignbit_b32", 8); run<2>("alignbyte_b32", 8); run<3>("and_b32", 8); run<4>("or_b32", 8); run<5>("xor_b32", 8);
    run<6>("not_b32", 8); run<7>("add_u32", 8); run<21>("sub_u32", 8); run<8>("bcnt_u32_b32", 8); run<9>("ffbl_b32", 8); run<10>("bfe_u32", 8); run<11>("bfi_b32", 8);
    run<12>("sad_u8", 8); run<25>("msad_u8", 8); run<13>("pk_sub_u16 clamp", 8); run<26>("pk_min_u16", 8); run<14>("mbcnt_lo", 8); run<15>("add_u32_dpp row_shr", 8); run<16>("mov_b32_dpp row_shr", 8);
    run<17>("and_b32 literal", 8); run<18>("and_b32 sgpr", 8); run<19>("lshlrev_b32", 8); run<20>("lshl_add_u32", 8); run<22>("xad_u32", 8); run<23>("or3_b32", 8);
    run<24>("cmp_ne + addc (2)", 16); run<27>("and_b32_sdwa", 8);
    return 0;
}
