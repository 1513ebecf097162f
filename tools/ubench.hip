// micro-benchmark: issue rate of v_bcnt_u32_b32 / v_xor / v_add on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed, int iters, const uint32_t* __restrict__ u) {
    uint32_t a[8], acc[8];
    for (int i = 0; i < 8; i++) { a[i] = seed * (threadIdx.x + i + 1); acc[i] = i; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (MODE == 0) acc[i] = __popc(a[i] ^ acc[i]) + acc[i];       // xor + bcnt(acc)
            if (MODE == 1) acc[i] = (a[i] ^ acc[i]) + acc[i];              // xor + add
            if (MODE == 2) acc[i] = __popc(a[i]) + acc[i];                 // bcnt only
            if (MODE == 3) acc[i] = a[i] + acc[i] * 3u;                    // mad-ish
            if (MODE == 4) acc[i] = (u[(it & 7) * 8 + i] ^ a[i]) + acc[i];    // xor with SGPR operand + add
            if (MODE == 5) acc[i] = __popc(u[(it & 7) * 8 + i] ^ a[i]) + acc[i]; // xor(SGPR) + bcnt(acc)
            if (MODE == 6) acc[i] = min(a[i], acc[i]) + 1;                // min + add
            if (MODE == 7) acc[i] = __builtin_rotateleft32(acc[i], 16) + a[i];   // alignbit + add
            if (MODE == 8) acc[i] = min(max(a[i], acc[i]), max(min(a[i], acc[i]), a[(i + 1) & 7])) + 1;   // med3 + add
            if (MODE == 9) acc[i] = __umul24(acc[i], 0x204081u) + a[i];   // mul_u24 + add (or mad_u24)
            if (MODE == 10) acc[i] = (acc[i] << 3) + a[i];                 // lshl_add
            if (MODE == 11) acc[i] = ((acc[i] >> 4) & 0xF) + a[i];         // bfe + add
            if (MODE == 12) acc[i] = (acc[i] & a[i]) + 1;                  // and + add
            if (MODE == 13) acc[i] = __float_as_uint(__fmaf_rn(__uint_as_float(a[i]), 1.0001f, __uint_as_float(acc[i])));   // v_fma_f32
            if (MODE == 15) acc[i] = __builtin_amdgcn_perm(acc[i], a[i], 0x07020500u) + 1;   // v_perm + add
            if (MODE == 16) acc[i] = __builtin_amdgcn_sad_u8(acc[i], a[i], 1u);              // v_sad_u8
            if (MODE == 17) acc[i] = __builtin_amdgcn_udot4(acc[i], a[i], 1u, false);        // v_dot4_u32_u8
        }
    }
    if (MODE == 14) {   // v_pk_fma_f32: 4 packed fmas = 8 fp32 fmas per trip
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 x[4], y[4];
        for (int i = 0; i < 4; i++) { x[i] = f2{__uint_as_float(a[2 * i]), __uint_as_float(a[2 * i + 1])}; y[i] = f2{(float)i, 1.f}; }
        const f2 m = f2{1.0001f, 0.9999f};
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int i = 0; i < 4; i++) y[i] = __builtin_elementwise_fma(x[i], m, y[i]);
        for (int i = 0; i < 4; i++) { acc[2 * i] = __float_as_uint(y[i].x); acc[2 * i + 1] = __float_as_uint(y[i].y); }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
static int g_blocks = 256 * 8;
template <int MODE>
void run(const char* name, int ops) {
    uint32_t* d;
    hipMalloc(&d, 256 * 8 * 256 * 4 * 4);
    const int blocks = g_blocks, iters = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 12345u, iters, (const uint32_t*)d);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 12345u, iters, (const uint32_t*)d);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double laneops = (double)blocks * 256 * iters * 8 * ops;
    printf("%-12s %.3f ms  %.2f T lane-instr/s\n", name, ms, laneops / (ms * 1e-3) / 1e12);
    hipFree(d);
}
int main(int argc, char** argv) {
    if (argc > 1) g_blocks = atoi(argv[1]);
    printf("blocks %d (%.2f waves/SIMD)\n", g_blocks, g_blocks / 256.0);
    run<0>("xor+bcnt", 2);
    run<1>("xor+add", 2);
    run<2>("bcnt", 1);
    run<3>("mad", 1);
    run<4>("xorS+add", 2);
    run<5>("xorS+bcnt", 2);
    run<6>("min+add", 2);
    run<7>("alignbit+add", 2);
    run<8>("med3+add", 2);
    run<9>("mul24+add", 2);
    run<10>("lshl_add", 1);
    run<11>("bfe+add", 2);
    run<12>("and+add", 2);
    run<13>("fma_f32", 1);
    run<14>("pk_fma_f32", 1);
    run<15>("perm+add", 2);
    run<16>("sad_u8", 1);
    run<17>("dot4_u8", 1);
    return 0;
}
