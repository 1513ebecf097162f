// micro-benchmark: does VALU work issue in the shadow of v_mfma_i32_32x32x32_i8?  One MFMA followed by NV independent
// half-rate VALU ops (v_med3_u32 / v_min_u32), 1 and 2 wavefronts per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
template <int NV>
__global__ __launch_bounds__(256) void k(int* out, int iters) {
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, 6, (int)threadIdx.x};
    v16i c0 = {0};
    uint32_t x[8];
    for (int i = 0; i < 8; i++) x[i] = threadIdx.x * (i + 3);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int s = 0; s < 8; s++) {
            c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; v++) {
                if (v & 1) asm volatile("v_min_u32_e32 %0, %1, %0" : "+v"(x[v & 7]) : "v"(x[(v + 1) & 7]));
                else asm volatile("v_med3_u32 %0, %0, %1, %2" : "+v"(x[v & 7]) : "v"(x[(v + 1) & 7]), "v"(x[(v + 2) & 7]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    int s = 0;
    for (int i = 0; i < 16; i++) s += c0[i];
    for (int i = 0; i < 8; i++) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NV>
void run(int blocks) {
    int* d;
    (void)hipMalloc(&d, (size_t)blocks * 256 * 4);
    const int iters = 2000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NV>, dim3(blocks), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<NV>, dim3(blocks), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)blocks * 4 / 1024.0 * iters * 8;
    printf("NV %2d, %.0f waves/SIMD: %.1f cycles per (MFMA + NV VALU) per SIMD @2.4GHz\n", NV, blocks / 256.0, ms * 1e-3 * 2.4e9 / per_simd);
    (void)hipFree(d);
}
int main() {
    for (int blocks : {256, 512}) { run<0>(blocks); run<2>(blocks); run<4>(blocks); run<6>(blocks); run<8>(blocks); run<12>(blocks); }
    return 0;
}
