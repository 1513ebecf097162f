// micro-benchmark: issue-to-issue cost of v_mfma_i32_32x32x32_i8 in one dependent accumulator chain vs two
// independent chains, at 1 and 2 wavefronts per SIMD.   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
template <int CHAINS>
__global__ __launch_bounds__(256) void k(int* out, int iters) {
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, 6, (int)threadIdx.x};
    v16i c0 = {0}, c1 = {0}, c2 = {0};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int s = 0; s < 8; s++) {
            c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
            if (CHAINS >= 2) c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
            if (CHAINS >= 3) c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c2, 0, 0, 0);
        }
    }
    int s = 0;
    for (int i = 0; i < 16; i++) s += c0[i] + c1[i] + c2[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CHAINS>
void run(int blocks) {
    int* d;
    (void)hipMalloc(&d, (size_t)blocks * 256 * 4);
    const int iters = 2000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)blocks * 4 / 1024.0 * iters * 8 * CHAINS;   // MFMAs per SIMD
    printf("chains %d, %.2f waves/SIMD: %.3f ms, %.1f cycles per MFMA per SIMD @2.4GHz, %.0f TOP/s\n", CHAINS, blocks / 256.0, ms,
           ms * 1e-3 * 2.4e9 / per_simd, (double)blocks * 4 * iters * 8 * CHAINS * 65536.0 / (ms * 1e-3) / 1e12);
    (void)hipFree(d);
}
int main() {
    for (int blocks : {256, 512, 1024}) { run<1>(blocks); run<2>(blocks); run<3>(blocks); }
    return 0;
}
