// fp64 issue rate of ONE wave (the five-point solver's situation at small batches): dependent chains of v_mul_f64 /
// v_add_f64 at ILP 1..8.   hipcc -w -O3 --offload-arch=gfx950 -ffp-contract=off tools/ubench_f64.hip -o tools/ubench_f64
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ILP>
__global__ __launch_bounds__(64) void k(double* out, double a, double b, int iters) {
    double x[ILP];
#pragma unroll
    for (int i = 0; i < ILP; i++) x[i] = a + i + threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int i = 0; i < ILP; i++) { x[i] = x[i] * a; x[i] = x[i] + b; }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < ILP; i++) s += x[i];
    out[threadIdx.x + blockIdx.x * 64] = s;
}
template <int ILP>
void run(double* d, int blocks) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<ILP><<<blocks, 64>>>(d, 1.0000001, 1e-9, 10);
    hipEventRecord(e0);
    k<ILP><<<blocks, 64>>>(d, 1.0000001, 1e-9, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)iters * 16 * ILP;
    printf("blocks %5d ILP %d: %.3f ms, %.2f ns / wave instruction (%.1f cycles at 2.4 GHz)\n", blocks, ILP, ms, ms * 1e6 / n, ms * 1e6 / n * 2.4);
}
int main() {
    double* d; hipMalloc(&d, 64 * 4096 * 8);
    for (int blocks : {1, 1024, 2048}) { run<1>(d, blocks); run<2>(d, blocks); run<4>(d, blocks); run<8>(d, blocks); }
    return 0;
}
