// Layout / scale check of v_mfma_scale_f32_32x32x64_f8f6f4 with fp4 (E2M1) operands, on exact integer data:
// A[32][64] in {0, 1}, B[64][32] in {+1, -1}, both block scales 2^6, so D = 4096 * (A B) exactly in f32.
// Hypothesis: lane l holds A[row l&31][k = 32 (l>>5) + j] and B[k = 32 (l>>5) + j][col l&31] as nibble j (low nibble first) of
// its first four operand dwords; D as every 32x32 MFMA: col = l&31, row = (i&3) + 8 (i>>2) + 4 (l>>5).
// hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_fp4.hip -o /tmp/fp4chk && /tmp/fp4chk
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
__global__ void k(const uint32_t* a, const uint32_t* b, float* c, int sa, int sb, int reps) {
    v8i A, B;
    for (int i = 0; i < 8; i++) { A[i] = i < 4 ? (int)a[threadIdx.x * 4 + i] : 0; B[i] = i < 4 ? (int)b[threadIdx.x * 4 + i] : 0; }
    v16f acc;
    for (int i = 0; i < 16; i++) acc[i] = (float)(1048576 + 2048 + (i & 3) + 8 * (i >> 2));
    for (int r = 0; r < reps; r++) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc, 4, 4, 0, sa, 0, sb);
    for (int i = 0; i < 16; i++) c[(blockIdx.x * 64 + threadIdx.x) * 16 + i] = acc[i];
}
int main() {
    static int Am[32][64], Bm[64][32];
    srand(1);
    for (int r = 0; r < 32; r++) for (int kk = 0; kk < 64; kk++) Am[r][kk] = rand() & 1;
    for (int kk = 0; kk < 64; kk++) for (int cidx = 0; cidx < 32; cidx++) Bm[kk][cidx] = (rand() & 1) ? 1 : -1;
    uint32_t ha[256] = {0}, hb[256] = {0};
    for (int l = 0; l < 64; l++)
        for (int j = 0; j < 32; j++) {
            const int kk = 32 * (l >> 5) + j;
            const uint32_t na = Am[l & 31][kk] ? 0x2u : 0x0u, nb = Bm[kk][l & 31] > 0 ? 0x2u : 0xAu;
            ha[l * 4 + j / 8] |= na << (4 * (j % 8));
            hb[l * 4 + j / 8] |= nb << (4 * (j % 8));
        }
    uint32_t *da, *db; float* dc;
    hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dc, 64 * 16 * 4);
    hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dc, 133, 133, 1);
    float hc[1024];
    hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++)
        for (int i = 0; i < 16; i++) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * (l >> 5), col = l & 31;
            int s = 0;
            for (int kk = 0; kk < 64; kk++) s += Am[row][kk] * Bm[kk][col];
            const float want = (float)(1048576 + 2048 + (i & 3) + 8 * (i >> 2)) + 4096.f * s;
            if (hc[l * 16 + i] != want) { if (bad < 8) printf("lane %d reg %d: got %.1f want %.1f\n", l, i, hc[l * 16 + i], want); bad++; }
        }
    printf("fp4 32x32x64, scales 2^6 x 2^6: %d of 1024 results differ from the hypothesis\n", bad);
    // rate: many blocks, dependent chain
    float* dbig; hipMalloc(&dbig, 2048 * 64 * 16 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(2048), dim3(64), 0, 0, da, db, dbig, 133, 133, 4096);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(2048), dim3(64), 0, 0, da, db, dbig, 133, 133, 4096);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("2048 wavefronts x 4096 dependent MFMAs: %.3f ms = %.1f cycles per MFMA per SIMD at 2.4 GHz (2 wavefronts per SIMD), %.2f POP/s\n", ms,
           ms * 1e-3 * 2.4e9 / (4096.0 * 2), 2048.0 * 4096 * 32 * 32 * 64 * 2 / (ms * 1e-3) / 1e15);
    return bad != 0;
}
