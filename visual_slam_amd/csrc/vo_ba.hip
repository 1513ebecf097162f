// vo_ba.hip -- bundle-adjustment residual / Jacobian pass on MI355X (gfx950).
//
// Stands behind the residual function the reference hands to scipy.optimize.least_squares,
// /root/reference/libs/vo_utils.py:218-245 (bundle_adjustment_residuals ->
// calc_reprojection_error :184-216 -> cv2.projectPoints with D = 0), called at
// /root/reference/vo.py:118 -- generalised from the reference's single camera to P poses
// (BASELINE.json configs[4]).  scipy differentiates that function by finite
// differences over a dense 6+3N-column Jacobian; here the Jacobians are analytic and reduced on
// the fly to the normal-equation blocks:
//   k_ba_obs    one lane per observation: r (2-vector and its norm = the reference's scalar),
//               Hpl = Jp^T Jl (6x3)
//   k_ba_pose   one workgroup per pose over its observations: Hpp = sum Jp^T Jp, bp = sum Jp^T r
//               (lane-private sums, wavefront shuffle reduction, LDS across wavefronts: no atomics,
//               deterministic)
//   k_ba_point  one lane per point over its observations: Hll = sum Jl^T Jl, bl = sum Jl^T r
// All fp64, compiled with -ffp-contract=off.  fp64 VALU bound; the pass is tiny next to the
// front-end (10^6 observations: a few hundred MFLOP).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <vector>

#include "vo_internal.h"

struct vo_ba_plan {
    int device = 0, P = 0, N = 0, M = 0;
    int *d_obs_pose = nullptr, *d_obs_point = nullptr;
    int *d_pose_start = nullptr, *d_pose_obs = nullptr;      // observations grouped by pose
    int *d_point_start = nullptr, *d_point_obs = nullptr;    // observations grouped by point
    double* d_work = nullptr;                                // staging for the host entry point
    size_t work_bytes = 0;
    hipStream_t stream = nullptr;
};

struct BaK { double fx, fy, cx, cy; };

// cv2.Rodrigues (vector -> matrix): R = c I + (1 - c) k k^T + s [k]x, k = r / theta
__device__ __forceinline__ void ba_rodrigues(const double* r, double R[9]) {
    const double th2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
    const double th = sqrt(th2);
    if (th < DBL_EPSILON) {
        R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
        return;
    }
    const double kx = r[0] / th, ky = r[1] / th, kz = r[2] / th;
    const double c = cos(th), s = sin(th), c1 = 1 - c;
    R[0] = c + c1 * kx * kx;      R[1] = c1 * kx * ky - s * kz; R[2] = c1 * kx * kz + s * ky;
    R[3] = c1 * kx * ky + s * kz; R[4] = c + c1 * ky * ky;      R[5] = c1 * ky * kz - s * kx;
    R[6] = c1 * kx * kz - s * ky; R[7] = c1 * ky * kz + s * kx; R[8] = c + c1 * kz * kz;
}

__device__ __forceinline__ void cross3(const double* a, const double* b, double* o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

// projection + Jacobians of one observation: proj (u,v), Jp[2][6] (d/d rvec, d/d tvec), Jl[2][3] (d/d X)
__device__ void ba_obs(const double* pose, const double* R, const double* X, const BaK& k, double proj[2], double Jp[12], double Jl[6]) {
    const double Y[3] = {R[0] * X[0] + R[1] * X[1] + R[2] * X[2], R[3] * X[0] + R[4] * X[1] + R[5] * X[2], R[6] * X[0] + R[7] * X[1] + R[8] * X[2]};
    const double Xc[3] = {Y[0] + pose[3], Y[1] + pose[4], Y[2] + pose[5]};
    const double iz = 1.0 / Xc[2];
    const double x = Xc[0] * iz, y = Xc[1] * iz;
    proj[0] = k.fx * x + k.cx;
    proj[1] = k.fy * y + k.cy;
    const double J0[3] = {k.fx * iz, 0.0, -k.fx * x * iz};   // d u / d Xc
    const double J1[3] = {0.0, k.fy * iz, -k.fy * y * iz};   // d v / d Xc
    // d (R X) / d r_i = (r_i (r x Y) + (r x (e_i - R e_i)) x Y) / |r|^2   (Gallego & Yezzi 2015); e_i x Y as |r| -> 0
    const double* v = pose;
    const double th2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    double dY[3][3];   // dY[i] = column i
    if (th2 < DBL_EPSILON * DBL_EPSILON) {
        dY[0][0] = 0;     dY[0][1] = -Y[2]; dY[0][2] = Y[1];
        dY[1][0] = Y[2];  dY[1][1] = 0;     dY[1][2] = -Y[0];
        dY[2][0] = -Y[1]; dY[2][1] = Y[0];  dY[2][2] = 0;
    } else {
        double vxY[3];
        cross3(v, Y, vxY);
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const double d[3] = {(i == 0 ? 1.0 : 0.0) - R[0 + i], (i == 1 ? 1.0 : 0.0) - R[3 + i], (i == 2 ? 1.0 : 0.0) - R[6 + i]};
            double w[3], wxY[3];
            cross3(v, d, w);
            cross3(w, Y, wxY);
#pragma unroll
            for (int c = 0; c < 3; c++) dY[i][c] = (v[i] * vxY[c] + wxY[c]) / th2;
        }
    }
#pragma unroll
    for (int i = 0; i < 3; i++) {
        Jp[i] = J0[0] * dY[i][0] + J0[1] * dY[i][1] + J0[2] * dY[i][2];
        Jp[6 + i] = J1[0] * dY[i][0] + J1[1] * dY[i][1] + J1[2] * dY[i][2];
        Jp[3 + i] = J0[i];
        Jp[9 + i] = J1[i];
        Jl[i] = J0[0] * R[i] + J0[1] * R[3 + i] + J0[2] * R[6 + i];
        Jl[3 + i] = J1[0] * R[i] + J1[1] * R[3 + i] + J1[2] * R[6 + i];
    }
}

__global__ __launch_bounds__(256) void k_ba_obs(int M, const int* __restrict__ obs_pose, const int* __restrict__ obs_point,
                                                const double* __restrict__ poses, const double* __restrict__ points,
                                                const double* __restrict__ uv, BaK k, double* __restrict__ r2, double* __restrict__ nrm,
                                                double* __restrict__ Hpl) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= M) return;
    const double* pose = poses + (long)obs_pose[o] * 6;
    const double* X = points + (long)obs_point[o] * 3;
    double R[9], proj[2], Jp[12], Jl[6];
    ba_rodrigues(pose, R);
    ba_obs(pose, R, X, k, proj, Jp, Jl);
    const double r0 = proj[0] - uv[2 * (long)o], r1 = proj[1] - uv[2 * (long)o + 1];
    if (r2) { r2[2 * (long)o] = r0; r2[2 * (long)o + 1] = r1; }
    if (nrm) nrm[o] = sqrt(r0 * r0 + r1 * r1);
    if (Hpl) {
        double* h = Hpl + (long)o * 18;
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) h[a * 3 + b] = Jp[a] * Jl[b] + Jp[6 + a] * Jl[3 + b];
    }
}

__global__ __launch_bounds__(256) void k_ba_pose(const int* __restrict__ pose_start, const int* __restrict__ pose_obs, const int* __restrict__ obs_point,
                                                 const double* __restrict__ poses, const double* __restrict__ points, const double* __restrict__ uv,
                                                 BaK k, double* __restrict__ Hpp, double* __restrict__ bp) {
    __shared__ double part[4][42];
    const int j = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double* pose = poses + (long)j * 6;
    double R[9];
    ba_rodrigues(pose, R);
    double acc[42];
#pragma unroll
    for (int i = 0; i < 42; i++) acc[i] = 0;
    const int s = pose_start[j], e = pose_start[j + 1];
    for (int q = s + tid; q < e; q += 256) {
        const int o = pose_obs[q];
        double proj[2], Jp[12], Jl[6];
        ba_obs(pose, R, points + (long)obs_point[o] * 3, k, proj, Jp, Jl);
        const double r0 = proj[0] - uv[2 * (long)o], r1 = proj[1] - uv[2 * (long)o + 1];
#pragma unroll
        for (int a = 0; a < 6; a++) {
#pragma unroll
            for (int b = 0; b < 6; b++) acc[a * 6 + b] += Jp[a] * Jp[b] + Jp[6 + a] * Jp[6 + b];
            acc[36 + a] += Jp[a] * r0 + Jp[6 + a] * r1;
        }
    }
#pragma unroll
    for (int i = 0; i < 42; i++) {
        double v = acc[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) part[wave][i] = v;
    }
    __syncthreads();
    if (tid < 42) {
        const double v = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
        if (tid < 36) { if (Hpp) Hpp[(long)j * 36 + tid] = v; }
        else if (bp) bp[(long)j * 6 + (tid - 36)] = v;
    }
}

__global__ __launch_bounds__(256) void k_ba_point(int N, const int* __restrict__ point_start, const int* __restrict__ point_obs,
                                                  const int* __restrict__ obs_pose, const double* __restrict__ poses,
                                                  const double* __restrict__ points, const double* __restrict__ uv, BaK k,
                                                  double* __restrict__ Hll, double* __restrict__ bl) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const double* X = points + (long)i * 3;
    double acc[12];
#pragma unroll
    for (int a = 0; a < 12; a++) acc[a] = 0;
    for (int q = point_start[i]; q < point_start[i + 1]; q++) {
        const int o = point_obs[q];
        const double* pose = poses + (long)obs_pose[o] * 6;
        double R[9], proj[2], Jp[12], Jl[6];
        ba_rodrigues(pose, R);
        ba_obs(pose, R, X, k, proj, Jp, Jl);
        const double r0 = proj[0] - uv[2 * (long)o], r1 = proj[1] - uv[2 * (long)o + 1];
#pragma unroll
        for (int a = 0; a < 3; a++) {
#pragma unroll
            for (int b = 0; b < 3; b++) acc[a * 3 + b] += Jl[a] * Jl[b] + Jl[3 + a] * Jl[3 + b];
            acc[9 + a] += Jl[a] * r0 + Jl[3 + a] * r1;
        }
    }
    if (Hll)
#pragma unroll
        for (int a = 0; a < 9; a++) Hll[(long)i * 9 + a] = acc[a];
    if (bl)
#pragma unroll
        for (int a = 0; a < 3; a++) bl[(long)i * 3 + a] = acc[9 + a];
}

// -------------------------------------------------------------------- C ABI
static void ba_free(vo_ba_plan* p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    (void)hipFree(p->d_obs_pose); (void)hipFree(p->d_obs_point); (void)hipFree(p->d_pose_start); (void)hipFree(p->d_pose_obs);
    (void)hipFree(p->d_point_start); (void)hipFree(p->d_point_obs); (void)hipFree(p->d_work);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
}

extern "C" int vo_ba_plan_create(int device, int P, int N, int M, const int32_t* obs_pose, const int32_t* obs_point, vo_ba_plan** out) {
    if (!out) return VO_ERR_INVALID;
    *out = nullptr;
    if (P < 1 || N < 1 || M < 0 || (M > 0 && (!obs_pose || !obs_point))) return VO_ERR_INVALID;
    for (int o = 0; o < M; o++)
        if (obs_pose[o] < 0 || obs_pose[o] >= P || obs_point[o] < 0 || obs_point[o] >= N) return VO_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return VO_ERR_HIP;   // no CPU fallback
    if (hipSetDevice(device) != hipSuccess) return VO_ERR_HIP;
    // counting sorts (stable: observation order inside a group = input order)
    std::vector<int> ps(P + 1, 0), ls(N + 1, 0), po(M), lo(M);
    for (int o = 0; o < M; o++) { ps[obs_pose[o] + 1]++; ls[obs_point[o] + 1]++; }
    for (int j = 0; j < P; j++) ps[j + 1] += ps[j];
    for (int i = 0; i < N; i++) ls[i + 1] += ls[i];
    {
        std::vector<int> pc(ps.begin(), ps.end() - 1), lc(ls.begin(), ls.end() - 1);
        for (int o = 0; o < M; o++) { po[pc[obs_pose[o]]++] = o; lo[lc[obs_point[o]]++] = o; }
    }
    vo_ba_plan* p = new vo_ba_plan;
    p->device = device; p->P = P; p->N = N; p->M = M;
    auto up = [&](int** d, const int* h, size_t n) -> bool {
        if (hipMalloc((void**)d, std::max<size_t>(n, 1) * sizeof(int)) != hipSuccess) return false;
        return n == 0 || hipMemcpy(*d, h, n * sizeof(int), hipMemcpyHostToDevice) == hipSuccess;
    };
    bool ok = up(&p->d_obs_pose, obs_pose, M) && up(&p->d_obs_point, obs_point, M) && up(&p->d_pose_start, ps.data(), P + 1) &&
              up(&p->d_pose_obs, po.data(), M) && up(&p->d_point_start, ls.data(), N + 1) && up(&p->d_point_obs, lo.data(), M);
    ok = ok && hipStreamCreate(&p->stream) == hipSuccess;
    if (!ok) { ba_free(p); return VO_ERR_HIP; }
    *out = p;
    return VO_OK;
}

extern "C" void vo_ba_plan_destroy(vo_ba_plan* p) { ba_free(p); }

extern "C" int vo_ba_pass_device(vo_ba_plan* p, const double* poses, const double* points, const double* uv, const double K[9], double* r2,
                                 double* nrm, double* Hpp, double* bp, double* Hll, double* bl, double* Hpl, void* stream) {
    if (!p || !poses || !points || !K || (p->M > 0 && !uv)) return VO_ERR_INVALID;
    if (hipSetDevice(p->device) != hipSuccess) return VO_ERR_HIP;
    hipStream_t st = stream ? (hipStream_t)stream : p->stream;
    const BaK k{K[0], K[4], K[2], K[5]};
    if (p->M > 0 && (r2 || nrm || Hpl))
        hipLaunchKernelGGL(k_ba_obs, dim3((p->M + 255) / 256), dim3(256), 0, st, p->M, p->d_obs_pose, p->d_obs_point, poses, points, uv, k, r2, nrm, Hpl);
    if (Hpp || bp)
        hipLaunchKernelGGL(k_ba_pose, dim3(p->P), dim3(256), 0, st, p->d_pose_start, p->d_pose_obs, p->d_obs_point, poses, points, uv, k, Hpp, bp);
    if (Hll || bl)
        hipLaunchKernelGGL(k_ba_point, dim3((p->N + 255) / 256), dim3(256), 0, st, p->N, p->d_point_start, p->d_point_obs, p->d_obs_pose, poses, points,
                           uv, k, Hll, bl);
    if (hipGetLastError() != hipSuccess) return VO_ERR_HIP;
    if (!stream && hipStreamSynchronize(st) != hipSuccess) return VO_ERR_HIP;
    return VO_OK;
}

extern "C" int vo_ba_pass(vo_ba_plan* p, const double* poses, const double* points, const double* uv, const double K[9], double* r2, double* nrm,
                          double* Hpp, double* bp, double* Hll, double* bl, double* Hpl) {
    if (!p || !poses || !points || !K) return VO_ERR_INVALID;
    if (hipSetDevice(p->device) != hipSuccess) return VO_ERR_HIP;
    const size_t P = p->P, N = p->N, M = p->M;
    // layout of the staging buffer (doubles)
    const size_t o_pose = 0, o_pts = o_pose + 6 * P, o_uv = o_pts + 3 * N, o_r = o_uv + 2 * M, o_n = o_r + 2 * M, o_Hpp = o_n + M,
                 o_bp = o_Hpp + 36 * P, o_Hll = o_bp + 6 * P, o_bl = o_Hll + 9 * N, o_Hpl = o_bl + 3 * N, total = o_Hpl + 18 * M;
    if (p->work_bytes < total * 8) {
        (void)hipFree(p->d_work);
        p->d_work = nullptr; p->work_bytes = 0;
        if (hipMalloc((void**)&p->d_work, total * 8) != hipSuccess) return VO_ERR_HIP;
        p->work_bytes = total * 8;
    }
    double* w = p->d_work;
    if (hipMemcpy(w + o_pose, poses, 48 * P, hipMemcpyHostToDevice) != hipSuccess) return VO_ERR_HIP;
    if (hipMemcpy(w + o_pts, points, 24 * N, hipMemcpyHostToDevice) != hipSuccess) return VO_ERR_HIP;
    if (M && hipMemcpy(w + o_uv, uv, 16 * M, hipMemcpyHostToDevice) != hipSuccess) return VO_ERR_HIP;
    int rc = vo_ba_pass_device(p, w + o_pose, w + o_pts, w + o_uv, K, r2 ? w + o_r : nullptr, nrm ? w + o_n : nullptr, Hpp ? w + o_Hpp : nullptr,
                               bp ? w + o_bp : nullptr, Hll ? w + o_Hll : nullptr, bl ? w + o_bl : nullptr, Hpl ? w + o_Hpl : nullptr, nullptr);
    if (rc != VO_OK) return rc;
    auto down = [&](double* h, size_t off, size_t n) { return !h || n == 0 || hipMemcpy(h, w + off, n * 8, hipMemcpyDeviceToHost) == hipSuccess; };
    if (!(down(r2, o_r, 2 * M) && down(nrm, o_n, M) && down(Hpp, o_Hpp, 36 * P) && down(bp, o_bp, 6 * P) && down(Hll, o_Hll, 9 * N) &&
          down(bl, o_bl, 3 * N) && down(Hpl, o_Hpl, 18 * M)))
        return VO_ERR_HIP;
    return VO_OK;
}
