// vo_ba.hip -- bundle-adjustment residual / Jacobian pass on MI355X (gfx950).
//
// Stands behind the residual function the reference hands to scipy.optimize.least_squares,
// /root/reference/libs/vo_utils.py:218-245 (bundle_adjustment_residuals ->
// calc_reprojection_error :184-216 -> cv2.projectPoints with D = 0), called at
// /root/reference/vo.py:118 -- generalised from the reference's single camera to P poses
// (BASELINE.json configs[4]).  scipy differentiates that function by finite
// differences over a dense 6+3N-column Jacobian; here the Jacobians are analytic and reduced on
// the fly to the normal-equation blocks:
//   k_ba_rot    one lane per pose: Rodrigues once per pose into a 128-byte table row (R, t, rvec, |rvec|^2), so that no
//               other kernel evaluates sin / cos (round 2 did it once per observation, three times over)
//   obs role    one lane per observation: r (2-vector and its norm = the reference's scalar), Hpl = Jp^T Jl (6x3).
//               HBM bound: 192 B per observation, 144 of them Hpl, which goes through LDS so that a wavefront stores
//               contiguous 16-byte runs (a lane's own 18 doubles are 144 bytes apart from its neighbour's)
//   pose role   one workgroup per pose over its observations: Hpp = sum Jp^T Jp (the 21 unique entries), bp = sum Jp^T r;
//               lane-private sums over observations taken four at a time (all gathers of a lane in flight together),
//               DPP row reduction inside the wavefront, LDS across wavefronts: no atomics, deterministic
//   point role  one lane per point over its observations: Hll = sum Jl^T Jl, bl = sum Jl^T r (needs R only, no d/d rvec)
// The three reductions are ONE launch (k_ba_pass: pose workgroups, then point workgroups, then observation workgroups),
// so the latency-bound gathers of the first two overlap the HBM-bound observation stream without any host-side forks.
// All fp64, compiled with -ffp-contract=off.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <vector>

#include "vo_internal.h"

struct vo_ba_plan {
    int device = 0, P = 0, N = 0, M = 0;
    int *d_obs_pose = nullptr, *d_obs_point = nullptr;
    int *d_pose_start = nullptr, *d_pose_obs = nullptr, *d_pose_pt = nullptr;        // observations grouped by pose (+ their point index)
    int *d_point_start = nullptr, *d_point_obs = nullptr, *d_point_pose = nullptr;   // observations grouped by point (+ their pose index)
    double* d_tab = nullptr;                                 // [P][BA_TAB]: R (9), t (3), W (9)
    int *d_blk_start = nullptr, *d_empty = nullptr;          // observation blocks (<= 256 each); points nobody observes
    int n_blk = 0, n_empty = 0;
    bool fuse_points = false;                                // observations sorted by point, blocks cut at point boundaries
    double* d_work = nullptr;                                // staging for the host entry point
    size_t work_bytes = 0;
    hipStream_t stream = nullptr;
    bool timing = false;                                     // vo_ba_set_kernel_timing
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    double t_ms[2] = {0, 0};
    int t_n[2] = {0, 0};
};

struct BaK { double fx, fy, cx, cy; };
#define BA_TAB 22   // doubles per pose in the table: R (9), t (3), W (9), pad: 176 bytes, rows 16-byte aligned

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

__global__ __launch_bounds__(256) void k_ba_rot(int P, const double* __restrict__ poses, double* __restrict__ tab, int n_empty,
                                                const int* __restrict__ empty_points, double* __restrict__ Hll, double* __restrict__ bl) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j < n_empty) {   // points without observations: their blocks are zero, and (with the point sums fused into the observation role) no other lane writes them
        const long i = empty_points[j];
        if (Hll) for (int a = 0; a < 9; a++) Hll[i * 9 + a] = 0;
        if (bl) for (int a = 0; a < 3; a++) bl[i * 3 + a] = 0;
    }
    if (j >= P) return;
    const double* pose = poses + (long)j * 6;
    double R[9];
    ba_rodrigues(pose, R);
    double* t = tab + (long)j * BA_TAB;
#pragma unroll
    for (int i = 0; i < 9; i++) t[i] = R[i];
    t[9] = pose[3]; t[10] = pose[4]; t[11] = pose[5];
    // d (R X) / d r_i = w_i x (R X) with w_i = (r_i r + r x (e_i - R e_i)) / |r|^2 (Gallego & Yezzi 2015; w_i = e_i as |r| -> 0):
    // the three w_i depend on the pose only, so the nine divisions are made here, once per pose, not once per observation
    const double* v = pose;
    const double th2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        double w[3] = {i == 0 ? 1.0 : 0.0, i == 1 ? 1.0 : 0.0, i == 2 ? 1.0 : 0.0};
        if (!(th2 < DBL_EPSILON * DBL_EPSILON)) {
            const double d[3] = {(i == 0 ? 1.0 : 0.0) - R[0 + i], (i == 1 ? 1.0 : 0.0) - R[3 + i], (i == 2 ? 1.0 : 0.0) - R[6 + i]};
            const double c[3] = {v[1] * d[2] - v[2] * d[1], v[2] * d[0] - v[0] * d[2], v[0] * d[1] - v[1] * d[0]};
#pragma unroll
            for (int a = 0; a < 3; a++) w[a] = (v[i] * v[a] + c[a]) / th2;
        }
        t[12 + 3 * i] = w[0]; t[13 + 3 * i] = w[1]; t[14 + 3 * i] = w[2];
    }
    t[21] = 0;
}

__device__ __forceinline__ void cross3(const double* a, const double* b, double* o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

// a pose's table row in registers (16-byte loads); `full` = with the W block the pose Jacobian needs
struct BaPose { double R[9], t[3], W[9]; };
template <bool FULL>
__device__ __forceinline__ BaPose ba_load_pose(const double* __restrict__ tab, int j) {
    const double2* q = reinterpret_cast<const double2*>(tab + (long)j * BA_TAB);
    double2 a[11];
#pragma unroll
    for (int i = 0; i < (FULL ? 11 : 6); i++) a[i] = q[i];
    BaPose p;
    p.R[0] = a[0].x; p.R[1] = a[0].y; p.R[2] = a[1].x; p.R[3] = a[1].y; p.R[4] = a[2].x; p.R[5] = a[2].y; p.R[6] = a[3].x; p.R[7] = a[3].y; p.R[8] = a[4].x;
    p.t[0] = a[4].y; p.t[1] = a[5].x; p.t[2] = a[5].y;
    if (FULL) { p.W[0] = a[6].x; p.W[1] = a[6].y; p.W[2] = a[7].x; p.W[3] = a[7].y; p.W[4] = a[8].x; p.W[5] = a[8].y; p.W[6] = a[9].x; p.W[7] = a[9].y; p.W[8] = a[10].x; }
    return p;
}

// projection and the derivative rows d u / d Xc, d v / d Xc of one observation; Y = R X
__device__ __forceinline__ void ba_project(const BaPose& p, const double* X, const BaK& k, double proj[2], double Y[3], double J0[3], double J1[3]) {
    const double* R = p.R;
    Y[0] = R[0] * X[0] + R[1] * X[1] + R[2] * X[2]; Y[1] = R[3] * X[0] + R[4] * X[1] + R[5] * X[2]; Y[2] = R[6] * X[0] + R[7] * X[1] + R[8] * X[2];
    const double Xc[3] = {Y[0] + p.t[0], Y[1] + p.t[1], Y[2] + p.t[2]};
    const double iz = 1.0 / Xc[2];
    const double x = Xc[0] * iz, y = Xc[1] * iz;
    proj[0] = k.fx * x + k.cx;
    proj[1] = k.fy * y + k.cy;
    J0[0] = k.fx * iz; J0[1] = 0.0; J0[2] = -k.fx * x * iz;
    J1[0] = 0.0; J1[1] = k.fy * iz; J1[2] = -k.fy * y * iz;
}
// Jl[2][3] (d/d X) = J R
__device__ __forceinline__ void ba_jl(const BaPose& p, const double J0[3], const double J1[3], double Jl[6]) {
    const double* R = p.R;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        Jl[i] = J0[0] * R[i] + J0[1] * R[3 + i] + J0[2] * R[6 + i];
        Jl[3 + i] = J1[0] * R[i] + J1[1] * R[3 + i] + J1[2] * R[6 + i];
    }
}
// Jp[2][6] (d/d rvec, d/d tvec): d Y / d r_i = w_i x Y
__device__ __forceinline__ void ba_jp(const BaPose& p, const double Y[3], const double J0[3], const double J1[3], double Jp[12]) {
#pragma unroll
    for (int i = 0; i < 3; i++) {
        double dY[3];
        cross3(p.W + 3 * i, Y, dY);
        Jp[i] = J0[0] * dY[0] + J0[1] * dY[1] + J0[2] * dY[2];
        Jp[6 + i] = J1[0] * dY[0] + J1[1] * dY[1] + J1[2] * dY[2];
        Jp[3 + i] = J0[i];
        Jp[9 + i] = J1[i];
    }
}

#define BA_HPL_PITCH 19   // doubles per lane in the LDS staging of Hpl (18 + 1: 38 dwords, so 16 lanes' 8-byte writes cover all 32 banks)
#define BA_PT_PITCH 13    // doubles per lane in the LDS staging of the point terms (12 + the point id: 26 dwords, conflict free like BA_HPL_PITCH)
// One workgroup takes the observations [blk_start[blk], blk_start[blk + 1]) (at most 256).  With FUSE (the observations are
// sorted by point and the plan cut the blocks at point boundaries) the workgroup also sums the point blocks Hll / bl of its
// points: every lane stages its observation's 12 terms in LDS and the lane of a point's FIRST observation adds the run up in
// observation order -- the order, and the bits, of the separate point role, which then has nothing left to do.
template <bool FUSE>
__device__ __forceinline__ void ba_role_obs(double* sh, int blk, const int* __restrict__ blk_start, const int* __restrict__ obs_pose,
                                            const int* __restrict__ obs_point, const double* __restrict__ tab, const double* __restrict__ points,
                                            const double* __restrict__ uv, const BaK& k, double* __restrict__ r2, double* __restrict__ nrm,
                                            double* __restrict__ Hpl, double* __restrict__ Hll, double* __restrict__ bl) {
    const int tid = threadIdx.x;
    const long o0 = blk_start[blk];
    const int n = (int)(blk_start[blk + 1] - o0);
    const long o = o0 + tid;
    const bool pts = FUSE && (Hll || bl);
    double hpl[18];
    int pid = -1;
    if (tid < n) {
        const BaPose p = ba_load_pose<true>(tab, obs_pose[o]);
        pid = obs_point[o];
        const double* Xp = points + (long)pid * 3;
        const double X[3] = {Xp[0], Xp[1], Xp[2]};
        const double2 m = reinterpret_cast<const double2*>(uv)[o];
        double proj[2], Y[3], J0[3], J1[3], Jp[12], Jl[6];
        ba_project(p, X, k, proj, Y, J0, J1);
        const double r0 = proj[0] - m.x, r1 = proj[1] - m.y;
        if (r2) reinterpret_cast<double2*>(r2)[o] = double2{r0, r1};
        if (nrm) nrm[o] = sqrt(r0 * r0 + r1 * r1);
        if (Hpl || pts) ba_jl(p, J0, J1, Jl);
        if (Hpl) {
            ba_jp(p, Y, J0, J1, Jp);
#pragma unroll
            for (int a = 0; a < 6; a++)
#pragma unroll
                for (int b = 0; b < 3; b++) hpl[a * 3 + b] = Jp[a] * Jl[b] + Jp[6 + a] * Jl[3 + b];
        }
        if (pts) {
            double* t = sh + tid * BA_PT_PITCH;
#pragma unroll
            for (int a = 0; a < 3; a++) {
#pragma unroll
                for (int b = 0; b < 3; b++) t[a * 3 + b] = Jl[a] * Jl[b] + Jl[3 + a] * Jl[3 + b];
                t[9 + a] = Jl[a] * r0 + Jl[3 + a] * r1;
            }
            reinterpret_cast<int*>(t + 12)[0] = pid;
        }
    }
    if (pts) {
        __syncthreads();
        const bool head = tid < n && (tid == 0 || reinterpret_cast<const int*>(sh + (tid - 1) * BA_PT_PITCH + 12)[0] != pid);
        if (head) {
            double acc[12];
#pragma unroll
            for (int a = 0; a < 12; a++) acc[a] = 0;
            for (int j = tid; j < n && reinterpret_cast<const int*>(sh + j * BA_PT_PITCH + 12)[0] == pid; j++) {
                const double* t = sh + j * BA_PT_PITCH;
#pragma unroll
                for (int a = 0; a < 12; a++) acc[a] += t[a];
            }
            if (Hll)
#pragma unroll
                for (int a = 0; a < 9; a++) Hll[(long)pid * 9 + a] = acc[a];
            if (bl)
#pragma unroll
                for (int a = 0; a < 3; a++) bl[(long)pid * 3 + a] = acc[9 + a];
        }
        if (!Hpl) return;
        __syncthreads();     // the staging area is reused for Hpl
    }
    if (!Hpl) return;
    if (tid < n) {
        double* h = sh + tid * BA_HPL_PITCH;
#pragma unroll
        for (int a = 0; a < 18; a++) h[a] = hpl[a];
    }
    __syncthreads();
    // the workgroup's n x 18 doubles are one contiguous run of Hpl: 16-byte stores, consecutive lanes consecutive addresses
    const int n2 = n * 9;                                           // double2 elements of the run
    double2* dst = reinterpret_cast<double2*>(Hpl + o0 * 18);
    for (int e2 = tid; e2 < n2; e2 += 256) {
        const int l = e2 / 9, kk = 2 * (e2 - 9 * l);
        const double* h = sh + l * BA_HPL_PITCH + kk;
        dst[e2] = double2{h[0], h[1]};
    }
}

// sum over the 64 lanes of a wavefront in a fixed order (DPP row shifts, then the row broadcasts); lane 63 holds the total
__device__ __forceinline__ double ba_dpp_step(double v, const int ctrl, const int row_mask) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    int lo2, hi2;
    switch (ctrl) {   // the control word must be a compile-time constant
        case 0x111: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x111, 0xf, 0xf, false); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x111, 0xf, 0xf, false); break;
        case 0x112: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x112, 0xf, 0xf, false); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x112, 0xf, 0xf, false); break;
        case 0x114: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x114, 0xf, 0xf, false); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x114, 0xf, 0xf, false); break;
        case 0x118: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x118, 0xf, 0xf, false); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x118, 0xf, 0xf, false); break;
        case 0x142: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x142, 0xa, 0xf, false); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x142, 0xa, 0xf, false); break;
        default:    lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x143, 0xc, 0xf, false); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x143, 0xc, 0xf, false); break;
    }
    (void)row_mask;
    return v + __hiloint2double(hi2, lo2);      // lanes without a source add +0.0
}
__device__ __forceinline__ double ba_wave_sum(double v) {
    v = ba_dpp_step(v, 0x111, 0xf);   // row_shr:1
    v = ba_dpp_step(v, 0x112, 0xf);   // row_shr:2
    v = ba_dpp_step(v, 0x114, 0xf);   // row_shr:4
    v = ba_dpp_step(v, 0x118, 0xf);   // row_shr:8: lane 15 of every row holds the row's sum
    v = ba_dpp_step(v, 0x142, 0xa);   // row_bcast:15 into rows 1 and 3
    v = ba_dpp_step(v, 0x143, 0xc);   // row_bcast:31 into rows 2 and 3
    return v;
}

#ifndef BA_POSE_UNROLL
#define BA_POSE_UNROLL 2
#endif
#define BA_NSYM 27   // 21 unique entries of the symmetric 6x6 block + the 6 of bp
// A workgroup serves BA_PPW poses, 4 / BA_PPW wavefronts each.  The role is latency bound (two dependent rounds of gathers per
// trip), so what it costs the pass is the time its workgroups hold their slots: with one pose per workgroup the 1000 poses of
// configs[4] took every slot of the chip for the first 17 us of the pass; four per workgroup (a wavefront each: 71 -> 60 us per pass; two: 63) leave two thirds of the slots to the
// observation role from the start.
#ifndef BA_PPW
#define BA_PPW 4
#endif
__device__ __forceinline__ void ba_role_pose(double (*part)[BA_NSYM], int blk, int P, const int* __restrict__ pose_start, const int* __restrict__ pose_obs,
                                             const int* __restrict__ pose_pt, const double* __restrict__ tab, const double* __restrict__ points,
                                             const double* __restrict__ uv, const BaK& k, double* __restrict__ Hpp, double* __restrict__ bp) {
    constexpr int WPP = 4 / BA_PPW, LP = 64 * WPP;           // wavefronts and lanes per pose
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = wave / WPP, tid = (int)threadIdx.x - g * LP;   // the pose's place in the workgroup, the lane's place in the pose
    const int jr = blk * BA_PPW + g;
    const int j = min(jr, P - 1);                             // a group past the last pose runs pose P - 1 again and stores nothing
    const BaPose p = ba_load_pose<true>(tab, j);
    double acc[BA_NSYM];
#pragma unroll
    for (int i = 0; i < BA_NSYM; i++) acc[i] = 0;
    const int s = pose_start[j], e = jr < P ? pose_start[j + 1] : s;
    for (int q0 = s; q0 < e; q0 += BA_POSE_UNROLL * LP) {
        // BA_POSE_UNROLL observations per lane and trip: every index load, then every gather, is issued before the first use
        int oo[BA_POSE_UNROLL], ii[BA_POSE_UNROLL];
#pragma unroll
        for (int u = 0; u < BA_POSE_UNROLL; u++) {
            const int q = q0 + u * LP + tid;
            oo[u] = q < e ? pose_obs[q] : -1;
            ii[u] = q < e ? pose_pt[q] : 0;
        }
        double X[BA_POSE_UNROLL][3];
        double2 m[BA_POSE_UNROLL];
#pragma unroll
        for (int u = 0; u < BA_POSE_UNROLL; u++) {
            const double* Xp = points + (long)ii[u] * 3;
            X[u][0] = Xp[0]; X[u][1] = Xp[1]; X[u][2] = Xp[2];
            m[u] = reinterpret_cast<const double2*>(uv)[oo[u] < 0 ? 0 : oo[u]];
        }
#pragma unroll
        for (int u = 0; u < BA_POSE_UNROLL; u++) {
            if (oo[u] < 0) continue;
            double proj[2], Y[3], J0[3], J1[3], Jp[12];
            ba_project(p, X[u], k, proj, Y, J0, J1);
            ba_jp(p, Y, J0, J1, Jp);
            const double r0 = proj[0] - m[u].x, r1 = proj[1] - m[u].y;
            int n = 0;
#pragma unroll
            for (int a = 0; a < 6; a++) {
#pragma unroll
                for (int b = a; b < 6; b++) acc[n++] += Jp[a] * Jp[b] + Jp[6 + a] * Jp[6 + b];
            }
#pragma unroll
            for (int a = 0; a < 6; a++) acc[21 + a] += Jp[a] * r0 + Jp[6 + a] * r1;
        }
    }
#pragma unroll
    for (int i = 0; i < BA_NSYM; i++) {
        const double v = ba_wave_sum(acc[i]);
        if (lane == 63) part[wave][i] = v;
    }
    __syncthreads();
    if (tid < 42 && jr < P) {
        // entry (a, b) of the 6x6 block lives at its upper-triangle slot
        int slot;
        if (tid < 36) {
            const int a = tid / 6, b = tid - 6 * a, lo = min(a, b), hi = max(a, b);
            slot = lo * 6 - lo * (lo - 1) / 2 + (hi - lo);
        } else slot = 21 + (tid - 36);
        const int w0 = g * WPP;
        double v;
        if (WPP == 4) v = (part[w0][slot] + part[w0 + 1][slot]) + (part[w0 + 2][slot] + part[w0 + 3][slot]);
        else if (WPP == 2) v = part[w0][slot] + part[w0 + 1][slot];
        else v = part[w0][slot];
        if (tid < 36) { if (Hpp) Hpp[(long)j * 36 + tid] = v; }
        else if (bp) bp[(long)j * 6 + (tid - 36)] = v;
    }
}

__device__ __forceinline__ void ba_role_point(int blk, int N, const int* __restrict__ point_start, const int* __restrict__ point_obs,
                                              const int* __restrict__ point_pose, const double* __restrict__ tab,
                                              const double* __restrict__ points, const double* __restrict__ uv, const BaK& k,
                                              double* __restrict__ Hll, double* __restrict__ bl) {
    const int i = blk * 256 + threadIdx.x;
    if (i >= N) return;
    const double X[3] = {points[(long)i * 3], points[(long)i * 3 + 1], points[(long)i * 3 + 2]};
    double acc[12];
#pragma unroll
    for (int a = 0; a < 12; a++) acc[a] = 0;
    const int qs = point_start[i], qe = point_start[i + 1];
    for (int q0 = qs; q0 < qe; q0 += 4) {
        // four observations per trip: the index loads, then the gathers (a 128-byte table row and a uv record each), are all
        // issued before the first use, so a point with 5 observations waits for two rounds of memory latency instead of ten
        int jj[4], oo[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int q = q0 + u;
            jj[u] = q < qe ? point_pose[q] : -1;
            oo[u] = q < qe ? point_obs[q] : 0;
        }
        double2 m[4];
#pragma unroll
        for (int u = 0; u < 4; u++) m[u] = reinterpret_cast<const double2*>(uv)[oo[u]];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (jj[u] < 0) continue;
            const BaPose p = ba_load_pose<false>(tab, jj[u]);
            double proj[2], Y[3], J0[3], J1[3], Jl[6];
            ba_project(p, X, k, proj, Y, J0, J1);
            ba_jl(p, J0, J1, Jl);
            const double r0 = proj[0] - m[u].x, r1 = proj[1] - m[u].y;
#pragma unroll
            for (int a = 0; a < 3; a++) {
#pragma unroll
                for (int b = 0; b < 3; b++) acc[a * 3 + b] += Jl[a] * Jl[b] + Jl[3 + a] * Jl[3 + b];
                acc[9 + a] += Jl[a] * r0 + Jl[3 + a] * r1;
            }
        }
    }
    if (Hll)
#pragma unroll
        for (int a = 0; a < 9; a++) Hll[(long)i * 9 + a] = acc[a];
    if (bl)
#pragma unroll
        for (int a = 0; a < 3; a++) bl[(long)i * 3 + a] = acc[9 + a];
}

// ONE launch for the three reductions (three launches on three streams cost the host ~17 us each and left the pass launch
// bound): the grid is the pose workgroups, then the point workgroups, then the observation workgroups, so the gathers of the
// first two are in flight when the observation blocks start to stream (interleaving the roles evenly measured slower: 92 us
// against 76 us per pass).
struct BaArgs {
    int P, N, M, n_pose_blk, n_point_blk;
    const int *obs_pose, *obs_point, *pose_start, *pose_obs, *pose_pt, *point_start, *point_obs, *point_pose, *blk_start;
    int fuse_points;
    const double *tab, *points, *uv;
    double *r2, *nrm, *Hpp, *bp, *Hll, *bl, *Hpl;
    BaK k;
};
__global__ __launch_bounds__(256) void k_ba_pass(BaArgs a) {
    __shared__ double sh[256 * BA_HPL_PITCH];
    int b = blockIdx.x;
#ifdef BA_OBS_FIRST
    {
        const int n_obs_blk = (int)gridDim.x - a.n_pose_blk - a.n_point_blk;
        if (b < n_obs_blk) {
            if (a.fuse_points) ba_role_obs<true>(sh, b, a.blk_start, a.obs_pose, a.obs_point, a.tab, a.points, a.uv, a.k, a.r2, a.nrm, a.Hpl, a.Hll, a.bl);
            else ba_role_obs<false>(sh, b, a.blk_start, a.obs_pose, a.obs_point, a.tab, a.points, a.uv, a.k, a.r2, a.nrm, a.Hpl, nullptr, nullptr);
            return;
        }
        b -= n_obs_blk;
    }
#endif
    if (b < a.n_pose_blk) {
        ba_role_pose(reinterpret_cast<double (*)[BA_NSYM]>(sh), b, a.P, a.pose_start, a.pose_obs, a.pose_pt, a.tab, a.points, a.uv, a.k, a.Hpp, a.bp);
        return;
    }
    b -= a.n_pose_blk;
    if (b < a.n_point_blk) {
        ba_role_point(b, a.N, a.point_start, a.point_obs, a.point_pose, a.tab, a.points, a.uv, a.k, a.Hll, a.bl);
        return;
    }
    b -= a.n_point_blk;
    if (a.fuse_points) ba_role_obs<true>(sh, b, a.blk_start, a.obs_pose, a.obs_point, a.tab, a.points, a.uv, a.k, a.r2, a.nrm, a.Hpl, a.Hll, a.bl);
    else ba_role_obs<false>(sh, b, a.blk_start, a.obs_pose, a.obs_point, a.tab, a.points, a.uv, a.k, a.r2, a.nrm, a.Hpl, nullptr, nullptr);
}

// -------------------------------------------------------------------- C ABI
static void ba_free(vo_ba_plan* p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    (void)hipFree(p->d_obs_pose); (void)hipFree(p->d_obs_point); (void)hipFree(p->d_pose_start); (void)hipFree(p->d_pose_obs); (void)hipFree(p->d_pose_pt);
    (void)hipFree(p->d_point_start); (void)hipFree(p->d_point_obs); (void)hipFree(p->d_point_pose); (void)hipFree(p->d_tab); (void)hipFree(p->d_work);
    (void)hipFree(p->d_blk_start); (void)hipFree(p->d_empty);
    for (hipEvent_t e : p->ev) if (e) (void)hipEventDestroy(e);
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
    std::vector<int> ps(P + 1, 0), ls(N + 1, 0), po(M), lo(M), ppt(M), lps(M);
    for (int o = 0; o < M; o++) { ps[obs_pose[o] + 1]++; ls[obs_point[o] + 1]++; }
    for (int j = 0; j < P; j++) ps[j + 1] += ps[j];
    for (int i = 0; i < N; i++) ls[i + 1] += ls[i];
    {
        std::vector<int> pc(ps.begin(), ps.end() - 1), lc(ls.begin(), ls.end() - 1);
        for (int o = 0; o < M; o++) { po[pc[obs_pose[o]]++] = o; lo[lc[obs_point[o]]++] = o; }
    }
    // the other index of every grouped observation, so that the kernels need no second dependent index load
    for (int q = 0; q < M; q++) { ppt[q] = obs_point[po[q]]; lps[q] = obs_pose[lo[q]]; }
    // observation blocks of the main role.  Observations sorted by point (the usual layout): cut the blocks at point
    // boundaries, at most 256 observations each, so that a point's observations never straddle two workgroups and its block
    // sums can be made where the terms are computed.  Otherwise (or if one point has more than 256): plain blocks of 256 and
    // the separate point role.
    bool sorted = true;
    for (int o = 1; o < M && sorted; o++) sorted = obs_point[o - 1] <= obs_point[o];
    std::vector<int> bs{0}, empty;
    bool fuse = sorted && M > 0;
    if (fuse) {
        int o = 0;
        while (o < M && fuse) {
            int e = o;           // extend by whole points while the block stays <= 256
            while (e < M) {
                int q = e;
                while (q < M && obs_point[q] == obs_point[e]) q++;
                if (q - o > 256) break;
                e = q;
            }
            if (e == o) fuse = false;      // a single point with more than 256 observations
            else { bs.push_back(e); o = e; }
        }
    }
    if (!fuse) {
        bs.assign(1, 0);
        for (int o = 0; o < M; o += 256) bs.push_back(std::min(o + 256, M));
    } else {
        for (int i = 0; i < N; i++)
            if (ls[i + 1] == ls[i]) empty.push_back(i);
    }
    vo_ba_plan* p = new vo_ba_plan;
    p->device = device; p->P = P; p->N = N; p->M = M;
    p->fuse_points = fuse; p->n_blk = (int)bs.size() - 1; p->n_empty = (int)empty.size();
    auto up = [&](int** d, const int* h, size_t n) -> bool {
        if (hipMalloc((void**)d, std::max<size_t>(n, 1) * sizeof(int)) != hipSuccess) return false;
        return n == 0 || hipMemcpy(*d, h, n * sizeof(int), hipMemcpyHostToDevice) == hipSuccess;
    };
    bool ok = up(&p->d_obs_pose, obs_pose, M) && up(&p->d_obs_point, obs_point, M) && up(&p->d_pose_start, ps.data(), P + 1) &&
              up(&p->d_pose_obs, po.data(), M) && up(&p->d_point_start, ls.data(), N + 1) && up(&p->d_point_obs, lo.data(), M) &&
              up(&p->d_pose_pt, ppt.data(), M) && up(&p->d_point_pose, lps.data(), M) && up(&p->d_blk_start, bs.data(), bs.size()) &&
              up(&p->d_empty, empty.data(), empty.size());
    ok = ok && hipMalloc((void**)&p->d_tab, (size_t)P * BA_TAB * sizeof(double)) == hipSuccess;
    ok = ok && hipStreamCreate(&p->stream) == hipSuccess;
    if (!ok) { ba_free(p); return VO_ERR_HIP; }
    *out = p;
    return VO_OK;
}

extern "C" void vo_ba_plan_destroy(vo_ba_plan* p) { ba_free(p); }

extern "C" int vo_ba_set_kernel_timing(vo_ba_plan* p, int on) {
    if (!p) return VO_ERR_INVALID;
    if (hipSetDevice(p->device) != hipSuccess) return VO_ERR_HIP;
    if (on && !p->ev[0])
        for (int i = 0; i < 3; i++)
            if (hipEventCreate(&p->ev[i]) != hipSuccess) return VO_ERR_HIP;
    p->timing = on != 0;
    if (on) { p->t_ms[0] = p->t_ms[1] = 0; p->t_n[0] = p->t_n[1] = 0; }
    return VO_OK;
}

extern "C" int vo_ba_get_kernel_timing(vo_ba_plan* p, double ms[2], int launches[2]) {
    if (!p || !ms || !launches) return VO_ERR_INVALID;
    for (int i = 0; i < 2; i++) { ms[i] = p->t_ms[i]; launches[i] = p->t_n[i]; }
    return VO_OK;
}

extern "C" int vo_ba_pass_device(vo_ba_plan* p, const double* poses, const double* points, const double* uv, const double K[9], double* r2,
                                 double* nrm, double* Hpp, double* bp, double* Hll, double* bl, double* Hpl, void* stream) {
    if (!p || !poses || !points || !K || (p->M > 0 && !uv)) return VO_ERR_INVALID;
    if (hipSetDevice(p->device) != hipSuccess) return VO_ERR_HIP;
    hipStream_t st = stream ? (hipStream_t)stream : p->stream;
    const BaK k{K[0], K[4], K[2], K[5]};
    // 16-byte accesses on uv / r / Hpl (double2); odd-aligned buffers are refused rather than read misaligned
    if ((((uintptr_t)uv | (uintptr_t)r2 | (uintptr_t)Hpl) & 15) != 0) return VO_ERR_INVALID;
    const bool fuse = p->fuse_points && (Hll || bl);
    const int n_empty = fuse ? p->n_empty : 0;
    if (p->timing) (void)hipEventRecord(p->ev[0], st);
    hipLaunchKernelGGL(k_ba_rot, dim3((std::max(p->P, n_empty) + 255) / 256), dim3(256), 0, st, p->P, poses, p->d_tab, n_empty, p->d_empty, Hll, bl);
    if (p->timing) (void)hipEventRecord(p->ev[1], st);
    BaArgs a;
    a.P = p->P; a.N = p->N; a.M = p->M;
    a.n_pose_blk = (Hpp || bp) ? (p->P + BA_PPW - 1) / BA_PPW : 0;
    a.n_point_blk = ((Hll || bl) && !fuse) ? (p->N + 255) / 256 : 0;
    const int n_obs_blk = (p->M > 0 && (r2 || nrm || Hpl || fuse)) ? p->n_blk : 0;
    a.blk_start = p->d_blk_start; a.fuse_points = fuse ? 1 : 0;
    a.obs_pose = p->d_obs_pose; a.obs_point = p->d_obs_point; a.pose_start = p->d_pose_start; a.pose_obs = p->d_pose_obs; a.pose_pt = p->d_pose_pt;
    a.point_start = p->d_point_start; a.point_obs = p->d_point_obs; a.point_pose = p->d_point_pose;
    a.tab = p->d_tab; a.points = points; a.uv = uv;
    a.r2 = r2; a.nrm = nrm; a.Hpp = Hpp; a.bp = bp; a.Hll = Hll; a.bl = bl; a.Hpl = Hpl; a.k = k;
    const int grid = a.n_pose_blk + a.n_point_blk + n_obs_blk;
    if (grid > 0) hipLaunchKernelGGL(k_ba_pass, dim3(grid), dim3(256), 0, st, a);
    if (hipGetLastError() != hipSuccess) return VO_ERR_HIP;
    if (p->timing) {
        if (hipEventRecord(p->ev[2], st) != hipSuccess || hipEventSynchronize(p->ev[2]) != hipSuccess) return VO_ERR_HIP;
        float m0 = 0, m1 = 0;
        if (hipEventElapsedTime(&m0, p->ev[0], p->ev[1]) != hipSuccess || hipEventElapsedTime(&m1, p->ev[1], p->ev[2]) != hipSuccess) return VO_ERR_HIP;
        p->t_ms[0] += m0; p->t_n[0]++;
        if (grid > 0) { p->t_ms[1] += m1; p->t_n[1]++; }
    }
    if (!stream && hipStreamSynchronize(st) != hipSuccess) return VO_ERR_HIP;
    return VO_OK;
}

extern "C" int vo_ba_pass(vo_ba_plan* p, const double* poses, const double* points, const double* uv, const double K[9], double* r2, double* nrm,
                          double* Hpp, double* bp, double* Hll, double* bl, double* Hpl) {
    if (!p || !poses || !points || !K) return VO_ERR_INVALID;
    if (hipSetDevice(p->device) != hipSuccess) return VO_ERR_HIP;
    const size_t P = p->P, N = p->N, M = p->M;
    // layout of the staging buffer (doubles)
    auto even = [](size_t x) { return (x + 1) & ~(size_t)1; };      // 16-byte aligned sections (double2 accesses on uv, r, Hpl)
    const size_t o_pose = 0, o_pts = o_pose + 6 * P, o_uv = even(o_pts + 3 * N), o_r = o_uv + 2 * M, o_n = o_r + 2 * M, o_Hpp = even(o_n + M),
                 o_bp = o_Hpp + 36 * P, o_Hll = o_bp + 6 * P, o_bl = o_Hll + 9 * N, o_Hpl = even(o_bl + 3 * N), total = o_Hpl + 18 * M;
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
