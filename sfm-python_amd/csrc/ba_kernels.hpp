// ba_kernels.hpp -- HIP kernels (gfx950 / CDNA4, wave64) of the bundle-adjustment back end.
//
// Model (what /root/reference/sfm_lite/bundle_adjustment.py:20-42 computes per observation):
//     r_i = pi(K R(w_c) (X_p - T_c)) - uv_i ,   x = [C x (w, T) | P x (X, Y, Z)]
// Observations are stored point-major (the order Graph.pt3ds_pt2ds yields,
// /root/reference/sfm_lite/graph.py:186-191): every point owns one contiguous run.
//
// HBM layout (fp64):
//   cam_idx, pt_idx   int32  [ld]          uv, r   double2 [ld]   (ld = N rounded up to 256)
//   J                 double2 [ld/64][6][64]  6 KiB tiles of 64 observations: planes 0..2 = d r/d w (2x3
//                                          row-major, two entries per plane), planes 3..5 = d r/d X;
//                                          d r/d T = -d r/d X is rebuilt in registers (see jaddr)
//                     -> every stream moves 16 B per lane (global_load/store_dwordx4, 1 KiB per wave
//                        instruction) with ONE observation per lane
//   camtab            double [C][17]       R(9) T(3) w(3) b c   -- staged in LDS by the sweeps
//   per point         V[P][6] (upper), Vinv[P][6], gp[P][3], dp[P][3], z[P][3] ...
//   per camera        Ugc[C][27] = U upper (21) | gc (6)
//   camera-major      cm_pt int32 [ld], cm_uv double2 [ld], chunk table: the same observations ordered by
//                     camera (structure only, built once per problem) for the per-camera sums
//
// Every kernel is HBM-bound streaming/gather work (~1.4 flop/byte); no MFMA on this path.  No kernel uses
// atomics: per-point sums are reduced inside a wave over the point-major order, per-camera sums inside a
// workgroup over the camera-major order, both in a fixed order.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tr2d.hpp"

namespace sfmba {

constexpr int kCamTab = 17;          // entries of a camera-table row: R (9) T (3) w (3) b c
constexpr int kCamRow = 18;          // ... and its stride: rows are 16-byte aligned (nine 16-byte loads per row)
// Point records (see k_fill_rec) and the inverse point blocks: what the camera-major passes GATHER per observation.
//   layout 0: rec [P][6] = X Y Z | z0 z1 z2, Vinv [P][6] in an array of its own (48-byte rows straddle 64-byte sectors)
//   layout 1: both padded to 8 doubles: every gather is one 64-byte sector
//   layout 2: ONE 128-byte record per point, X Y Z z0 z1 z2 . . | Vinv(6) . . : the pass that needs both
//             (k_cam_rhs_diag) touches one cache line per observation instead of two
// Measured: at 100k points the compact layout 0 wins (its 4.8 MB tables all but fit an XCD's 4 MiB
// L2; the camera-major pass B takes 13.8 us against 15.3 / 18.1 us with layouts 1 / 2), at 1M points layouts 1 / 2 win
// pass B (201 -> 168 / 174 us) and layout 2 wins the pass that gathers both (432 -> 192 us).  Shipped: layout 0 for
// the records, plus a 128-byte record of its own for that one pass (kRhsRec below) at every size.
#ifndef SFMBA_REC_LAYOUT
#define SFMBA_REC_LAYOUT 0
#endif
constexpr int kRecLayout = SFMBA_REC_LAYOUT;
constexpr int kRec = kRecLayout == 0 ? 6 : (kRecLayout == 1 ? 8 : 16);       // doubles per point record
constexpr int kVinvRow = kRecLayout == 0 ? 6 : (kRecLayout == 1 ? 8 : 16);   // stride of the inverse point blocks
constexpr int kVinvInRec = kRecLayout == 2 ? 8 : -1;                         // offset inside the record (layout 2)
// What k_cam_rhs_diag gathers per observation, one cache line: X Y Z e0 e1 e2 . . | Vinv(6) . .  (written by k_prep,
// which produces e and Vinv of every point anyway)
constexpr int kRhsRec = 16;
constexpr int kSweepThreads = 1024;  // one workgroup per CU, 16 waves sharing one LDS camera table
constexpr int kWavesPerSweepBlock = kSweepThreads / 64;

struct KMat { double k[9]; };

// The Jacobian of one observation is stored COMPACT: d r/d T = -d r/d X (the model depends on X - T
// only), so the 2x6 camera block [d r/d w | d r/d T] is never written; the six doubles of d r/d w and
// the six of d r/d X are enough (96 B instead of 144 B per observation) and every consumer rebuilds
// the translation part in registers with three negations per row.
// Address (in doubles) of pair-plane m of observation i; planes 0..2 = d r/d w row-major
// (w0 w1 | w2 w0' | w1' w2'), planes 3..5 = d r/d X likewise.  Tiled layout: the six 16-byte pairs of
// 64 consecutive observations form one contiguous 6 KiB tile [i/64][m][i%64], so a wave reads or
// writes ONE region per batch instead of six streams that lie megabytes apart.
constexpr int kJPlanes = 6;
__device__ __forceinline__ size_t jaddr(int64_t ld, int i, int m) {
    (void)ld;
    return 2 * (((size_t)(i >> 6) * kJPlanes + m) * 64 + (i & 63));
}

// device-resident control block of the PCG (lets the host enqueue iterations without reading back)
// The PCG keeps two sets of its five camera-sized vectors and two accumulators and alternates between
// them every iteration (set = iters & 1), so that the update kernel can run on several workgroups that
// all read the old set completely while each writes its slice of the new one.
constexpr int kPcgVecs = 6;     // x r p s u (plane-major [k][C]) and u once more camera-major [C][6]
constexpr int kPcgX = 0, kPcgR = 1, kPcgP = 2, kPcgS = 3, kPcgU = 4, kPcgUcm = 5;
constexpr int kPcgUpdateBlocks = 8;
constexpr int kPcgInitDeferred = 7;   // PcgCtrl::pad of a solve started by k_pcg_init_local: gamma_0 comes with the first update

struct PcgCtrl {
    double rz;        // gamma_i = r^T M^-1 r of the current iterate
    double rz0;       // ... of the initial residual
    double tol2;      // (pcg_tol)^2
    double rz_prev;   // gamma_{i-1}
    double alpha_prev;
    int    iters;
    int    max_iters;
    int    done;      // 1 converged, 2 max_iters, 3 breakdown (pAp <= 0 or non-finite)
    int    pad;
};

// Cross-lane moves that stay in the VALU (DPP row shifts, v_readlane) instead of going through the LDS pipe
// as ds_bpermute does; row_shl:N makes lane i read lane i+N of its 16-lane row (0 when that leaves the row).
template <int CTRL>
__device__ __forceinline__ int dpp_int(int x) {        // source lane out of the row: 0
    return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ __forceinline__ double dpp_double(double x) {
    const int lo = dpp_int<CTRL>(__double2loint(x)), hi = dpp_int<CTRL>(__double2hiint(x));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_double(double x, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), src);
    return __hiloint2double(hi, lo);
}

// Sum over the wave, result on every lane: suffix sums inside the four rows, then the four row totals.
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_double<0x101>(v);
    v += dpp_double<0x102>(v);
    v += dpp_double<0x104>(v);
    v += dpp_double<0x108>(v);
    return ((readlane_double(v, 0) + readlane_double(v, 16)) + readlane_double(v, 32)) + readlane_double(v, 48);
}
__device__ __forceinline__ double quad_sum(double v) {        // sum over the four lanes of a quad, on every lane
    v += dpp_double<0xB1>(v);                                  // quad_perm [1,0,3,2]
    v += dpp_double<0x4E>(v);                                  // quad_perm [2,3,0,1]
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}

// Block-wide sum of NQ values per thread; result valid on thread 0.  `red` holds >= 16*NQ doubles.
template <int NQ>
__device__ __forceinline__ void block_sum(double (&v)[NQ], double* red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int q = 0; q < NQ; ++q) v[q] = wave_sum(v[q]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) red[w * NQ + q] = v[q];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            double s = 0.0;
            for (int k = 0; k < nw; ++k) s += red[k * NQ + q];   // fixed order
            v[q] = s;
        }
    }
}

// Block-wide sum of one value per thread, result on EVERY thread, one barrier: each wave leaves its sum in
// `slots` (an array no other reduction of the kernel uses, so nothing has to be fenced before the write) and
// every thread adds the slots in the same fixed order.
__device__ __forceinline__ double block_sum_all(double v, double* slots) {
    const int nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) slots[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    for (int k = 0; k < nw; ++k) s += slots[k];
    return s;
}

constexpr int kNQ = 9;

// Final sums of partial rows.  One wave per quantity; quantity k < first_sum is a max, else a sum;
// slice y covers the partial rows [row0[y], row0[y] + nrows[y]) and writes quantity k to
// out[slot[y][k]] (slot < 0: not written).  Fixed lane->row mapping: deterministic.
constexpr int kFinishCols = 10;      // widest partial row: k_backsub's (G12, G22, q5..q8 of the points, q5..q8 of the cameras)
struct FinishJob { int row0[2], nrows[2], slot[2][kFinishCols]; };

__device__ __forceinline__ void finish_task(const double* __restrict__ part, const FinishJob& job, int y, int k,
                                            int nq, int first_sum, double* __restrict__ out) {     // one wave
    const int lane = threadIdx.x & 63;
    const double* __restrict__ rows = part + (size_t)job.row0[y] * nq;
    const int nparts = job.nrows[y];
    double s = 0.0;
    if (k < first_sum) {
        for (int b = lane; b < nparts; b += 64) s = fmax(s, rows[(size_t)b * nq + k]);
        s = wave_max(s);
    } else {
        for (int b = lane; b < nparts; b += 64) s += rows[(size_t)b * nq + k];
        s = wave_sum(s);
    }
    const int dst = job.slot[y][k];
    if (lane == 0 && dst >= 0) out[dst] = s;
}

// A finish job that rides along with another launch instead of being a ~5 us launch of its own: the
// host appends ONE workgroup to the grid of a kernel that neither reads nor writes what the job touches,
// and that workgroup does nothing but these sums (its waves share the ny x nq reductions).
struct Piggyback {
    const double* __restrict__ part;      // null: no rider
    double* __restrict__ out;
    FinishJob job;
    int ny, nq, first_sum;
};
__device__ __forceinline__ void finish_in_block(const Piggyback& pb) {
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int t = w; t < pb.ny * pb.nq; t += nw)
        finish_task(pb.part, pb.job, t / pb.nq, t % pb.nq, pb.nq, pb.first_sum, pb.out);
}

// Segmented reduction over a wave whose keys are sorted: afterwards the first lane of every run of
// equal keys holds the sum over its run (other lanes: partial sums).  All 64 lanes must call it.
//
// Cross-lane traffic goes through DPP and v_readlane, i.e. the VALU, not through ds_bpermute: the sweeps
// already load the LDS pipe with their camera-table reads, and the 6 x (1 + 2 NV) bpermutes of a shuffle-based
// reduction came to more LDS-pipe time than everything else in those kernels together.
//   rows of 16 lanes: suffix sums by row_shl:1,2,4,8 (lane i reads lane i+n of its row); a lane adds only
//   when the key n lanes up equals its own -- keys are sorted, so everything between belongs to the run;
//   across rows: from the top row down, the first lane of the next row (by then complete) is broadcast with
//   v_readlane and added by the lanes of this row that carry its key.
// Value by value (all seven steps of v[0], then v[1], ...) with the lane masks computed once: a kernel that carries
// a software pipeline across the reduction cannot afford the registers of NV interleaved chains (k_resjac spilled
// 100 dwords per lane that way).
template <int NV>
__device__ __forceinline__ void seg_reduce_serial(double (&v)[NV], int key, int lane) {
    constexpr int kRowShl = 0x100;
    const int col = lane & 15, row = lane >> 4;
    // (the cross-lane reads first and unconditionally: inside the short-circuit of `&&` they would run under a
    // partial EXEC mask, and a DPP read of a disabled lane returns 0)
    const int n1 = dpp_int<kRowShl + 1>(key), n2 = dpp_int<kRowShl + 2>(key);
    const int n4 = dpp_int<kRowShl + 4>(key), n8 = dpp_int<kRowShl + 8>(key);
    const int f48 = __builtin_amdgcn_readlane(key, 48), f32 = __builtin_amdgcn_readlane(key, 32);
    const int f16 = __builtin_amdgcn_readlane(key, 16);
    // the masks as factors 1.0 / 0.0: "add when the key matches" is then one FMA per step instead of an add and two
    // selects (a non-finite neighbour would leak through 0 * inf; a non-finite block fails the evaluation anyway)
    const double m1 = ((col + 1 < 16) & (n1 == key)) ? 1.0 : 0.0, m2 = ((col + 2 < 16) & (n2 == key)) ? 1.0 : 0.0;
    const double m4 = ((col + 4 < 16) & (n4 == key)) ? 1.0 : 0.0, m8 = ((col + 8 < 16) & (n8 == key)) ? 1.0 : 0.0;
    const double r2 = ((row == 2) & (key == f48)) ? 1.0 : 0.0, r1 = ((row == 1) & (key == f32)) ? 1.0 : 0.0;
    const double r0 = ((row == 0) & (key == f16)) ? 1.0 : 0.0;
#pragma unroll
    for (int n = 0; n < NV; ++n) {
        double x = v[n];
        x = fma(dpp_double<kRowShl + 1>(x), m1, x);
        x = fma(dpp_double<kRowShl + 2>(x), m2, x);
        x = fma(dpp_double<kRowShl + 4>(x), m4, x);
        x = fma(dpp_double<kRowShl + 8>(x), m8, x);
        x = fma(readlane_double(x, 48), r2, x);
        x = fma(readlane_double(x, 32), r1, x);
        x = fma(readlane_double(x, 16), r0, x);
        v[n] = x;
        __builtin_amdgcn_sched_barrier(0);
    }
}

// key of lane - 1 (lane 0: 0) through DPP and v_readlane instead of ds_bpermute
__device__ __forceinline__ int lane_below(int key, int lane) {
    int prev = dpp_int<0x111>(key);                          // row_shr:1
    const int k15 = __builtin_amdgcn_readlane(key, 15), k31 = __builtin_amdgcn_readlane(key, 31);
    const int k47 = __builtin_amdgcn_readlane(key, 47);
    return lane == 16 ? k15 : (lane == 32 ? k31 : (lane == 48 ? k47 : prev));
}

// fp32-STORAGE mode (BASELINE config 5): uv, r, t1 and the compact Jacobian are kept as floats, all
// arithmetic and every accumulation stays fp64.  The Jacobian tile of 64 observations is then three
// float4 planes: (w0 w1 w2 w0') (w1' w2' X0 X1) (X2 X0' X1' X2')  -- 48 B per observation.
__device__ __forceinline__ size_t jaddr_f32(int i, int m) {       // in float4 units
    return ((size_t)(i >> 6) * 3 + m) * 64 + (i & 63);
}
// element i of an array of pairs stored as double2 (f32 = 0) or float2 (f32 = 1)
__device__ __forceinline__ double2 load_pair(const double* __restrict__ base, int f32, int i) {
    if (f32) {
        const float2 v = reinterpret_cast<const float2*>(base)[i];
        return make_double2((double)v.x, (double)v.y);
    }
    return reinterpret_cast<const double2*>(base)[i];
}
__device__ __forceinline__ void store_pair(double* __restrict__ base, int f32, int i, double a, double b) {
    if (f32) reinterpret_cast<float2*>(base)[i] = make_float2((float)a, (float)b);
    else reinterpret_cast<double2*>(base)[i] = make_double2(a, b);
}

// ---------------------------------------------------------------------------------------------
// K0: per-camera table.  R = I + a [w]x + b [w]x^2 (== scipy Rotation.from_rotvec(w).as_matrix(),
// bundle_adjustment.py:25, which the reference rebuilds per OBSERVATION), and the coefficients of the
// right Jacobian Jr = I - b [w]x + c [w]x^2 used by d r / d w.
// ---------------------------------------------------------------------------------------------
// The table buffer holds three views of the same data: rows [C][17] (K1, K2, the camera-major passes), the
// compact [C][12] = R | T that the recomputing Schur pass stages in LDS, and w, b, c plane-major [5][C] for its
// one-thread-per-camera prologue.
constexpr int kCamRT = 12, kCamWbc = 5;
__host__ __device__ constexpr size_t cam_rt_offset(int C) { return (size_t)C * kCamRow; }
__host__ __device__ constexpr size_t cam_wbc_offset(int C) { return cam_rt_offset(C) + (size_t)C * kCamRT; }
__host__ __device__ constexpr size_t cam_table_doubles(int C) { return cam_wbc_offset(C) + (size_t)C * kCamWbc; }
// the 17 entries of one camera's row: R (9), T (3), w (3), b, c  (t may be global or LDS)
__device__ __forceinline__ void cam_row_values(const double* __restrict__ prm, double* __restrict__ t) {
    const double wx = prm[0], wy = prm[1], wz = prm[2];
    const double th2 = wx * wx + wy * wy + wz * wz;
    const double th = sqrt(th2);
    double a, b, cc;
    if (th < 1e-4) {
        a = 1.0 - th2 / 6.0 + th2 * th2 / 120.0;
        b = 0.5 - th2 / 24.0 + th2 * th2 / 720.0;
    } else {
        a = sin(th) / th;
        const double h = 0.5 * th;
        const double s = sin(h) / h;
        b = 0.5 * s * s;                       // (1 - cos t)/t^2 without cancellation
    }
    if (th < 0.3) {
        cc = 1.0 / 6.0 - th2 / 120.0 + th2 * th2 / 5040.0 - th2 * th2 * th2 / 362880.0 +
             th2 * th2 * th2 * th2 / 39916800.0 - th2 * th2 * th2 * th2 * th2 / 6227020800.0;
    } else {
        cc = (th - sin(th)) / (th2 * th);
    }
    t[0] = 1.0 + b * (wx * wx - th2); t[1] = -a * wz + b * wx * wy;     t[2] = a * wy + b * wx * wz;
    t[3] = a * wz + b * wx * wy;      t[4] = 1.0 + b * (wy * wy - th2); t[5] = -a * wx + b * wy * wz;
    t[6] = -a * wy + b * wx * wz;     t[7] = a * wx + b * wy * wz;      t[8] = 1.0 + b * (wz * wz - th2);
    t[9] = prm[3]; t[10] = prm[4]; t[11] = prm[5];
    t[12] = wx; t[13] = wy; t[14] = wz;
    t[15] = b; t[16] = cc;
}
__device__ __forceinline__ void cam_table_row(const double* __restrict__ prm, double* __restrict__ tab, int C, int c) {
    double* __restrict__ t = tab + (size_t)c * kCamRow;
    t[kCamTab] = 0.0;
    cam_row_values(prm, t);
    double* __restrict__ rt = tab + cam_rt_offset(C) + (size_t)c * kCamRT;
#pragma unroll
    for (int k = 0; k < kCamRT; ++k) rt[k] = t[k];
    double* __restrict__ wbc = tab + cam_wbc_offset(C);
#pragma unroll
    for (int k = 0; k < kCamWbc; ++k) wbc[(size_t)k * C + c] = t[12 + k];
}

__global__ void k_cam_table(const double* __restrict__ xc, int C, double* __restrict__ tab) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double prm[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) prm[k] = xc[6 * (size_t)c + k];
    cam_table_row(prm, tab, C, c);
}

// The first trust-region step of an outer iteration, decided on the device so that the host does not
// have to read the model's scalars back before the trial point can be evaluated: one thread builds
// the 2-D model from the (rank-reduced) exchange scalars and solves it for the radius the host passed.
// sc[25..31] = c1, c2, predicted reduction, |step_h|, |step|, skip flag, Delta used (slots 16..24 hold the
// camera-slice sums).
__global__ void k_tr_step(double* __restrict__ sc, double Delta, const PcgCtrl* __restrict__ ctrl,
                          Piggyback pb) {
    if (pb.part != nullptr) {            // the last reduction the model needs (k_backsub's partial rows), done here
        finish_in_block(pb);
        __syncthreads();
    }
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    // The step is only meaningful when the speculatively enqueued PCG iterations sufficed.  If they did
    // not, raise the skip flag: every kernel of the trial evaluation behind this one returns at once,
    // so J, r and the normal blocks of x stay intact and the host can finish the PCG and retry.
    if (ctrl != nullptr && ctrl->done == 0) { sc[30] = 1.0; return; }
    auto q = [&](int point_slot, int k) { return sc[point_slot] + sc[16 + k]; };   // points (reduced) + cameras
    const TrModel m = tr_build_model(sc[1], sc[2], sc[3], q(8, 1), q(4, 5), q(5, 6), q(11, 4), q(6, 7), q(7, 8));
    const TrStep st = tr_solve_step(m, Delta);
    sc[25] = st.c1; sc[26] = st.c2; sc[27] = st.predicted; sc[28] = st.step_h_norm; sc[29] = st.step_norm;
    sc[30] = 0.0; sc[31] = Delta;
}

// Trial point and its camera table in one launch: x_new = x + c1 (g / si^2) + c2 p (step = D step_h,
// SCIPY trf.py:495-497).  Blocks [0, bc) take one camera per thread (six parameters, then the table
// row of the NEW parameters); the remaining blocks stream the point coordinates.
__global__ __launch_bounds__(256) void k_step_table(const double* __restrict__ x,
                                                    const double* __restrict__ sg,
                                                    const double* __restrict__ p, double c1, double c2,
                                                    const double* __restrict__ coef,
                                                    int C, int64_t n, int bc, double* __restrict__ x_new,
                                                    double* __restrict__ tab, double* __restrict__ rec_new,
                                                    const double* __restrict__ skip) {
    if (skip != nullptr && *skip != 0.0) return;   // speculative launch cancelled by k_tr_step
    if (coef != nullptr) { c1 = coef[0]; c2 = coef[1]; }       // coefficients decided by k_tr_step
    if ((int)blockIdx.x < bc) {
        const int c = blockIdx.x * blockDim.x + threadIdx.x;
        if (c >= C) return;
        double prm[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const size_t e = 6 * (size_t)c + k;
            prm[k] = x[e] + c1 * sg[e] + c2 * p[e];
            x_new[e] = prm[k];
        }
        cam_table_row(prm, tab, C, c);
        return;
    }
    const int64_t n6 = 6 * (int64_t)C;
    const int nb = gridDim.x - bc;
    for (int64_t e = n6 + (blockIdx.x - bc) * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)nb * blockDim.x) {
        const double v = x[e] + c1 * sg[e] + c2 * p[e];
        x_new[e] = v;
        const int q = (int)(e - n6), pt = q / 3;        // 32-bit: n < 2^31 (checked by set_problem)
        rec_new[(size_t)kRec * pt + (q - 3 * pt)] = v;  // coordinate slots of the point's record (see k_fill_rec)
    }
}

// Point records [P][6] = X Y Z | z0 z1 z2: what the camera-major passes gather per observation.  One 48-byte
// record is three 16-byte loads; coordinates and the per-point vector z as separate arrays of doubles were six
// 8-byte loads, and a gathered load costs the texture addresser ~64 cycles per wave whatever its width.  The
// coordinate half follows x (k_step_table, k_fill_rec); the z half is written by pass A of the Schur product (z_p)
// and by k_prep (e_p = Vinv g_p, for the reduced right-hand side).
__global__ void k_fill_rec(const double* __restrict__ pts, int P, double* __restrict__ rec) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= 3 * P) return;
    const int p = q / 3;
    rec[(size_t)kRec * p + (q - 3 * p)] = pts[q];
}

// One observation: residual and (JAC) the 2x6 / 2x3 blocks.
//   v = X - T, q = R v, p = K q, A = dpi/dp K;  dr/dX = A R;  dr/dT = -A R;
//   row k of dr/dw = -(m - b (m x w) + c ((m x w) x w)),  m = (row k of A R) x v.
template <bool JAC>
__device__ __forceinline__ void observe(const double* __restrict__ t, double X, double Y, double Z,
                                        double u, double v, const KMat& K, double& rx, double& ry,
                                        double* __restrict__ jc, double* __restrict__ jp) {
    const double vx = X - t[9], vy = Y - t[10], vz = Z - t[11];
    const double qx = t[0] * vx + t[1] * vy + t[2] * vz;
    const double qy = t[3] * vx + t[4] * vy + t[5] * vz;
    const double qz = t[6] * vx + t[7] * vy + t[8] * vz;
    const double px = K.k[0] * qx + K.k[1] * qy + K.k[2] * qz;
    const double py = K.k[3] * qx + K.k[4] * qy + K.k[5] * qz;
    const double pz = K.k[6] * qx + K.k[7] * qy + K.k[8] * qz;
    const double iz = 1.0 / pz;
    const double u0 = px * iz, u1 = py * iz;
    rx = u0 - u;
    ry = u1 - v;
    if (JAC) {
        const double wx = t[12], wy = t[13], wz = t[14], b = t[15], c = t[16];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const double pk = k == 0 ? u0 : u1;
            const double a0 = (K.k[3 * k + 0] - pk * K.k[6]) * iz;
            const double a1 = (K.k[3 * k + 1] - pk * K.k[7]) * iz;
            const double a2 = (K.k[3 * k + 2] - pk * K.k[8]) * iz;
            const double j0 = a0 * t[0] + a1 * t[3] + a2 * t[6];      // row k of A R
            const double j1 = a0 * t[1] + a1 * t[4] + a2 * t[7];
            const double j2 = a0 * t[2] + a1 * t[5] + a2 * t[8];
            jp[3 * k + 0] = j0; jp[3 * k + 1] = j1; jp[3 * k + 2] = j2;
            const double m0 = j1 * vz - j2 * vy, m1 = j2 * vx - j0 * vz, m2 = j0 * vy - j1 * vx;
            const double s0 = m1 * wz - m2 * wy, s1 = m2 * wx - m0 * wz, s2 = m0 * wy - m1 * wx;
            const double t0 = s1 * wz - s2 * wy, t1 = s2 * wx - s0 * wz, t2 = s0 * wy - s1 * wx;
            jc[6 * k + 0] = -(m0 - b * s0 + c * t0);
            jc[6 * k + 1] = -(m1 - b * s1 + c * t1);
            jc[6 * k + 2] = -(m2 - b * s2 + c * t2);
            jc[6 * k + 3] = -j0; jc[6 * k + 4] = -j1; jc[6 * k + 5] = -j2;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K1 / K1r: residual (+ Jacobian) sweep.  One lane per observation, one wavefront per 64-observation
// batch, persistent 1024-thread workgroups (one per CU) that stage the camera table in LDS (LDS_TAB;
// past 160 KiB it stays in L2) and grid-stride over the observations.  All streams are coalesced
// 16 B/lane except the two int32 index streams; points are gathered (point-major order: neighbouring
// lanes share a point).  The loop is software pipelined by hand: indices are fetched two batches
// ahead and uv/point one batch ahead, so the loads of the next batch are in flight while the 6+1
// dwordx4 stores of the current one issue (the kernel is HBM-write bound: 112 of its 136 B per
// observation are writes).  cost_part[block] = sum r^2 of the block.
// ---------------------------------------------------------------------------------------------
// 16-byte streaming store of two doubles (one global_store_dwordx4)
__device__ __forceinline__ void st16(double* __restrict__ p, double a, double b) {
    *reinterpret_cast<double2*>(p) = make_double2(a, b);
}

// Camera rows of one 64-observation tile gathered THROUGH the LDS (more cameras than the LDS holds a table of, BASELINE
// config 5): the wave requests the 64 rows its lanes need -- 64 x 144 B = nine 1-KiB pieces -- with LDS-DMA loads whose
// per-lane SOURCE address walks the rows piece by piece (lane l of piece k fetches 16-byte piece (64 k + l) mod 9 of the
// row of lane (64 k + l) / 9), so that one wave instruction touches the ~8 rows of 7-8 lanes contiguously instead of 64
// rows at a stride (nine lane-divergent 16-byte loads per observation kept the texture addresser busy for 387 us of
// K1 at 10M observations: 0.26 of the HBM roofline).  The rows land in lane order in the wave's private 9 KiB slab and
// every lane reads its own row back with nine ds_read_b128 (rows at a stride of 36 dwords: conflict-free).  No
// registers are held while the pieces fly; the request for tile i + 1 is issued as soon as tile i's rows have been read
// out of the slab.
constexpr int kRowPieces = kCamRow / 2;                       // 16-byte pieces per row
constexpr size_t kRowSlabDoubles = 64 * (size_t)kCamRow;      // per wave
// (Tried and dropped, A/B on one box: the pieces issued from inline assembly with counted waits, so that the stores of a
// tile need not be acknowledged before the next request -- while a builtin LDS-DMA load is in flight hipcc waits vmcnt(0)
// at the next use of any loaded register -- and every load of k_resjac's loop issued from inline assembly with one
// counted wait per tile.  Same durations to within 2 %: these sweeps are not waiting on their own stores.)
__device__ __forceinline__ void rows_request(const double* __restrict__ table, int cam, int lane, double* slab) {
#pragma unroll
    for (int k = 0; k < kRowPieces; ++k) {
        const int q = k * 64 + lane;
        const int row = q / kRowPieces, piece = q - kRowPieces * row;
        const int cr = __shfl(cam, row);
        __builtin_amdgcn_global_load_lds(table + (size_t)cr * kCamRow + 2 * piece,
                                         (__attribute__((address_space(3))) void*)(slab + 2 * 64 * k), 16, 0, 0);
        __builtin_amdgcn_sched_barrier(0);       // piece by piece: nine address computations in flight together spill
    }
}
// the pieces have landed (every earlier vector-memory operation of the wave has completed) ...
__device__ __forceinline__ void rows_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// ... and, before the slab is requested again, the reads of it have returned
__device__ __forceinline__ void rows_read_done() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Point blocks ride along (pb.V != null): V_p = sum Jp^T Jp (6, packed upper triangle) and g_p = sum Jp^T r (3) are
// sums over the point's run of observations, and the lanes of a wave hold 64 consecutive observations of the
// point-major order with Jp and r in registers.  The nine products are reduced over the runs inside the wave
// (seg_reduce_serial) and the first lane of every run that lies
// inside the tile stores the point's row.  A run cut by a tile boundary leaves its pieces in edge[tile][0] (the
// run that continues from the previous tile) and edge[tile][1] (the run that continues into the next one);
// point_edge_fixup, riding with the camera pass that follows, adds the pieces in tile order.  No atomics; a
// separate kernel (k_point_blocks until then, 19 us at 1M observations) re-read the indices and pixels and
// recomputed every block.  Points without observations keep the zeros set_problem wrote.
constexpr int kEdgeRow = 10;                     // 9 sums, padded to a multiple of 16 bytes
struct PointBlocksOut {
    double* __restrict__ V;                      // [P][6]; null: no block sums
    double* __restrict__ gp;                     // [P][3]
    double* __restrict__ edge;                   // [tiles][2][kEdgeRow]
};

// STORE_J = false (the J-free iteration, debug option jfree): the blocks are formed for the point sums only and not
// written -- every consumer recomputes them (k_jdot / k_backsub<.., JFREE>).
template <bool LDS_TAB, bool JAC, bool STORE_R, bool F32, bool BLOCKS = false, bool STORE_J = true>
__global__ __launch_bounds__(kSweepThreads) void k_resjac(
    const double* __restrict__ camtab, const double* __restrict__ pts, const int* __restrict__ cam_idx,
    const int* __restrict__ pt_idx, const double* __restrict__ uv, double* __restrict__ r,
    double* __restrict__ J, int N, int64_t ld, int C, KMat K,
    double* __restrict__ cost_part, const double* __restrict__ skip, PointBlocksOut pb) {
    extern __shared__ __align__(16) double smem[];
    __shared__ double red[kWavesPerSweepBlock];
    if (skip != nullptr && *skip != 0.0) return;   // speculative launch cancelled by k_tr_step
    const int stride = gridDim.x * blockDim.x;
    const int lane = threadIdx.x & 63;
    static_assert(JAC || !BLOCKS, "the point blocks are sums of Jacobian entries");
    constexpr bool blocks = BLOCKS;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    // pipeline registers: batch i (uv, X ready), batch i+stride (indices ready)
    int c0 = 0, p0 = 0, c1 = 0, p1 = 0;
    double2 uv0 = make_double2(0.0, 0.0);
    double X0 = 0.0, Y0 = 0.0, Z0 = 0.0;
    if (i < N) { c0 = cam_idx[i]; p0 = pt_idx[i]; uv0 = load_pair(uv, F32, i); }
    if (i + stride < N) { c1 = cam_idx[i + stride]; p1 = pt_idx[i + stride]; }
    if (i < N) { const double* __restrict__ Xp = pts + 3 * (size_t)p0; X0 = Xp[0]; Y0 = Xp[1]; Z0 = Xp[2]; }
    if (LDS_TAB) {                               // stage the camera table while those loads fly
        const int n2 = (C * kCamRow) >> 1;
        const double2* __restrict__ src = reinterpret_cast<const double2*>(camtab);
        double2* __restrict__ dst = reinterpret_cast<double2*>(smem);
        for (int k = threadIdx.x; k < n2; k += blockDim.x) dst[k] = src[k];
        __syncthreads();
    }
    // !LDS_TAB: the rows of each tile are gathered through the wave's LDS slab (rows_request)
    double* const slab = smem + (size_t)(threadIdx.x >> 6) * kRowSlabDoubles;
    if (!LDS_TAB && i - lane < N) rows_request(camtab, c0, lane, slab);
    double acc = 0.0;
    while (i - lane < N) {                       // wave-uniform: the segmented reduction needs all 64 lanes
        const bool on = i < N;
        const int in = i + stride, in2 = in + stride;
        double tl[kCamRow];                      // the camera row by nine 16-byte reads (LDS table, or the wave's slab)
        if (!LDS_TAB) rows_wait();
        {
            const double2* __restrict__ trow = reinterpret_cast<const double2*>(LDS_TAB ? smem + (size_t)c0 * kCamRow
                                                                                         : slab + (size_t)lane * kCamRow);
#pragma unroll
            for (int k = 0; k < kCamRow / 2; ++k) { const double2 q = trow[k]; tl[2 * k] = q.x; tl[2 * k + 1] = q.y; }
        }
        if (!LDS_TAB) {
            rows_read_done();
            if (in - lane < N) rows_request(camtab, c1, lane, slab);     // the next tile's rows
        }
        // issue the next batch's loads first
        int c2 = 0, p2 = 0;
        double2 uv1 = make_double2(0.0, 0.0);
        double X1 = 0.0, Y1 = 0.0, Z1 = 0.0;
        if (in2 < N) { c2 = cam_idx[in2]; p2 = pt_idx[in2]; }
        if (in < N) {
            uv1 = load_pair(uv, F32, in);
            const double* __restrict__ Xp = pts + 3 * (size_t)p1;
            X1 = Xp[0]; Y1 = Xp[1]; Z1 = Xp[2];
        }
        // the points either side of the tile (wave-uniform addresses): does a run continue across its boundaries?
        const int ib = i - lane;
        int pprev = -1, pnext = -2;
        if (blocks) {
            if (ib > 0) pprev = pt_idx[ib - 1];
            if (ib + 64 < N) pnext = pt_idx[ib + 64];
        }
        double jc[12], jp[6], rx, ry;
        observe<JAC>(tl, X0, Y0, Z0, uv0.x, uv0.y, K, rx, ry, jc, jp);
        if (on) {
            acc += rx * rx + ry * ry;
            if (STORE_R) {
                if (F32) store_pair(r, 1, i, rx, ry); else st16(r + 2 * (size_t)i, rx, ry);
            }
            if (JAC && STORE_J) {
                // compact form: d r/d w (jc[0..2], jc[6..8]) and d r/d X (jp); d r/d T = -jp is not stored
                if (F32) {
                    float4* __restrict__ Jf = reinterpret_cast<float4*>(J);
                    Jf[jaddr_f32(i, 0)] = make_float4((float)jc[0], (float)jc[1], (float)jc[2], (float)jc[6]);
                    Jf[jaddr_f32(i, 1)] = make_float4((float)jc[7], (float)jc[8], (float)jp[0], (float)jp[1]);
                    Jf[jaddr_f32(i, 2)] = make_float4((float)jp[2], (float)jp[3], (float)jp[4], (float)jp[5]);
                } else {
                    st16(J + jaddr(ld, i, 0), jc[0], jc[1]);
                    st16(J + jaddr(ld, i, 1), jc[2], jc[6]);
                    st16(J + jaddr(ld, i, 2), jc[7], jc[8]);
#pragma unroll
                    for (int m = 0; m < 3; ++m) st16(J + jaddr(ld, i, 3 + m), jp[2 * m], jp[2 * m + 1]);
                }
            }
        }
        if (blocks) {
            double v[9];
            v[0] = jp[0] * jp[0] + jp[3] * jp[3]; v[1] = jp[0] * jp[1] + jp[3] * jp[4];
            v[2] = jp[0] * jp[2] + jp[3] * jp[5]; v[3] = jp[1] * jp[1] + jp[4] * jp[4];
            v[4] = jp[1] * jp[2] + jp[4] * jp[5]; v[5] = jp[2] * jp[2] + jp[5] * jp[5];
            v[6] = jp[0] * rx + jp[3] * ry; v[7] = jp[1] * rx + jp[4] * ry; v[8] = jp[2] * rx + jp[5] * ry;
            const int key = on ? p0 : -1;
            if (!on) {
#pragma unroll
                for (int q = 0; q < 9; ++q) v[q] = 0.0;
            }
            seg_reduce_serial<9>(v, key, lane);                  // first lane of every run: the run's sums
            const int kp = lane_below(key, lane), k63 = __builtin_amdgcn_readlane(key, 63);
            const bool first = on && (lane == 0 || key != kp);
            const bool head = lane == 0 && key == pprev;         // continues a run of the previous tile
            const bool tail = key == k63 && k63 == pnext;        // continues into the next tile
            if (first) {
                if (!head && !tail) {
                    double* __restrict__ vr = pb.V + 6 * (size_t)key;
                    st16(vr, v[0], v[1]); st16(vr + 2, v[2], v[3]); st16(vr + 4, v[4], v[5]);
                    double* __restrict__ gr = pb.gp + 3 * (size_t)key;
                    gr[0] = v[6]; gr[1] = v[7]; gr[2] = v[8];
                } else {
                    double* __restrict__ e = pb.edge + ((size_t)(ib >> 6) * 2 + (head ? 0 : 1)) * kEdgeRow;
                    st16(e, v[0], v[1]); st16(e + 2, v[2], v[3]); st16(e + 4, v[4], v[5]); st16(e + 6, v[6], v[7]);
                    e[8] = v[8];
                }
            }
        }
        i = in;
        c0 = c1; p0 = p1; c1 = c2; p1 = p2;
        uv0 = uv1; X0 = X1; Y0 = Y1; Z0 = Z1;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) s += red[k];
        cost_part[blockIdx.x] = s;
    }
}

// The pieces of the point rows that k_resjac left in `edge` (see there): one thread per tile boundary.  The tile in
// which a cut run STARTS owns it: its piece (slot 1), then slot 0 of every following tile the run reaches, in tile
// order.
__device__ __forceinline__ void point_edge_fixup(int t, const int* __restrict__ pt_idx, int N,
                                                 const PointBlocksOut& pb) {
    const int e0 = 64 * (t + 1);                           // first observation of the next tile
    if (e0 >= N) return;
    const int key = pt_idx[e0 - 1];
    if (key != pt_idx[e0]) return;                         // no run crosses this boundary
    if (t > 0 && pt_idx[64 * t] == key && pt_idx[64 * t - 1] == key) return;      // started earlier: not the owner
    double s[9];
    const double* __restrict__ a = pb.edge + ((size_t)t * 2 + 1) * kEdgeRow;
#pragma unroll
    for (int q = 0; q < 9; ++q) s[q] = a[q];
    for (int u = t + 1;; ++u) {
        const double* __restrict__ b = pb.edge + (size_t)u * 2 * kEdgeRow;
#pragma unroll
        for (int q = 0; q < 9; ++q) s[q] += b[q];
        const int n0 = 64 * (u + 1);
        if (!(n0 < N && pt_idx[n0 - 1] == key && pt_idx[n0] == key)) break;       // the run ends inside tile u
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) pb.V[6 * (size_t)key + q] = s[q];
#pragma unroll
    for (int q = 0; q < 3; ++q) pb.gp[3 * (size_t)key + q] = s[6 + q];
}

// Unpack the Jacobian pair-planes into the row-major (N,2,6)/(N,2,3) blocks of the C-ABI (test entry).
__global__ void k_unpack_jac(const double* __restrict__ J, int N, int64_t ld, int f32,
                             double* __restrict__ jc_out, double* __restrict__ jp_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    double w[6], x[6];
    if (f32) {
        const float4* __restrict__ Jf = reinterpret_cast<const float4*>(J);
        const float4 a = Jf[jaddr_f32(i, 0)], b = Jf[jaddr_f32(i, 1)], c = Jf[jaddr_f32(i, 2)];
        w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y;
        x[0] = b.z; x[1] = b.w; x[2] = c.x; x[3] = c.y; x[4] = c.z; x[5] = c.w;
    } else {
        for (int m = 0; m < 3; ++m) {
            const double2 a = *reinterpret_cast<const double2*>(J + jaddr(ld, i, m));
            const double2 b = *reinterpret_cast<const double2*>(J + jaddr(ld, i, 3 + m));
            w[2 * m] = a.x; w[2 * m + 1] = a.y; x[2 * m] = b.x; x[2 * m + 1] = b.y;
        }
    }
    for (int row = 0; row < 2; ++row)
        for (int k = 0; k < 3; ++k) {
            jc_out[(size_t)i * 12 + 6 * row + k] = w[3 * row + k];
            jc_out[(size_t)i * 12 + 6 * row + 3 + k] = -x[3 * row + k];
            jp_out[(size_t)i * 6 + 3 * row + k] = x[3 * row + k];
        }
}

// Stored per-observation data of the point-major sweeps.
struct ObsArrays {
    const int* __restrict__ cam_idx;
    const int* __restrict__ pt_idx;
    const int* __restrict__ pt_ptr;   // [P+1] run offsets
    const double* __restrict__ J;     // six pair-planes per observation (jaddr) / three float4 planes (f32)
    int64_t ld;
    int f32;                          // fp32-storage mode: J, r, t1 hold floats
};

// jc (2x6 row-major) and jp (2x3 row-major) of observation i from the compact pair-planes
__device__ __forceinline__ void load_blocks(const ObsArrays& o, int i, double* jc, double* jp) {
    double w[6];
    if (o.f32) {
        const float4* __restrict__ Jf = reinterpret_cast<const float4*>(o.J);
        const float4 a = Jf[jaddr_f32(i, 0)], b = Jf[jaddr_f32(i, 1)], c = Jf[jaddr_f32(i, 2)];
        w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y;
        jp[0] = b.z; jp[1] = b.w; jp[2] = c.x; jp[3] = c.y; jp[4] = c.z; jp[5] = c.w;
    } else {
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const double2 a = *reinterpret_cast<const double2*>(o.J + jaddr(o.ld, i, m));
            const double2 b = *reinterpret_cast<const double2*>(o.J + jaddr(o.ld, i, 3 + m));
            w[2 * m] = a.x; w[2 * m + 1] = a.y;
            jp[2 * m] = b.x; jp[2 * m + 1] = b.y;
        }
    }
#pragma unroll
    for (int row = 0; row < 2; ++row)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            jc[6 * row + k] = w[3 * row + k];
            jc[6 * row + 3 + k] = -jp[3 * row + k];
        }
}

//
// Mailbox: the hand-off of an outer iteration.  One wave copies the 32 exchange scalars and the PCG
// control block into coherent host memory and raises a sequence number behind them; the host polls
// that word instead of waiting on an event behind two blit copies.  mbox = [0..31] scalars,
// [32..] control block, [63] sequence number.
constexpr int kMboxCtrl = 32, kMboxErr = 62, kMboxSeq = 63;
struct Mailbox {
    double* __restrict__ host;            // device-visible address of the pinned block; null: no post
    const double* __restrict__ sc;
    const PcgCtrl* __restrict__ ctrl;
    unsigned long long seq;
    const unsigned* __restrict__ err;     // error word of the direct all-reduce (null: none): travels with every post,
};                                        // so that a collective that timed out aborts the solve at the next hand-off
static_assert(kMboxCtrl + (int)((sizeof(PcgCtrl) + 7) / 8) <= kMboxErr, "mailbox layout");

__device__ __forceinline__ void post_mailbox(const Mailbox& mb) {     // one full wave
    const int lane = threadIdx.x & 63;
    if (lane < kMboxCtrl) mb.host[lane] = mb.sc[lane];
    constexpr int nc = (int)((sizeof(PcgCtrl) + 7) / 8);
    if (mb.ctrl != nullptr && lane >= kMboxCtrl && lane < kMboxCtrl + nc)
        mb.host[lane] = reinterpret_cast<const double*>(mb.ctrl)[lane - kMboxCtrl];
    if (lane == kMboxErr) mb.host[lane] = mb.err != nullptr ? (double)*mb.err : 0.0;          // (the code: p2p_timeout_text)
    __threadfence_system();                       // every lane's element is out before lane 0 raises the number
    if (lane == 0)
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(mb.host + kMboxSeq), mb.seq, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
}


// ---------------------------------------------------------------------------------------------
// Camera-major side.  Per-camera sums (U_c, g_c, the reduced right-hand side, the camera half of the
// implicit Schur product) are NOT scattered from the point-major sweeps with atomics: set_problem builds,
// once, the camera-major order of the observations (a stable counting sort of the point-major positions by
// camera; structure only) and cuts every camera's run into chunks.  One 256-thread workgroup owns one chunk:
// the camera row is wave-uniform (scalar registers), the point is gathered from the L2-resident parameter
// vector, and the Jacobian blocks of the observation are RECOMPUTED from them (~250 flop, against 96 B of
// stored blocks that exist in point-major order only: a second, camera-major copy of J would cost K1 another
// 96 B of writes per observation).  Every lane keeps its sums in registers over the chunk, the wave adds
// them with DPP, the four waves through LDS in wave order, chunks of one camera in chunk order
// (k_cam_combine): plain stores, no atomics, bitwise reproducible from run to run.
// ---------------------------------------------------------------------------------------------
struct CamMajor {
    const int4* __restrict__ chunks;   // (camera, first, end, number of chunks of that camera), ordered by camera
    const int* __restrict__ pt;        // [ld] point index of the k-th observation in camera-major order
    const double* __restrict__ uv;     // [ld] its pixel (double2; float2 in fp32-storage mode)
};
constexpr int kCamThreads = 256;
constexpr int kCamWaves = kCamThreads / 64;
constexpr int kCamUnroll = 4;
constexpr int kWaveChunkCams = 4, kWaveChunkRanges = 8;      // the XCD-aware chunk table (k_cam_schur_w, k_cam_blocks_w, k_cam_rhs_diag_w)

// NV sums over the 64 lanes of a wave at the price of ~NV + 6 exchanges instead of 6 NV: a butterfly that HALVES the
// values a lane carries at every step -- at mask m the lanes with bit m clear keep the lower half of their values, those
// with bit m set the upper half, each sends the other half to its partner lane ^ m and adds what it receives.  After the
// six steps value q's total sits in element 0 of ONE lane (returned index; -1: the lane holds none).  For the 27 sums of
// the camera-major passes: 29 kept values x (two selects, one exchange, one add) against 27 x 6 DPP steps of three
// instructions -- the reduction was a quarter of the vector instructions of k_cam_blocks (a wave there reduces 27 values
// after four observations per lane).  Fixed order: bitwise reproducible.
template <int N, int M>
__device__ __forceinline__ void wave_fold_step(double* a, int lane, int& idx, int& nreal) {
    constexpr int H = (N + 1) / 2;
    const bool up = (lane & M) != 0;
#pragma unroll
    for (int i = 0; i < H; ++i) {
        const double lo = a[i], hi = (i + H < N) ? a[i + H] : 0.0;
        const double keep = up ? hi : lo, send = up ? lo : hi;
        a[i] = keep + __shfl_xor(send, M);
    }
    if (up) { idx += H; nreal -= H; } else { nreal = nreal < H ? nreal : H; }
    if constexpr (M > 1) wave_fold_step<H, M / 2>(a, lane, idx, nreal);
}
template <int NV>
__device__ __forceinline__ int wave_fold_sum(double (&a)[NV], int lane) {
    int idx = 0, nreal = NV;
    wave_fold_step<NV, 32>(a, lane, idx, nreal);
    return nreal >= 1 ? idx : -1;
}

// sums of one workgroup: NV values per lane -> out[col] (thread col < NV holds the total afterwards)
template <int NV, int NW = kCamWaves>
__device__ __forceinline__ double cam_block_total(double (&a)[NV], double (*red)[NV]) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int q = wave_fold_sum<NV>(a, lane);
    if (q >= 0) red[w][q] = a[0];
    __syncthreads();
    double s = 0.0;
    if ((int)threadIdx.x < NV) {
#pragma unroll
        for (int k = 0; k < NW; ++k) s += red[k][threadIdx.x];            // fixed order
    }
    return s;
}

// Sharded solves, the all-reduce of a per-camera sum INSIDE the kernel that produces it (world > 1): the workgroup of
// camera c holds the camera's local sums -- pass B's six, the 27 of K3 or of the right-hand-side pass -- the moment its own
// reduction ends, so its first lanes exchange them themselves -- each stores its value into
// slot [parity][rank][k][c] of every PEER's staging buffer (remote stores over xGMI), polls the slots of the peers in
// its own buffer until they are no longer EMPTY, adds in RANK ORDER (bitwise the same total on all ranks), marks the
// slots EMPTY again and carries on with the camera's PCG bookkeeping as on a single rank.  No launch of its own
// (k_p2p_pcg: 8-10 us per PCG iteration, a third of a sharded iteration at 125k observations per rank), no grid-wide
// arrival, and NO FENCE: a value is its own flag (EMPTY is a NaN payload no computation produces), so nothing has to
// be ordered against anything -- a first version with a flag per camera behind a system-scope release / acquire pair
// in every one of the 1000-5000 workgroups ran 1.4x (cfg4 / 8) to 2.8x (cfg5 / 8) SLOWER than the separate launch:
// every such fence writes back / invalidates the L2 under the other workgroups' gathers.
// Pass B alternates two slot sets (parity = iters & 1): a rank reaches iteration i + 2 only after every peer has posted
// i + 1, i.e. has consumed (and emptied) i; between two solves lie grid-wide collectives (cost, scalars).  K3 and the
// right-hand-side pass have one slot set each: two launches of either are always separated by such a collective.  Launches behind convergence return
// before they exchange, on every rank alike.  Deadlock-free although lanes wait: a workgroup posts BEFORE it waits and
// the hardware dispatches workgroups in index order, so the lowest-indexed waiting camera of any rank always finds its
// peers' posts on the way.  The wait gives up after `timeout` ticks and raises *error, so the grid always drains.
constexpr unsigned long long kCamSlotEmpty = 0xFFF8A5A5FFF8A5A5ull;      // (both halves equal: set by a 32-bit fill)
struct CamExchange {
    double* data[16];                    // camera-slot region of every rank's staging buffer (own included):
                                         // [2][W][6][C] pass B | [W][27][C] K3 | [W][27][C] right-hand side
    int rank, world;                     // world <= 1: no exchange
    int C;
    unsigned* error;                     // local: set to 1 on timeout
    long long timeout;                   // ticks of the 100 MHz wall clock
};

// lane k (< nv) of a camera's workgroup: `out` = its local sum -> the sum over the ranks, in rank order.  base = first
// double of the slot set inside the region; slot (q, k, c) at base + (q nv + k) C + c.
// `kind` (2 pass B, 3 K3, 4 right-hand side) goes into the error word with k and c: 1 << 31 | kind << 28 | k << 23 | c.
__device__ __forceinline__ double cam_exchange_value(const CamExchange& cx, size_t base, int nv, int k, int C, int c, double out,
                                                     unsigned kind) {
    if (*cx.error != 0u) return out;                         // an earlier exchange of this solve gave up: do not wait again
    const size_t mine = base + ((size_t)cx.rank * nv + k) * C + c;
    for (int q = 0; q < cx.world; ++q)
        if (q != cx.rank) __hip_atomic_store(cx.data[q] + mine, out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    double sum = 0.0;
    const long long t0 = wall_clock64();
    for (int q = 0; q < cx.world; ++q) {                     // rank order: bitwise the same sum on every rank
        double v = out;
        if (q != cx.rank) {
            unsigned long long* slot = reinterpret_cast<unsigned long long*>(cx.data[cx.rank] + base + ((size_t)q * nv + k) * C + c);
            unsigned long long bits;
            while ((bits = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) == kCamSlotEmpty) {
                if (wall_clock64() - t0 > cx.timeout) {
                    atomicCAS(cx.error, 0u, 0x80000000u | kind << 28 | (unsigned)k << 23 | ((unsigned)c & 0x7FFFFFu));
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            __hip_atomic_store(slot, kCamSlotEmpty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);     // consumed
            v = __longlong_as_double((long long)bits);
        }
        sum = q == 0 ? v : sum + v;
    }
    return sum;
}
__host__ __device__ constexpr size_t cam_slots_pcg(int world, int C, int par) { return (size_t)par * world * 6 * C; }
__host__ __device__ constexpr size_t cam_slots_k3(int world, int C) { return (size_t)2 * world * 6 * C; }
__host__ __device__ constexpr size_t cam_slots_rhs(int world, int C) { return cam_slots_k3(world, C) + (size_t)world * 27 * C; }
__host__ __device__ constexpr size_t cam_slots_total(int world, int C) { return cam_slots_rhs(world, C) + (size_t)world * 27 * C; }

// K3: U_c = sum Jc^T Jc (21, packed upper triangle), g_c = sum Jc^T r (6) of one camera chunk.
// A camera with a single chunk stores straight into Ugc[c][27]; otherwise the chunk's 27 sums go to
// partial[chunk][27] and k_cam_combine adds them.
template <bool F32>
__global__ __launch_bounds__(kCamThreads) void k_cam_blocks(CamMajor cm, const double* __restrict__ camtab,
                                                            const double* __restrict__ rec, KMat K,
                                                            double* __restrict__ Ugc, double* __restrict__ partial,
                                                            const double* __restrict__ skip, int n_chunks,
                                                            const int* __restrict__ pt_idx, int N, PointBlocksOut pb,
                                                            Piggyback fin, Mailbox mb, CamExchange cx) {
    __shared__ double red[kCamWaves][27];
    // One more rider (fin.part != null: the FIRST workgroup of the grid): the sum of the cost partials the residual
    // launch in front of this one left, and the hand-off post behind it -- k_finish's work without its launch; the host
    // has the trial cost while this launch is still building the blocks it speculates on.
    const int bid = (int)blockIdx.x - (fin.part != nullptr ? 1 : 0);        // (the FIRST workgroup: dispatched at once)
    if (bid < 0) {
        if (!(skip != nullptr && *skip != 0.0)) { finish_in_block(fin); __syncthreads(); }
        if (mb.host != nullptr && threadIdx.x < 64) post_mailbox(mb);      // (also when the trial was cancelled)
        return;
    }
    if (skip != nullptr && *skip != 0.0) return;   // speculative launch cancelled by k_tr_step
    if (bid >= n_chunks) {                         // riders: the point rows K1's tiles cut (see k_resjac)
        point_edge_fixup((bid - n_chunks) * kCamThreads + (int)threadIdx.x, pt_idx, N, pb);
        return;
    }
    const int4 ch = cm.chunks[bid];
    double t[kCamTab];
#pragma unroll
    for (int k = 0; k < kCamTab; ++k) t[k] = camtab[(size_t)ch.x * kCamRow + k];      // wave-uniform: scalar loads
    double a[27];
#pragma unroll
    for (int q = 0; q < 27; ++q) a[q] = 0.0;
    // kU observations per lane and trip, all index loads first, then all gathers (kU = 2 here: with 27 sums in
    // registers a deeper unroll pushes the kernel past 128 VGPRs, i.e. below four workgroups per CU, and 1000
    // camera workgroups then take two rounds)
    constexpr int kU = 2;
    for (int k0 = ch.y + (int)threadIdx.x; k0 < ch.z; k0 += kCamThreads * kU) {
        int p[kU];
        double2 uv[kU];
        double X[kU][3];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int k = k0 + u * kCamThreads;
            p[u] = -1;
            uv[u] = make_double2(0.0, 0.0);
            if (k < ch.z) { p[u] = cm.pt[k]; uv[u] = load_pair(cm.uv, F32, k); }
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {                    // two 16-byte loads of the point's record: X Y | Z .
            const double2* __restrict__ rp = reinterpret_cast<const double2*>(rec + (size_t)kRec * (p[u] < 0 ? 0 : p[u]));
            const double2 xy = rp[0], zz = rp[1];
            X[u][0] = xy.x; X[u][1] = xy.y; X[u][2] = zz.x;
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            if (p[u] < 0) continue;
            double jc[12], jp[6], rx, ry;
            observe<true>(t, X[u][0], X[u][1], X[u][2], uv[u].x, uv[u].y, K, rx, ry, jc, jp);
            int n = 0;
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = i; j < 6; ++j) a[n++] += jc[i] * jc[j] + jc[6 + i] * jc[6 + j];
#pragma unroll
            for (int i = 0; i < 6; ++i) a[21 + i] += jc[i] * rx + jc[6 + i] * ry;
        }
    }
    double s = cam_block_total<27>(a, red);
    if (threadIdx.x < 27) {
        if (cx.world > 1) s = cam_exchange_value(cx, cam_slots_k3(cx.world, cx.C), 27, threadIdx.x, cx.C, ch.x, s, 3u);
        if (ch.w == 1) Ugc[(size_t)ch.x * 27 + threadIdx.x] = s;
        else partial[(size_t)bid * 27 + threadIdx.x] = s;
    }
}

// K3 over the XCD-aware chunk table (many points; see k_cam_schur_w): ONE WAVE PER CHUNK.  Workgroup 8 g + k runs on
// XCD k and its four waves take range k of cameras 4 g .. 4 g + 3, so each L2 serves an eighth of the point records; a
// wave walks its ~250 observations two per lane and trip and reduces its 27 sums with wave_fold_sum alone (no LDS, no
// barrier).  Row q of `partial` = the 27 sums of chunk q (zeros for an empty one); k_cam_combine_w<27> adds a camera's
// eight rows in range order.  (Round 3 measured the cut as a LOSS for this pass -- 208 -> 320 us at cfg5 -- when a chunk
// was a 256-thread workgroup with its block reduction and every wave paid 27 x 6 DPP steps; the halving butterfly
// and one wave per chunk turn it round.)  Riders as in k_cam_blocks.
template <bool F32>
__global__ __launch_bounds__(kCamThreads) void k_cam_blocks_w(CamMajor cm, const double* __restrict__ camtab,
                                                              const double* __restrict__ rec, KMat K,
                                                              double* __restrict__ partial,
                                                              const double* __restrict__ skip, int n_wgs,
                                                              const int* __restrict__ pt_idx, int N, PointBlocksOut pb,
                                                              Piggyback fin, Mailbox mb) {
    const int bid = (int)blockIdx.x - (fin.part != nullptr ? 1 : 0);        // (the FIRST workgroup: dispatched at once)
    if (bid < 0) {
        if (!(skip != nullptr && *skip != 0.0)) { finish_in_block(fin); __syncthreads(); }
        if (mb.host != nullptr && threadIdx.x < 64) post_mailbox(mb);      // (also when the trial was cancelled)
        return;
    }
    if (skip != nullptr && *skip != 0.0) return;   // speculative launch cancelled by k_tr_step
    if (bid >= n_wgs) {                            // riders: the point rows K1's tiles cut (see k_resjac)
        point_edge_fixup((bid - n_wgs) * kCamThreads + (int)threadIdx.x, pt_idx, N, pb);
        return;
    }
    const int lane = threadIdx.x & 63;
    const int q = __builtin_amdgcn_readfirstlane(bid * kWaveChunkCams + ((int)threadIdx.x >> 6));   // wave-uniform
    const int4 ch = cm.chunks[q];
    if (ch.x < 0) return;                                       // padding behind the last camera
    double t[kCamTab];
#pragma unroll
    for (int k = 0; k < kCamTab; ++k) t[k] = camtab[(size_t)ch.x * kCamRow + k];      // wave-uniform: scalar loads
    double a[27];
#pragma unroll
    for (int n = 0; n < 27; ++n) a[n] = 0.0;
    constexpr int kU = 2;
    for (int k0 = ch.y + lane; k0 < ch.z; k0 += 64 * kU) {
        int p[kU];
        double2 uv[kU];
        double X[kU][3];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int k = k0 + u * 64;
            p[u] = -1;
            uv[u] = make_double2(0.0, 0.0);
            if (k < ch.z) { p[u] = cm.pt[k]; uv[u] = load_pair(cm.uv, F32, k); }
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const double2* __restrict__ rp = reinterpret_cast<const double2*>(rec + (size_t)kRec * (p[u] < 0 ? 0 : p[u]));
            const double2 xy = rp[0], zz = rp[1];
            X[u][0] = xy.x; X[u][1] = xy.y; X[u][2] = zz.x;
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            if (p[u] < 0) continue;
            double jc[12], jp[6], rx, ry;
            observe<true>(t, X[u][0], X[u][1], X[u][2], uv[u].x, uv[u].y, K, rx, ry, jc, jp);
            int n = 0;
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = i; j < 6; ++j) a[n++] += jc[i] * jc[j] + jc[6 + i] * jc[6 + j];
#pragma unroll
            for (int i = 0; i < 6; ++i) a[21 + i] += jc[i] * rx + jc[6 + i] * ry;
        }
    }
    const int col = wave_fold_sum<27>(a, lane);
    if (col >= 0) partial[(size_t)q * 27 + col] = a[0];
}

// out[c * cs + col * ks] = sum of the chunk rows of camera c, in chunk order, for the cameras that have more
// than one chunk (single-chunk cameras were stored by their workgroup).  `done`: PCG finished, nothing new.
__global__ __launch_bounds__(256) void k_cam_combine(const int* __restrict__ chunk_ptr, const double* __restrict__ partial,
                                                     int C, int ncols, double* __restrict__ out, int cs, int ks,
                                                     const double* __restrict__ skip, const int* __restrict__ done) {
    if (skip != nullptr && *skip != 0.0) return;
    if (done != nullptr && *done != 0) return;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= C * ncols) return;
    const int c = e / ncols, col = e - c * ncols;
    const int j0 = chunk_ptr[c], j1 = chunk_ptr[c + 1];
    if (j1 - j0 <= 1) return;
    double s = 0.0;
    for (int j = j0; j < j1; ++j) s += partial[(size_t)j * ncols + col];
    out[(size_t)c * cs + (size_t)col * ks] = s;
}

// camera-major copies of the point index and the pixel, gathered on the device from the point-major arrays
// through the permutation set_problem uploads (4 B per observation cross PCIe instead of 20)
__global__ void k_build_cam_major(const int* __restrict__ perm, const int* __restrict__ pt_idx,
                                  const double* __restrict__ uv, int f32, int N, int* __restrict__ cm_pt,
                                  double* __restrict__ cm_uv) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N) return;
    const int i = perm[k];
    cm_pt[k] = pt_idx[i];
    if (f32) reinterpret_cast<float2*>(cm_uv)[k] = reinterpret_cast<const float2*>(uv)[i];
    else reinterpret_cast<double2*>(cm_uv)[k] = reinterpret_cast<const double2*>(uv)[i];
}

// The observation arrays cross PCIe PACKED when they can (every call of the reference is a new problem, sfm.py:59-71, and at
// a million observations the 24 MB of int32 indices and fp64 pixels were the longest single item of a call): camera
// indices as uint16 (fewer than 65536 cameras), pixels as int16 pairs (the reference's pixels are integers, graph.py:112-113;
// checked value by value on the host), and no point indices at all -- they follow from the run offsets.  6 bytes per
// observation instead of 24; expanded here into the arrays the kernels read.
__global__ __launch_bounds__(256) void k_unpack_obs(const unsigned short* __restrict__ ci16, const short2* __restrict__ uv16,
                                                    int from, int to, int f32, int* __restrict__ cam_idx, double* __restrict__ uv) {
    const int k = from + blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= to) return;
    cam_idx[k] = (int)ci16[k];
    const short2 q = uv16[k];
    if (f32) reinterpret_cast<float2*>(uv)[k] = make_float2((float)q.x, (float)q.y);
    else reinterpret_cast<double2*>(uv)[k] = make_double2((double)q.x, (double)q.y);
}
// pt_idx[k] = p for k in [pt_ptr[p], pt_ptr[p + 1]); the padding behind the last observation is 0 (one wave per 64 points)
__global__ __launch_bounds__(256) void k_expand_pt_idx(const int* __restrict__ pt_ptr, int P, int N, int ld, int* __restrict__ pt_idx) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < P) {
        const int b = pt_ptr[p], e = pt_ptr[p + 1];
        for (int k = b; k < e; ++k) pt_idx[k] = p;
    } else if (p == P) {
        for (int k = N; k < ld; ++k) pt_idx[k] = 0;
    }
}

// The camera-major order built ON THE DEVICE (the reference hands a new problem to every call, sfm.py:59-71, and the
// host's counting sort + the upload of its permutation were 0.4-0.5 ms of a 3.6 ms call at a million observations): a
// stable counting sort of the point-major positions by camera in three launches.  The observations are cut into
// kSortSlices contiguous slices, one WAVE each:
//   k_cam_hist     per-slice histogram over the cameras (LDS counters)          -> hist[slice][C]
//   k_cam_offsets  one thread per camera: first destination of every slice's observations of that camera
//                  = cam_ptr[c] + sum of the earlier slices' counts (in place)
//   k_cam_scatter  every wave walks its slice in order, 64 observations at a time; a lane's destination = the running
//                  offset of its camera (LDS) + the number of LOWER lanes with the same camera, found without a sort:
//                  one ballot per key bit gives every lane the mask of its equals (ceil(log2 C) ballots per batch)
// Same order as the host's stable sort, entry for entry (a test compares the tables), so every per-camera sum keeps
// its summation order.  Cameras held still (`fixed`, may be null) contribute no entries.
constexpr int kSortSlices = 512;
__global__ __launch_bounds__(64) void k_cam_hist(const int* __restrict__ cam_idx, const unsigned char* __restrict__ fixed,
                                                 int N, int C, int per, int* __restrict__ hist) {
    extern __shared__ int cnt[];
    for (int c = threadIdx.x; c < C; c += 64) cnt[c] = 0;
    __syncthreads();
    const int b0 = blockIdx.x * per, b1 = min(N, b0 + per);
    for (int k0 = b0 + (int)threadIdx.x; k0 < b1; k0 += 64 * 8) {      // eight index loads in flight per lane
        int c[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int k = k0 + 64 * u; c[u] = k < b1 ? cam_idx[k] : -1; }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (c[u] >= 0 && (fixed == nullptr || fixed[c[u]] == 0)) atomicAdd(&cnt[c[u]], 1);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 64) hist[(size_t)blockIdx.x * C + c] = cnt[c];
}
// (off is a second array: written in place the loads of the scan could not be issued ahead of its stores)
__global__ __launch_bounds__(256) void k_cam_offsets(const int* __restrict__ hist, const int* __restrict__ cam_ptr, int B, int C,
                                                     int* __restrict__ off) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    int run = cam_ptr[c];
    for (int b0 = 0; b0 < B; b0 += 16) {
        int n[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) n[u] = b0 + u < B ? hist[(size_t)(b0 + u) * C + c] : 0;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (b0 + u < B) off[(size_t)(b0 + u) * C + c] = run;
            run += n[u];
        }
    }
}
__global__ __launch_bounds__(64) void k_cam_scatter(const int* __restrict__ cam_idx, const unsigned char* __restrict__ fixed,
                                                    const int* __restrict__ pt_idx, const double* __restrict__ uv, int f32,
                                                    int N, int C, int per, int key_bits, const int* __restrict__ off,
                                                    int* __restrict__ cm_pt, double* __restrict__ cm_uv) {
    extern __shared__ int cur[];
    for (int c = threadIdx.x; c < C; c += 64) cur[c] = off[(size_t)blockIdx.x * C + c];
    __syncthreads();
    const int lane = threadIdx.x;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    const int b0 = blockIdx.x * per, b1 = min(N, b0 + per);
    // the walk is a chain (a batch's destinations depend on the offsets the batch before it left): everything that does
    // NOT depend on it -- the batch's camera, point index and pixel -- is requested one batch ahead
    auto fetch = [&](int k, int& c, int& p, double2& q) {
        c = -1; p = 0; q = make_double2(0.0, 0.0);
        if (k < b1) {
            c = cam_idx[k];
            p = pt_idx[k];
            if (f32) { const float2 t = reinterpret_cast<const float2*>(uv)[k]; q = make_double2((double)t.x, (double)t.y); }
            else q = reinterpret_cast<const double2*>(uv)[k];
        }
    };
    int cn, pn;
    double2 qn;
    fetch(b0 + lane, cn, pn, qn);
    for (int k0 = b0; k0 < b1; k0 += 64) {
        int c = cn;
        const int p = pn;
        const double2 q = qn;
        fetch(k0 + 64 + lane, cn, pn, qn);
        if (c >= 0 && fixed != nullptr && fixed[c] != 0) c = -1;
        const bool act = c >= 0;
        unsigned long long eq = __ballot(act);                 // lanes with the same camera as this one
        for (int bit = 0; bit < key_bits; ++bit) {
            const unsigned long long m = __ballot(act && ((c >> bit) & 1));
            eq &= ((c >> bit) & 1) ? m : ~m;
        }
        if (act) {
            const int dst = cur[c] + __popcll(eq & lt);
            cm_pt[dst] = p;
            if (f32) reinterpret_cast<float2*>(cm_uv)[dst] = make_float2((float)q.x, (float)q.y);
            else reinterpret_cast<double2*>(cm_uv)[dst] = q;
            if ((eq >> lane) == 1ull) cur[c] += __popcll(eq);   // the highest lane of the group advances the offset
        }
        // One wave: its LDS operations execute in program order, so the next batch's reads of `cur` see this batch's
        // updates without waiting for anything -- a __syncthreads() here also drained the stores and the prefetch
        // (s_waitcnt vmcnt(0)): 1.9 us per batch instead of 0.2.  The wave barrier only stops the compiler reordering.
        __builtin_amdgcn_wave_barrier();
    }
}
// The XCD-aware chunk table of pass B (k_cam_schur_w) from the camera-major point indices: row (8 g + k) 4 + j = the
// piece of camera 4 g + j inside point range k (its list is ascending in the point index: two binary searches).
__global__ __launch_bounds__(256) void k_xcd_chunks(const int* __restrict__ cam_ptr, const int* __restrict__ cm_pt, int C, int P,
                                                    int4* __restrict__ chunks) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int groups = (C + 3) / 4;
    if (e >= groups * 32) return;
    const int j = e & 3, k = (e >> 2) & 7, g = e >> 5;
    const int c = 4 * g + j;
    if (c >= C) { chunks[e] = make_int4(-1, 0, 0, 8); return; }
    const int b = cam_ptr[c], en = cam_ptr[c + 1];
    auto bound = [&](int kk) {                                  // first entry of the list whose point lies in range >= kk
        if (kk <= 0) return b;
        if (kk >= 8) return en;
        const int p_hi = (int)(((long long)P * kk) / 8);
        int lo = b, hi = en;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (cm_pt[mid] < p_hi) lo = mid + 1; else hi = mid; }
        return lo;
    };
    chunks[e] = make_int4(c, bound(k), bound(k + 1), 8);
}

// ---------------------------------------------------------------------------------------------
// Point-major sweep skeleton.  Each WAVE owns an observation range cut at point boundaries
// (host-built) and walks it in precomputed steps: steps[s] = (first observation, count); count <= 64 is a
// batch that ends on a point boundary, count > 64 a single point with that many observations.
// wsteps[wave] = (first step, number of steps).  With the step list known, the index loads of step s+1 are
// issued while step s computes.  Per-point sums are reduced inside the wave (seg_reduce_serial); no run ever spans
// two waves.
// ---------------------------------------------------------------------------------------------
struct StepTable {
    const int2* __restrict__ wsteps;
    const int2* __restrict__ steps;
    int n_waves;
};

// Reductions over the parameter vector are taken separately over the camera slice [0, 6C) (replicated
// on every rank) and the point slice [6C, n) (local to the shard, summed over ranks): blocks
// [0, bc) of the grid cover the cameras, blocks [bc, grid) the points; one partial row per block.
//   q0 = max|g|   q1 = sum (g/si)^2   q2 = sum (x si)^2   q3 = sum x^2   q4 = sum (g/si^2)^2
//   q5 = sum g p  q6 = sum (p si)^2   q7 = sum (g/si^2) p q8 = sum p^2
__device__ __forceinline__ void write_partials(double (&q)[kNQ], double* __restrict__ part) {
    __shared__ double red[4 * kNQ];
    __shared__ double mred[4];
    const double m = wave_max(q[0]);
    q[0] = 0.0;
    if ((threadIdx.x & 63) == 0) mred[threadIdx.x >> 6] = m;
    block_sum<kNQ>(q, red);
    if (threadIdx.x == 0) {
        double mm = 0.0;
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) mm = fmax(mm, mred[k]);
        part[(size_t)blockIdx.x * kNQ + 0] = mm;
        for (int k = 1; k < kNQ; ++k) part[(size_t)blockIdx.x * kNQ + k] = q[k];
    }
}

// Column scale of x_scale='jac' (SCIPY common.py:598-610): si = |J col|_2 = sqrt(diag(J^T J)),
// zeros -> 1 on the first call, running max afterwards; g = J^T f gathered from the blocks;
// sg = g / si^2 (= D^2 g); and, in the same pass, the partial sums q0..q4 of the new iterate.
// Blocks [0, bc): the camera slice, one element per thread and trip.  Blocks [bc, grid): the point slice, one POINT per
// thread and trip with every load of the trip issued before the first use -- V, g_p and x were written by other XCDs a
// moment ago, so each dependent round trip is an L2 miss (the element-wise loop with its 64-bit divisions took 11.5 us
// in the solve at 306k parameters where its traffic is worth 3).
template <int kScalePts>                         // points per thread and trip (2 from 64k points on, else 1)
__global__ __launch_bounds__(256) void k_update_scale(const double* __restrict__ Ugc,
                                                      const double* __restrict__ V,
                                                      const double* __restrict__ gp,
                                                      const double* __restrict__ x, int C, int P,
                                                      int first, int bc, double* __restrict__ si,
                                                      double* __restrict__ g, double* __restrict__ sg,
                                                      double* __restrict__ part) {
    const int n6 = 6 * C;
    double q[kNQ];
#pragma unroll
    for (int k = 0; k < kNQ; ++k) q[k] = 0.0;
    auto element = [&](double d, double ge, double s_old, double xe, double& s, double& sge) {
        s = sqrt(d);
        if (first) { if (s == 0.0) s = 1.0; } else { s = fmax(s, s_old); }
        sge = ge / (s * s);
        q[0] = fmax(q[0], fabs(ge));
        q[1] += (ge / s) * (ge / s);
        q[2] += (xe * s) * (xe * s);
        q[3] += xe * xe;
        q[4] += sge * sge;
    };
    if ((int)blockIdx.x < bc) {
        const int diagU[6] = {0, 6, 11, 15, 18, 20};
        for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n6; e += bc * blockDim.x) {
            const int c = e / 6, k = e - 6 * c;
            const double d = Ugc[(size_t)c * 27 + diagU[k]], ge = Ugc[(size_t)c * 27 + 21 + k];
            double s, sge;
            element(d, ge, first ? 0.0 : si[e], x[e], s, sge);
            si[e] = s; g[e] = ge; sg[e] = sge;
        }
    } else {
        const int nb = (int)gridDim.x - bc, b = (int)blockIdx.x - bc;
        const double* __restrict__ xp = x + n6;
        double* __restrict__ sip = si + n6;
        double* __restrict__ gpo = g + n6;
        double* __restrict__ sgp = sg + n6;
        for (int p0 = (b * kScalePts) * (int)blockDim.x + (int)threadIdx.x; p0 < P; p0 += nb * kScalePts * (int)blockDim.x) {
            double d[kScalePts][3], ge[kScalePts][3], so[kScalePts][3], xe[kScalePts][3];
#pragma unroll
            for (int u = 0; u < kScalePts; ++u) {
                const int p = p0 + u * (int)blockDim.x;
                const size_t pp = (size_t)(p < P ? p : 0);
                d[u][0] = V[pp * 6 + 0]; d[u][1] = V[pp * 6 + 3]; d[u][2] = V[pp * 6 + 5];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    ge[u][k] = gp[pp * 3 + k];
                    xe[u][k] = xp[pp * 3 + k];
                    so[u][k] = first ? 0.0 : sip[pp * 3 + k];
                }
            }
#pragma unroll
            for (int u = 0; u < kScalePts; ++u) {
                const int p = p0 + u * (int)blockDim.x;
                if (p >= P) continue;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    double s, sge;
                    element(d[u][k], ge[u][k], so[u][k], xe[u][k], s, sge);
                    sip[(size_t)p * 3 + k] = s; gpo[(size_t)p * 3 + k] = ge[u][k]; sgp[(size_t)p * 3 + k] = sge;
                }
            }
        }
    }
    write_partials(q, part);
}


__global__ void k_post(Mailbox mb) { post_mailbox(mb); }

__global__ void k_finish(const double* __restrict__ part, FinishJob job, int nq, int first_sum,
                         double* __restrict__ out, const double* __restrict__ skip, Mailbox mb) {
    if (skip != nullptr && *skip != 0.0) {         // speculative launch cancelled by k_tr_step
        if (mb.host != nullptr && blockIdx.x == 0 && threadIdx.x < 64) post_mailbox(mb);
        return;
    }
    const int k = threadIdx.x >> 6;
    if (k < nq) finish_task(part, job, (int)blockIdx.x, k, nq, first_sum, out);
    if (mb.host != nullptr) {                      // single block (the host launches it that way)
        __syncthreads();
        if (threadIdx.x < 64) post_mailbox(mb);
    }
}

// t1_i = J (D^2 g) per observation and sum |t1|^2 (the quadratic of the 1-D Cauchy problem,
// SCIPY trf.py:471-475 / common.py:251-299).  Camera slice of D^2 g staged in LDS.
// The J-FREE iteration (debug option jfree, an A/B measurement: DESIGN.md): the consumers of the stored Jacobian, k_jdot
// and k_backsub, recompute an observation's blocks from the camera table (staged in LDS, where the camera vector
// otherwise sits; the vector is then read from L2) and its point, as K1 itself forms them -- same function, same bits in
// fp64 storage -- and K1 no longer writes them (STORE_J = false): 96 of its 136 bytes per observation.
struct Recompute {
    const double* __restrict__ camtab;   // [C][kCamRow]
    const double* __restrict__ pts;      // [P][3]
    KMat K;
};
__device__ __forceinline__ void stage_cam_table(const double* __restrict__ camtab, int C, double* __restrict__ smem) {
    const int n2 = (C * kCamRow) >> 1;
    const double2* __restrict__ src = reinterpret_cast<const double2*>(camtab);
    double2* __restrict__ dst = reinterpret_cast<double2*>(smem);
    for (int k = threadIdx.x; k < n2; k += blockDim.x) dst[k] = src[k];
    __syncthreads();
}
__device__ __forceinline__ void recompute_blocks(const Recompute& rc, const double* __restrict__ tab_lds, int c, int p,
                                                 double* jc, double* jp) {
    double tl[kCamRow];
    const double2* __restrict__ trow = reinterpret_cast<const double2*>(tab_lds + (size_t)c * kCamRow);
#pragma unroll
    for (int k = 0; k < kCamRow / 2; ++k) { const double2 q = trow[k]; tl[2 * k] = q.x; tl[2 * k + 1] = q.y; }
    const double* __restrict__ Xp = rc.pts + 3 * (size_t)p;
    double rx, ry;
    observe<true>(tl, Xp[0], Xp[1], Xp[2], 0.0, 0.0, rc.K, rx, ry, jc, jp);
}

template <bool LDS_VEC, bool JFREE = false>
__global__ __launch_bounds__(kSweepThreads) void k_jdot(ObsArrays o, const double* __restrict__ sgc,
                                                        const double* __restrict__ sgp, int N, int C,
                                                        double* __restrict__ t1,
                                                        double* __restrict__ part, Piggyback pb, Recompute rc) {
    static_assert(!(LDS_VEC && JFREE), "J-free: the LDS holds the camera table, the vector comes from L2");
    extern __shared__ __align__(16) double smem[];
    __shared__ double red[kWavesPerSweepBlock];
    const int nwork = pb.part != nullptr ? (int)gridDim.x - 1 : (int)gridDim.x;
    if ((int)blockIdx.x == nwork) { finish_in_block(pb); return; }      // rider block: another kernel's sums
    if (LDS_VEC) {
        for (int i = threadIdx.x; i < 6 * C; i += blockDim.x) smem[i] = sgc[i];
        __syncthreads();
    }
    if (JFREE) stage_cam_table(rc.camtab, C, smem);
    const double* __restrict__ vc = LDS_VEC ? smem : sgc;
    double acc = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += nwork * blockDim.x) {
        double jc[12], jp[6];
        if (JFREE) recompute_blocks(rc, smem, o.cam_idx[i], o.pt_idx[i], jc, jp);
        else load_blocks(o, i, jc, jp);
        const double2* a2 = reinterpret_cast<const double2*>(vc + 6 * o.cam_idx[i]);    // 16-byte aligned rows
        const double2 a01 = a2[0], a23 = a2[1], a45 = a2[2];
        const double a[6] = {a01.x, a01.y, a23.x, a23.y, a45.x, a45.y};
        const double* b = sgp + 3 * (size_t)o.pt_idx[i];
        double t0 = 0.0, t1v = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) { t0 += jc[k] * a[k]; t1v += jc[6 + k] * a[k]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) { t0 += jp[k] * b[k]; t1v += jp[3 + k] * b[k]; }
        store_pair(t1, o.f32, i, t0, t1v);
        acc += t0 * t0 + t1v * t1v;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) s += red[k];
        part[blockIdx.x] = s;
    }
}

__device__ __forceinline__ void chol3_inverse(const double* a /*upper 6*/, double* inv /*upper 6*/) {
    // A = L L^T, inv = L^-T L^-1
    const double l00 = sqrt(a[0]);
    const double l10 = a[1] / l00, l20 = a[2] / l00;
    const double l11 = sqrt(a[3] - l10 * l10);
    const double l21 = (a[4] - l20 * l10) / l11;
    const double l22 = sqrt(a[5] - l20 * l20 - l21 * l21);
    const double m00 = 1.0 / l00, m11 = 1.0 / l11, m22 = 1.0 / l22;      // M = L^-1 (lower)
    const double m10 = -l10 * m00 * m11;
    const double m21 = -l21 * m11 * m22;
    const double m20 = -(l20 * m00 + l21 * m10) * m22;
    inv[0] = m00 * m00 + m10 * m10 + m20 * m20;
    inv[1] = m10 * m11 + m20 * m21;
    inv[2] = m20 * m22;
    inv[3] = m11 * m11 + m21 * m21;
    inv[4] = m21 * m22;
    inv[5] = m22 * m22;
}

// Regularisation of the damped Gauss-Newton step from the 1-D Cauchy problem along -g_h
// (SCIPY trf.py:471-475, common.py:251-322), evaluated from the exchange scalars by whoever needs it.
__device__ __forceinline__ double reg_from_a11(double a11, double G11, double Delta, double reg_min) {
    const double a = 0.5 * G11, b = -a11;           // G11 = |J_h g_h|^2, a11 = |g_h|^2
    double reg = reg_min;
    if (a11 > 0.0) {
        const double to_tr = Delta / sqrt(a11);
        double best = 0.0;
        const double y_ub = to_tr * (a * to_tr + b);
        if (y_ub < best) best = y_ub;
        if (a != 0.0) {
            const double ext = -0.5 * b / a;
            if (ext > 0.0 && ext < to_tr) { const double y = ext * (a * ext + b); if (y < best) best = y; }
        }
        reg = fmax(-best / (Delta * Delta), reg_min);
    }
    return reg;
}
__device__ __forceinline__ double reg_from_scalars(const double* __restrict__ sc, double G11, double Delta,
                                                   double reg_min) {
    return reg_from_a11(sc[8] + sc[16 + 1], G11, Delta, reg_min);      // |g_h|^2: point slice (summed over ranks) + cameras
}

__device__ __forceinline__ void point_prep_one(const double* __restrict__ V, const double* __restrict__ gp,
                                               const double* __restrict__ sip, const double* __restrict__ dp_extra,
                                               int p, double reg, double* __restrict__ Vinv,
                                               double* __restrict__ e, const double* __restrict__ pts = nullptr,
                                               double* __restrict__ rhsrec = nullptr) {
    double a[6], inv[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) a[k] = V[(size_t)p * 6 + k];
    if (dp_extra) {                       // explicit diagonal (test entry sfmba_schur_matvec)
        a[0] += dp_extra[3 * (size_t)p]; a[3] += dp_extra[3 * (size_t)p + 1]; a[5] += dp_extra[3 * (size_t)p + 2];
    } else {
        const double s0 = sip[3 * (size_t)p], s1 = sip[3 * (size_t)p + 1], s2 = sip[3 * (size_t)p + 2];
        a[0] += reg * s0 * s0; a[3] += reg * s1 * s1; a[5] += reg * s2 * s2;
    }
    chol3_inverse(a, inv);
#pragma unroll
    for (int k = 0; k < 6; ++k) Vinv[(size_t)p * kVinvRow + k] = inv[k];
    if (e) {                              // e points at the z slot of the point records: stride kRec
        const double g0 = gp[3 * (size_t)p], g1 = gp[3 * (size_t)p + 1], g2 = gp[3 * (size_t)p + 2];
        const double e0 = inv[0] * g0 + inv[1] * g1 + inv[2] * g2;
        const double e1 = inv[1] * g0 + inv[3] * g1 + inv[4] * g2;
        const double e2 = inv[2] * g0 + inv[4] * g1 + inv[5] * g2;
        e[(size_t)kRec * p + 0] = e0; e[(size_t)kRec * p + 1] = e1; e[(size_t)kRec * p + 2] = e2;
        if (rhsrec != nullptr) {
            double* __restrict__ rr = rhsrec + (size_t)kRhsRec * p;
            st16(rr, pts[3 * (size_t)p], pts[3 * (size_t)p + 1]); st16(rr + 2, pts[3 * (size_t)p + 2], e0); st16(rr + 4, e1, e2);
            st16(rr + 8, inv[0], inv[1]); st16(rr + 10, inv[2], inv[3]); st16(rr + 12, inv[4], inv[5]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// MIXED-PRECISION Schur product (BASELINE config 5: "mixed-precision PCG with fp64 accumulation"; sfmba_set_precision).
// The product S u inside the PCG only has to be good to the forcing term (1e-2 ... 0.1), so its OPERANDS are kept in
// fp32 and everything computed from them -- blocks, dot products, every sum -- in fp64:
//   rt32   [C][12] float   R | T - o           per parameter vector (k_prep)
//   rec32  [P][8]  float   X - o | z0 z1 z2 . .   one 32-byte record per point: coordinates per parameter vector
//                          (k_prep), z by pass A of every product -- what pass B gathers per observation (one sector
//                          instead of a 48-byte record that straddles two)
//   table  [C][20] float   R | T - o | a' | u_T . .   pass A's camera rows, 80 bytes instead of 144: in LDS (<= 1100
//                          cameras: half the staging traffic and half the LDS reads per observation) or in global
//                          memory behind five LDS-DMA pieces per row instead of nine
// o = an origin near the points (mean of a sample of x0's points, fixed for the solve): coordinates are rounded RELATIVE
// to it, so a scene far from the coordinate origin keeps its fp32 digits for X - T.  Both passes see the same rounded
// R, T, X, a', u_T, hence blocks of ONE Jacobian; what is not symmetric to the last bit is the rounding of z and a'
// themselves (fp32-class, 1e-7 relative).  Gradient, cost, normal blocks, preconditioner, right-hand side, the PCG's
// vectors and scalars and the back-substitution stay fp64 as before.
// ---------------------------------------------------------------------------------------------
constexpr int kRt32 = 12, kRec32 = 8, kRc32Row = 20;
constexpr int kRow32Pieces = kRc32Row / 4;                     // 16-byte pieces per row
constexpr size_t kRow32SlabFloats = 64 * (size_t)kRc32Row;     // per wave
struct Origin { double x, y, z; };
struct MixedPrep {                     // rt32 == null: not the mixed mode
    const double* __restrict__ camtab;
    float* __restrict__ rt32;
    float* __restrict__ rec32;
    double* __restrict__ rtd;          // [C][12]: the values of rt32 as doubles (pass B reads its camera's row with scalar
    Origin o;                          // loads: a conversion there would move twelve doubles from SGPRs to VGPRs)
};
__device__ __forceinline__ void mixed_prep_camera(const MixedPrep& mx, int C, int c) {
    const double* __restrict__ rt = mx.camtab + cam_rt_offset(C) + (size_t)kCamRT * c;
    float4* __restrict__ out = reinterpret_cast<float4*>(mx.rt32 + (size_t)kRt32 * c);
    out[0] = make_float4((float)rt[0], (float)rt[1], (float)rt[2], (float)rt[3]);
    out[1] = make_float4((float)rt[4], (float)rt[5], (float)rt[6], (float)rt[7]);
    out[2] = make_float4((float)rt[8], (float)(rt[9] - mx.o.x), (float)(rt[10] - mx.o.y), (float)(rt[11] - mx.o.z));
    double* __restrict__ d = mx.rtd + (size_t)kRt32 * c;
    const float4 a = out[0], b = out[1], e = out[2];
    d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w; d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
    d[8] = e.x; d[9] = e.y; d[10] = e.z; d[11] = e.w;
}
__device__ __forceinline__ void mixed_prep_point(const MixedPrep& mx, const double* __restrict__ pts, int p) {
    float4* __restrict__ out = reinterpret_cast<float4*>(mx.rec32 + (size_t)kRec32 * p);
    out[0] = make_float4((float)(pts[3 * (size_t)p] - mx.o.x), (float)(pts[3 * (size_t)p + 1] - mx.o.y),
                         (float)(pts[3 * (size_t)p + 2] - mx.o.z), 0.f);
}
// the fp32 operands of the parameter vector (pts, camtab), outside a solve (test and timing entries)
__global__ void k_mixed_prep(MixedPrep mx, const double* __restrict__ pts, int C, int P, int bc) {
    if ((int)blockIdx.x < bc) {
        const int c = blockIdx.x * blockDim.x + threadIdx.x;
        if (c < C) mixed_prep_camera(mx, C, c);
        return;
    }
    const int p = (blockIdx.x - bc) * blockDim.x + threadIdx.x;
    if (p < P) mixed_prep_point(mx, pts, p);
}

// Per point: Vinv = (V + reg diag(si_p^2))^-1 and e_p = Vinv g_p.
__global__ void k_point_prep(const double* __restrict__ V, const double* __restrict__ gp,
                             const double* __restrict__ sip, const double* __restrict__ dp_extra,
                             int P, double reg, double* __restrict__ Vinv, double* __restrict__ e,
                             const double* __restrict__ pts, double* __restrict__ rhsrec) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    point_prep_one(V, gp, sip, dp_extra, p, reg, Vinv, e, pts, rhsrec);
}

// out = packed upper triangle of A^-1 for a symmetric positive definite 6x6 A (Cholesky, inverse of the factor)
__device__ __forceinline__ void spd6_inverse(const double (&A)[6][6], double (&out)[21]) {
    double Lm[6][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double sj = A[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) sj -= Lm[j][k] * Lm[j][k];
        const double ljj = sqrt(sj);
        Lm[j][j] = ljj;
        const double inv = 1.0 / ljj;
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
            double t = A[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) t -= Lm[i][k] * Lm[j][k];
            Lm[i][j] = t * inv;
        }
    }
    double M[6][6];                         // M = L^-1 (lower triangular)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        M[j][j] = 1.0 / Lm[j][j];
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
            double t = 0.0;
#pragma unroll
            for (int k = j; k < i; ++k) t -= Lm[i][k] * M[k][j];
            M[i][j] = t / Lm[i][i];
        }
    }
    int n = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = a; b < 6; ++b) {
            double t = 0.0;
#pragma unroll
            for (int k = b; k < 6; ++k) t += M[k][a] * M[k][b];
            out[n++] = t;
        }
}

// Per camera: Minv = (U + diag(Dc))^-1 (6x6, Cholesky), the block-Jacobi preconditioner of the
// reduced camera system; Dc = reg si_c^2 is also written out.  Camera-sized PCG vectors (Dc, Minv,
// acc, x, r, p, s, u) are stored plane-major, element k of camera c at [k*C + c], so that the
// one-thread-per-camera PCG kernels read and write them fully coalesced.
// sd (optional, plane-major [21][C]): sum_i W_i Vinv W_i^T of the camera's own observations -- with it the block
// is the true diagonal block of the reduced camera matrix S (Schur-diagonal preconditioner) and Dc is read, not
// written (it was formed before the pass that produced sd).
__device__ __forceinline__ void cam_prep_regs(const double* __restrict__ Ugc, const double* __restrict__ sic,
                                              const double* __restrict__ dc_extra, int C, int c, double reg,
                                              double* __restrict__ Dc, double* __restrict__ Minv,
                                              const double* __restrict__ sd, double (&out)[21]) {
    double A[6][6];
    {
        int n = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = a; b < 6; ++b) {
                A[a][b] = Ugc[(size_t)c * 27 + n] - (sd ? sd[(size_t)n * C + c] : 0.0);
                A[b][a] = A[a][b];
                ++n;
            }
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        double d;
        if (sd) d = Dc[(size_t)a * C + c];
        else {
            d = dc_extra ? dc_extra[6 * (size_t)c + a] : reg * sic[6 * (size_t)c + a] * sic[6 * (size_t)c + a];
            Dc[(size_t)a * C + c] = d;                  // plane-major over cameras (coalesced in the PCG)
        }
        A[a][a] += d;
    }
    spd6_inverse(A, out);
#pragma unroll
    for (int n = 0; n < 21; ++n) Minv[(size_t)n * C + c] = out[n];          // packed upper triangle, plane n = [C]
}
__device__ __forceinline__ void cam_prep_one(const double* __restrict__ Ugc, const double* __restrict__ sic,
                                             const double* __restrict__ dc_extra, int C, int c, double reg,
                                             double* __restrict__ Dc, double* __restrict__ Minv,
                                             const double* __restrict__ sd = nullptr) {
    double out[21];
    cam_prep_regs(Ugc, sic, dc_extra, C, c, reg, Dc, Minv, sd, out);
}

__global__ void k_cam_prep(const double* __restrict__ Ugc, const double* __restrict__ sic,
                           const double* __restrict__ dc_extra, int C, double reg,
                           double* __restrict__ Dc, double* __restrict__ Minv) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    cam_prep_one(Ugc, sic, dc_extra, C, c, reg, Dc, Minv);
}

// Schur-diagonal preconditioner: Minv = (U + Dc - sum_i W_i Vinv W_i^T)^-1 per camera, after the pass that summed
// the last term (k_cam_schur MODE 1 with DIAG) and its all-reduce.
__global__ __launch_bounds__(64) void k_cam_prep_schur(const double* __restrict__ Ugc, const double* __restrict__ sd, int C,
                                                       double* __restrict__ Dc, double* __restrict__ Minv) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    cam_prep_one(Ugc, nullptr, nullptr, C, c, 0.0, Dc, Minv, sd);
}

// Everything between the Cauchy product and the reduced right-hand side in one launch: the
// regularisation term from the exchange scalars (every thread evaluates the same few flops, block 0
// publishes it in scalar slot 13), blocks [0, bc): one camera per thread (Dc, Minv),
// blocks [bc, grid): one point per thread (Vinv, e).
__global__ __launch_bounds__(64) void k_prep(double* __restrict__ sc, double Delta, double reg_min,
                                             const double* __restrict__ Ugc, const double* __restrict__ V,
                                             const double* __restrict__ gp, const double* __restrict__ si,
                                             int C, int P, int bc, double* __restrict__ Dc,
                                             double* __restrict__ Minv,
                                             double* __restrict__ Vinv, double* __restrict__ e,
                                             const double* __restrict__ jd_part, int jd_n, double eta_min, double eta_max,
                                             const double* __restrict__ pts, double* __restrict__ rhsrec, MixedPrep mx) {
    // G11 = |J D^2 g|^2: either already in slot 1 (k_finish, then all-reduced over ranks) or summed here from
    // k_jdot's per-workgroup partials -- every workgroup (one wave) repeats the same 256-term sum in
    // k_finish's order, which is cheaper than a launch in between
    double G11;
    if (jd_part != nullptr) {
        double s = 0.0;
        for (int b = threadIdx.x; b < jd_n; b += 64) s += jd_part[b];
        G11 = wave_sum(s);
        G11 = __shfl(G11, 0);
        if (blockIdx.x == 0 && threadIdx.x == 0) sc[1] = G11;
    } else {
        G11 = sc[1];
    }
    const double reg = reg_from_scalars(sc, G11, Delta, reg_min);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        sc[13] = reg;
        // forcing term of this iteration's PCG (slot 15) from the drop of the scaled gradient norm (slot 14 keeps the
        // previous |g_h|^2; 0 at the start of a solve)
        const double a11 = sc[8] + sc[16 + 1], prev = sc[14];
        double eta = eta_max;
        if (prev > 0.0 && a11 >= 0.0) eta = fmin(eta_max, fmax(eta_min, sqrt(a11 / prev)));
        sc[15] = eta;
        sc[14] = a11;
    }
    if ((int)blockIdx.x < bc) {
        const int c = blockIdx.x * blockDim.x + threadIdx.x;
        if (c >= C) return;
        if (mx.rt32 != nullptr) mixed_prep_camera(mx, C, c);
        if (Minv != nullptr) { cam_prep_one(Ugc, si, nullptr, C, c, reg, Dc, Minv); return; }
#pragma unroll                                        // Schur-diagonal preconditioner: only Dc here, Minv by k_cam_prep_schur
        for (int a = 0; a < 6; ++a) Dc[(size_t)a * C + c] = reg * si[6 * (size_t)c + a] * si[6 * (size_t)c + a];
        return;
    }
    const int p = (blockIdx.x - bc) * blockDim.x + threadIdx.x;
    if (p >= P) return;
    if (mx.rt32 != nullptr) mixed_prep_point(mx, pts, p);
    point_prep_one(V, gp, si + 6 * (size_t)C, nullptr, p, reg, Vinv, e, pts, rhsrec);
}

// ---------------------------------------------------------------------------------------------
// K4/K5: the implicit Schur complement  (S - Dc) v = sum_i Jc_i^T ( Jc_i v_c - Jp_i z_p ),
// z_p = Vinv_p sum_{i in p} Jp_i^T Jc_i v_c, in two passes and without atomics:
//   A  k_point_sweep  (point-major, reads the stored Jacobian): z_p for every point, reduced inside the wave
//   B  k_cam_schur    (camera-major, blocks recomputed): the camera sums, one workgroup per camera chunk
// S v = acc + Dc v is completed by the PCG update.  The reduced right-hand side -sum Jc^T Jp e_p is pass B
// alone with z = e (MODE 1).
// ---------------------------------------------------------------------------------------------
// row k of the packed-upper-triangle 6x6 block m times r (block-Jacobi preconditioner)
__device__ __forceinline__ double minv_row(const double* m, const double* r, int k) {
    double z = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int a = k < j ? k : j, b = k < j ? j : k;
        z += m[a * 6 - a * (a - 1) / 2 + (b - a)] * r[j];       // packed upper triangle
    }
    return z;
}

// FUSED (v in LDS, C <= blockDim.x): the launch of pass A is also the PCG update.  Its prologue performs the
// update that k_pcg_update would have done after the PREVIOUS product -- every workgroup redundantly, one
// thread per camera, from the complete old vector set, leaving the new u directly in the LDS table the sweep
// reads -- and stores only its own few cameras into the other vector set.  Nothing a workgroup reads is
// written during the same launch (vector sets and control blocks alternate; the product `acc` is written by
// pass B, a launch of its own), so no inter-workgroup synchronisation is needed.  Launch 0 does k_pcg_init's
// work instead of an update (acc holds the reduced right-hand side term).  Returns at once when the solve
// has finished.
struct PcgFused {
    const double* __restrict__ Dc;
    const double* __restrict__ Minv;
    const double* __restrict__ Ugc;      // launch 0 builds the right-hand side from g_c and acc
    double* __restrict__ vecs;
    PcgCtrl* __restrict__ ctrl2;
    double tol;                          // (unused: kept for the layout of the debug printouts)
    int max_iters;
    const double* __restrict__ tol_dev;  // the forcing term of this iteration (k_prep leaves it in an exchange scalar)
    // LOCAL form (one rank, every camera a single chunk): pass B's workgroup owns its camera, so the per-camera
    // bookkeeping of the iteration (w = acc + Dc u, s, p, x, r, m = Minv s and the partial dot products
    // w.u, s.u, s.m, r.u) runs THERE, once per camera, and the prologue of pass A only needs, for all cameras,
    //     u_new = u - alpha m            (12 doubles per camera instead of 51, no 6x6 products)
    // with alpha from the summed partials and gamma_new = gamma - 2 alpha (s.u) + alpha^2 (s.m), gamma = sum r.u
    // being the TRUE value of the previous iterate (one recurrence step from an exact anchor: no drift).
    double* __restrict__ part;           // [4][C] partial dot products of the cameras (null: not the local form)
};
constexpr int kPcgM = kPcgUcm;           // the local form keeps m = Minv s where the two-kernel form keeps its copy of u

// The scalars of one iteration of the LOCAL form (see PcgFused), from the partial dot products pass B left per camera:
// every thread of the workgroup (one camera each; 16 waves) returns alpha and the control block `co` that follows `ci`
// (co.done = 3: S not SPD / NaN, alpha unusable).  One order of summation wherever it is evaluated -- the prologue of the
// next pass A, or the prologue of k_backsub when the host left the last pass A out (FinalUpdate).
__device__ __forceinline__ double pcg_local_step(const double* __restrict__ part, int C, const PcgCtrl& ci, PcgCtrl& co) {
    __shared__ double red4[16][4];
    const int cam = threadIdx.x;
    double q4[4] = {0.0, 0.0, 0.0, 0.0};
    if (cam < C) {
#pragma unroll
        for (int k = 0; k < 4; ++k) q4[k] = part[(size_t)k * C + cam];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) q4[k] = wave_sum(q4[k]);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) red4[threadIdx.x >> 6][k] = q4[k];
    }
    __syncthreads();
    double tot[4] = {0.0, 0.0, 0.0, 0.0};
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) {
#pragma unroll
        for (int k = 0; k < 4; ++k) tot[k] += red4[w][k];
    }
    const double delta = tot[0], su = tot[1], sm = tot[2], gamma = tot[3];     // w.u, s.u, s.m, r.u (true gamma_i)
    const double beta = ci.iters == 0 ? 0.0 : ci.rz / ci.rz_prev;             // the beta pass B built s and p with
    const double den = delta - (ci.iters == 0 ? 0.0 : beta * gamma / ci.alpha_prev);
    const double alpha = gamma / den;
    co = ci;
    if (!(den > 0.0) || !isfinite(alpha)) { co.done = 3; return 0.0; }
    const double rz = gamma - 2.0 * alpha * su + alpha * alpha * sm;          // gamma_{i+1}
    int done = 0;
    if (!(rz > ci.tol2 * ci.rz0)) done = 1;                   // also catches NaN and a cancelled-out (<= 0) value
    else if (ci.iters + 1 >= ci.max_iters) done = 2;
    co.rz_prev = gamma; co.alpha_prev = alpha; co.rz = rz; co.iters = ci.iters + 1; co.done = done;
    return alpha;
}

// The PCG update in the prologue of a fused pass-A launch (see PcgFused): returns false when the launch has
// nothing more to do (the solve had finished or finishes here; grid-uniform).  Otherwise uu = the new u of
// camera threadIdx.x (zeros for threads without a camera), which the caller puts into its LDS table.
__device__ __forceinline__ bool pcg_fused_update(const PcgFused& pf, const double* __restrict__ acc, int C, int L,
                                                 double (&uu)[6]) {
    __shared__ double red_a[16];
    __shared__ double red_b[16];
    const int n6 = 6 * C;
    // Until the solve finishes, launch L >= 1 sees iters == L - 1, so every address below follows from
    // L alone and all vector loads of the prologue are in flight together.
    PcgCtrl* __restrict__ cout = pf.ctrl2 + ((L + 1) & 1);
    const bool writer = blockIdx.x == 0 && threadIdx.x == 0;
    const int cam = threadIdx.x;
    const bool has = cam < C;
    const int slice = (C + (int)gridDim.x - 1) / (int)gridDim.x;
    const bool own = has && cam >= (int)blockIdx.x * slice && cam < ((int)blockIdx.x + 1) * slice;
    const int set = L == 0 ? 0 : (L - 1) & 1;
    const double* __restrict__ vold = pf.vecs + (size_t)set * kPcgVecs * n6;
    double* __restrict__ vnew = pf.vecs + (size_t)(L == 0 ? 0 : set ^ 1) * kPcgVecs * n6;   // = set L & 1
#pragma unroll
    for (int k = 0; k < 6; ++k) uu[k] = 0.0;
    double m[21];
    PcgCtrl ci;
    if (L != 0) {
        ci = pf.ctrl2[L & 1];
        if (ci.done != 0) {                                   // grid-uniform; before any other load is issued
            if (writer) *cout = ci;
            return false;
        }
    }
    if (pf.part != nullptr && L != 0) {
        // ---- local form: iteration i = L - 1; its product and camera-side bookkeeping were done by pass B(L-1) ----
        double ue[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, me[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (has) {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const size_t e = (size_t)k * C + cam;
                ue[k] = vold[kPcgU * n6 + e];
                me[k] = pf.vecs[kPcgM * n6 + e];
            }
        }
        PcgCtrl co;
        const double alpha = pcg_local_step(pf.part, C, ci, co);
        if (writer) *cout = co;
        if (co.done == 3) return false;                        // S not SPD / NaN: x stays the last good iterate
        const int done = co.done;
        if (has) {
#pragma unroll
            for (int k = 0; k < 6; ++k) uu[k] = ue[k] - alpha * me[k];
            if (own) {
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const size_t e = (size_t)k * C + cam;
                    vnew[kPcgU * n6 + e] = uu[k];
                    if (done != 0) {                          // the deferred x += alpha p of the last iteration (x is kept
                        const double xk = pf.vecs[kPcgX * n6 + e] + alpha * pf.vecs[kPcgP * n6 + e];     // in both sets)
                        pf.vecs[kPcgX * n6 + e] = xk;
                        pf.vecs[(size_t)kPcgVecs * n6 + kPcgX * n6 + e] = xk;
                    }
                }
            }
        }
        return done == 0;                                     // grid-uniform
    }
    if (has) {
#pragma unroll
        for (int n = 0; n < 21; ++n) m[n] = pf.Minv[(size_t)n * C + cam];
    }
    if (L == 0) {
        // start of a solve (k_pcg_init's work): rhs = -g_c - acc (acc = -sum W e from pass B, MODE 1),
        // x = p = s = 0, r = rhs, u = Minv r
        double rr[6];
        double t = 0.0;
        if (has) {
#pragma unroll
            for (int k = 0; k < 6; ++k) rr[k] = -pf.Ugc[(size_t)cam * 27 + 21 + k] - acc[(size_t)k * C + cam];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                uu[k] = minv_row(m, rr, k);
                t += uu[k] * rr[k];
            }
            if (own) {
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const size_t e = (size_t)k * C + cam;
                    vnew[kPcgX * n6 + e] = 0.0; vnew[kPcgP * n6 + e] = 0.0; vnew[kPcgS * n6 + e] = 0.0;
                    vnew[kPcgR * n6 + e] = rr[k];
                    vnew[kPcgU * n6 + e] = uu[k];
                }
            }
        }
        const double rz = block_sum_all(t, red_b);
        const int done = (rz > 0.0) ? 0 : (rz == 0.0 ? 1 : 3);
        if (writer) {
            PcgCtrl c0;
            const double tolv = *pf.tol_dev;
            c0.rz = rz; c0.rz0 = rz; c0.tol2 = tolv * tolv; c0.rz_prev = 1.0; c0.alpha_prev = 1.0;
            c0.iters = 0; c0.max_iters = pf.max_iters; c0.done = done; c0.pad = 1;
            *cout = c0;
        }
        return done == 0;                                         // grid-uniform
    }
    double ue[6], we[6], so[6], ro[6];
    if (has) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const size_t e = (size_t)k * C + cam;
            ue[k] = vold[kPcgU * n6 + e];
            we[k] = acc[e] + pf.Dc[e] * ue[k];
            so[k] = vold[kPcgS * n6 + e];
            ro[k] = vold[kPcgR * n6 + e];
        }
    }
    double d = 0.0;
    if (has) {
#pragma unroll
        for (int k = 0; k < 6; ++k) d += we[k] * ue[k];
    }
    const double delta = block_sum_all(d, red_a);
    const double gamma = ci.rz;
    const double beta = ci.iters == 0 ? 0.0 : gamma / ci.rz_prev;
    const double den = delta - (ci.iters == 0 ? 0.0 : beta * gamma / ci.alpha_prev);
    const double alpha = gamma / den;
    if (!(den > 0.0) || !isfinite(alpha)) {                    // S not SPD / NaN: uniform in the grid
        if (writer) { PcgCtrl co = ci; co.done = 3; *cout = co; }
        return false;
    }
    double t = 0.0;
    if (has) {
        double rr[6], ss[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            ss[k] = we[k] + beta * so[k];
            rr[k] = ro[k] - alpha * ss[k];
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            uu[k] = minv_row(m, rr, k);
            t += uu[k] * rr[k];
        }
        if (own) {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const size_t e = (size_t)k * C + cam;
                // p and x of the few cameras this workgroup stores are read late: keeping them in
                // registers from the top would spill
                const double pk = ue[k] + beta * vold[kPcgP * n6 + e];
                vnew[kPcgP * n6 + e] = pk;
                vnew[kPcgX * n6 + e] = vold[kPcgX * n6 + e] + alpha * pk;
                vnew[kPcgR * n6 + e] = rr[k];
                vnew[kPcgS * n6 + e] = ss[k];
                vnew[kPcgU * n6 + e] = uu[k];
            }
        }
    }
    const double rz = block_sum_all(t, red_b);
    int done = 0;
    if (!(rz > ci.tol2 * ci.rz0)) done = 1;                   // also catches NaN
    else if (ci.iters + 1 >= ci.max_iters) done = 2;
    if (writer) {
        PcgCtrl co = ci;
        co.rz_prev = ci.rz; co.alpha_prev = alpha; co.rz = rz; co.iters = ci.iters + 1; co.done = done;
        *cout = co;
    }
    return done == 0;                                             // grid-uniform
}

template <bool LDS_VEC, bool FUSED>
__global__ __launch_bounds__(kSweepThreads) void k_point_sweep(
    StepTable st, ObsArrays o, const double* __restrict__ vin, const double* __restrict__ Vinv,
    double* __restrict__ zout, const double* __restrict__ acc, int C,
    const PcgCtrl* __restrict__ ctrl2, int L, PcgFused pf) {
    extern __shared__ __align__(16) double smem[];
    static_assert(!FUSED || LDS_VEC, "the fused PCG launch keeps v in LDS");
    const int n6 = 6 * C;
    // The wave's step list and the indices of its first step form a chain of three dependent loads that
    // depends on nothing else: request it first, so that it overlaps the PCG prologue / the LDS staging.
    const int wg = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    int s = 0, s_end = 0;
    if (wg < st.n_waves) { const int2 w = st.wsteps[wg]; s = w.x; s_end = w.x + w.y; }
    int2 cur = make_int2(0, 0);
    if (s < s_end) cur = st.steps[s];
    int i = cur.x + lane, c = 0, p = 0;
    if (cur.y <= 64 && lane < cur.y) { c = o.cam_idx[i]; p = o.pt_idx[i]; }
    if (FUSED) {
        double uu[6];
        if (!pcg_fused_update(pf, acc, C, L, uu)) return;
        if ((int)threadIdx.x < C) {
#pragma unroll
            for (int k = 0; k < 6; ++k) smem[6 * threadIdx.x + k] = uu[k];
        }
        __syncthreads();
    } else if (ctrl2 != nullptr) {
        // inside the two-kernel PCG: `vin` is the base of the ping-pong vector sets; the control block of this
        // launch (written by the previous k_pcg_update) says which set is current
        const PcgCtrl* __restrict__ ctrl = ctrl2 + (L & 1);
        if (ctrl->done != 0) return;                         // grid-uniform
        vin += (size_t)((ctrl->iters & 1) * kPcgVecs + (LDS_VEC ? kPcgU : kPcgUcm)) * n6;
    }
    if (LDS_VEC && !FUSED) {
        for (int e = threadIdx.x; e < n6; e += blockDim.x) {       // e = k*C + c (plane-major global, coalesced)
            const int k = e / C, cc = e - k * C;
            smem[6 * cc + k] = vin[e];
        }
        __syncthreads();
    }
    // v: LDS table, or the camera-major copy [C][6] in global memory (L2) when 6 C doubles exceed the LDS
    const double* __restrict__ vv = LDS_VEC ? smem : vin;
    auto jcv = [&](const double* jc, int cc, double& t0, double& t1) {    // rows of 48 bytes, 16-byte aligned
        const double2* a2 = reinterpret_cast<const double2*>(vv + 6 * cc);
        const double2 a01 = a2[0], a23 = a2[1], a45 = a2[2];
        const double a[6] = {a01.x, a01.y, a23.x, a23.y, a45.x, a45.y};
        t0 = 0.0; t1 = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) { t0 += jc[k] * a[k]; t1 += jc[6 + k] * a[k]; }
    };

    while (s < s_end) {
        // ---- request the next step's indices ---------------------------------------------------
        int2 nxt = make_int2(0, 0);
        if (s + 1 < s_end) nxt = st.steps[s + 1];
        const int in_ = nxt.x + lane;
        int cn = 0, pn = 0;
        if (nxt.y <= 64 && lane < nxt.y) { cn = o.cam_idx[in_]; pn = o.pt_idx[in_]; }

        double jc[12], jp[6];
        double y[3] = {0.0, 0.0, 0.0};
        if (cur.y > 64) {                          // one point with more than 64 observations
            const int run_end = cur.x + cur.y;
            const int pp = o.pt_idx[cur.x];
            for (int j = cur.x + lane; j < run_end; j += 64) {
                load_blocks(o, j, jc, jp);
                double t0, t1;
                jcv(jc, o.cam_idx[j], t0, t1);
                y[0] += jp[0] * t0 + jp[3] * t1; y[1] += jp[1] * t0 + jp[4] * t1;
                y[2] += jp[2] * t0 + jp[5] * t1;
            }
            y[0] = wave_sum(y[0]); y[1] = wave_sum(y[1]); y[2] = wave_sum(y[2]);
            if (lane == 0) {
                const double* vi = Vinv + kVinvRow * (size_t)pp;
                zout[(size_t)kRec * pp + 3] = vi[0] * y[0] + vi[1] * y[1] + vi[2] * y[2];
                zout[(size_t)kRec * pp + 4] = vi[1] * y[0] + vi[3] * y[1] + vi[4] * y[2];
                zout[(size_t)kRec * pp + 5] = vi[2] * y[0] + vi[4] * y[1] + vi[5] * y[2];
            }
        } else {
            const bool act = lane < cur.y;
            double vi[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            if (act) {
                load_blocks(o, i, jc, jp);
#pragma unroll                                     // every lane of a run reads its point's block: no dependent
                for (int k = 0; k < 6; ++k) vi[k] = Vinv[kVinvRow * (size_t)p + k];       // gather after the reduction
                double t0, t1;
                jcv(jc, c, t0, t1);
                y[0] = jp[0] * t0 + jp[3] * t1; y[1] = jp[1] * t0 + jp[4] * t1;
                y[2] = jp[2] * t0 + jp[5] * t1;
            }
            const int key = act ? p : -1 - lane;                  // run key = point index
            seg_reduce_serial<3>(y, key, lane);
            const int prev = lane_below(key, lane);
            if (act && (lane == 0 || prev != key)) {              // first lane of the point's run
                zout[(size_t)kRec * p + 3] = vi[0] * y[0] + vi[1] * y[1] + vi[2] * y[2];
                zout[(size_t)kRec * p + 4] = vi[1] * y[0] + vi[3] * y[1] + vi[4] * y[2];
                zout[(size_t)kRec * p + 5] = vi[2] * y[0] + vi[4] * y[1] + vi[5] * y[2];
            }
        }
        cur = nxt; i = in_; c = cn; p = pn;
        ++s;
    }
}

// Pass A, recomputing form (C <= kRcMaxCams): no stored Jacobian is read.  With j_k = row k of d r/d X = A R,
// v = X - T and the camera vector u = (u_w, u_T):
//     row k of d r/d w . u_w = -(j_k x v) . a',   a' = u_w - b (w x u_w) + c (w x (w x u_w))
//     (Jc u)_k = -j_k . g,   g = v x a' + u_T,          y_p = sum_i Jp_i^T (Jc_i u),   z_p = Vinv_p y_p
// (the identity (m x w).u = m.(w x u) moves the rotation Jacobian from the observation to the camera), so the
// LDS table holds 18 doubles per camera, R | T | a' | u_T, 144 KB at 1000 cameras: R | T copied from the compact
// view of the camera table, a' and u_T built per launch from u by one thread per camera.  Per observation: 8 B of
// indices from HBM, 144 B from LDS, ~90 flop -- against 104 B from HBM for the form that reads J.
constexpr int kRcRow = 18;
constexpr int kRcMaxCams = 1100;          // 18 doubles x C within the 160 KiB LDS (FUSED additionally needs C <= 1024)

// The table of the recomputing pass A in global memory (GTAB): R | T | a' | u_T per camera, 18 doubles, for the vector
// u of the current PCG iterate (u_planes: plane-major [6][C]; with ctrl2 the base of the vector sets, as in the sweep).
__global__ __launch_bounds__(256) void k_rc_table(const double* __restrict__ camtab, const double* __restrict__ u_planes,
                                                  const PcgCtrl* __restrict__ ctrl2, int L, int C,
                                                  double* __restrict__ rctab) {
    if (ctrl2 != nullptr) {
        const PcgCtrl* __restrict__ ctrl = ctrl2 + (L & 1);
        if (ctrl->done != 0) return;                          // grid-uniform
        u_planes += (size_t)((ctrl->iters & 1) * kPcgVecs + kPcgU) * 6 * C;
    }
    const int cam = blockIdx.x * blockDim.x + threadIdx.x;
    if (cam >= C) return;
    const double* __restrict__ rt = camtab + cam_rt_offset(C) + (size_t)kCamRT * cam;
    const double* __restrict__ wbc = camtab + cam_wbc_offset(C);
    double* __restrict__ row = rctab + (size_t)kRcRow * cam;
#pragma unroll
    for (int k = 0; k < kCamRT; ++k) row[k] = rt[k];
    double u[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) u[k] = u_planes[(size_t)k * C + cam];
    const double wx = wbc[cam], wy = wbc[(size_t)C + cam], wz = wbc[2 * (size_t)C + cam];
    const double b = wbc[3 * (size_t)C + cam], cc = wbc[4 * (size_t)C + cam];
    const double c0 = wy * u[2] - wz * u[1], c1 = wz * u[0] - wx * u[2], c2 = wx * u[1] - wy * u[0];   // w x u_w
    const double d0 = wy * c2 - wz * c1, d1 = wz * c0 - wx * c2, d2 = wx * c1 - wy * c0;               // w x (w x u_w)
    row[12] = u[0] - b * c0 + cc * d0; row[13] = u[1] - b * c1 + cc * d1; row[14] = u[2] - b * c2 + cc * d2;
    row[15] = u[3]; row[16] = u[4]; row[17] = u[5];
}

// the same table with fp32 operands (mixed-precision product): [C][20] float = R | T - o (from rt32) | a' | u_T . .
__global__ __launch_bounds__(256) void k_rc_table32(const double* __restrict__ camtab, const float* __restrict__ rt32,
                                                    const double* __restrict__ u_planes, const PcgCtrl* __restrict__ ctrl2,
                                                    int L, int C, float* __restrict__ rctab32) {
    if (ctrl2 != nullptr) {
        const PcgCtrl* __restrict__ ctrl = ctrl2 + (L & 1);
        if (ctrl->done != 0) return;                          // grid-uniform
        u_planes += (size_t)((ctrl->iters & 1) * kPcgVecs + kPcgU) * 6 * C;
    }
    const int cam = blockIdx.x * blockDim.x + threadIdx.x;
    if (cam >= C) return;
    const float4* __restrict__ src = reinterpret_cast<const float4*>(rt32 + (size_t)kRt32 * cam);
    float4* __restrict__ row = reinterpret_cast<float4*>(rctab32 + (size_t)kRc32Row * cam);
    row[0] = src[0]; row[1] = src[1]; row[2] = src[2];
    const double* __restrict__ wbc = camtab + cam_wbc_offset(C);
    double u[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) u[k] = u_planes[(size_t)k * C + cam];
    const double wx = wbc[cam], wy = wbc[(size_t)C + cam], wz = wbc[2 * (size_t)C + cam];
    const double b = wbc[3 * (size_t)C + cam], cc = wbc[4 * (size_t)C + cam];
    const double c0 = wy * u[2] - wz * u[1], c1 = wz * u[0] - wx * u[2], c2 = wx * u[1] - wy * u[0];   // w x u_w
    const double d0 = wy * c2 - wz * c1, d1 = wz * c0 - wx * c2, d2 = wx * c1 - wy * c0;               // w x (w x u_w)
    row[3] = make_float4((float)(u[0] - b * c0 + cc * d0), (float)(u[1] - b * c1 + cc * d1), (float)(u[2] - b * c2 + cc * d2),
                         (float)u[3]);
    row[4] = make_float4((float)u[4], (float)u[5], 0.f, 0.f);
}
// five LDS-DMA pieces per 80-byte row (rows_request with the fp32 table)
__device__ __forceinline__ void rows_request32(const float* __restrict__ table, int cam, int lane, float* slab) {
#pragma unroll
    for (int k = 0; k < kRow32Pieces; ++k) {
        int q = k * 64 + lane;
        asm volatile("" : "+v"(q));              // recomputed per request: hoisted out of the walk, the piece addresses spill
        const int row = q / kRow32Pieces, piece = q - kRow32Pieces * row;
        const int cr = __shfl(cam, row);
        __builtin_amdgcn_global_load_lds(table + (size_t)cr * kRc32Row + 4 * piece,
                                         (__attribute__((address_space(3))) void*)(slab + 4 * 64 * k), 16, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// GTAB (more cameras than the LDS holds): the table [C][18] was written to global memory by k_rc_table and its rows are
// gathered from L2 by nine 16-byte loads per observation; `vin` is not used.
template <bool FUSED, bool GTAB = false>
__global__ __launch_bounds__(kSweepThreads) void k_point_sweep_rc(
    StepTable st, const int* __restrict__ cam_idx, const int* __restrict__ pt_idx,
    const double* __restrict__ camtab, const double* __restrict__ pts, KMat K, const double* __restrict__ vin,
    const double* __restrict__ Vinv, double* __restrict__ zout, const double* __restrict__ acc, int C,
    const PcgCtrl* __restrict__ ctrl2, int L, PcgFused pf, const double* __restrict__ rctab) {
    static_assert(!(FUSED && GTAB), "the fused PCG update needs the table in LDS");
    extern __shared__ __align__(16) double smem[];
    if (kVinvInRec >= 0) Vinv = zout + kVinvInRec;            // record layout 2: one base address for both
    const int n6 = 6 * C;
    const int wg = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    int s = 0, s_end = 0;
    if (wg < st.n_waves) { const int2 w = st.wsteps[wg]; s = w.x; s_end = w.x + w.y; }
    // Software pipeline of the walk: step descriptors three steps ahead, indices two ahead, the point and its
    // inverse block one ahead -- every load of a step is in flight while the previous step computes.
    int2 cur = make_int2(0, 0), nxt = make_int2(0, 0), nn = make_int2(0, 0);
    if (s < s_end) cur = st.steps[s];
    if (s + 1 < s_end) nxt = st.steps[s + 1];
    if (s + 2 < s_end) nn = st.steps[s + 2];
    int c = 0, p = 0, cn = 0, pn = 0;
    if (cur.y <= 64 && lane < cur.y) { c = cam_idx[cur.x + lane]; p = pt_idx[cur.x + lane]; }
    if (nxt.y <= 64 && lane < nxt.y) { cn = cam_idx[nxt.x + lane]; pn = pt_idx[nxt.x + lane]; }
    const double* __restrict__ rt = camtab + cam_rt_offset(C);                    // compact [C][12] = R | T
    const double* __restrict__ wbc = camtab + cam_wbc_offset(C);                  // w, b, c plane-major [5][C]
    auto put_au = [&](int cam, const double* u) {          // a' and u_T of camera `cam` from its u (6)
        const double wx = wbc[cam], wy = wbc[(size_t)C + cam], wz = wbc[2 * (size_t)C + cam];
        const double b = wbc[3 * (size_t)C + cam], cc = wbc[4 * (size_t)C + cam];
        const double c0 = wy * u[2] - wz * u[1], c1 = wz * u[0] - wx * u[2], c2 = wx * u[1] - wy * u[0];   // w x u_w
        const double d0 = wy * c2 - wz * c1, d1 = wz * c0 - wx * c2, d2 = wx * c1 - wy * c0;               // w x (w x u_w)
        double* __restrict__ row = smem + (size_t)kRcRow * cam;
        row[12] = u[0] - b * c0 + cc * d0; row[13] = u[1] - b * c1 + cc * d1; row[14] = u[2] - b * c2 + cc * d2;
        row[15] = u[3]; row[16] = u[4]; row[17] = u[5];
    };
    if (FUSED) {
        double uu[6];
        if (!pcg_fused_update(pf, acc, C, L, uu)) return;
        if ((int)threadIdx.x < C) put_au(threadIdx.x, uu);
    } else {
        if (ctrl2 != nullptr) {                               // two-kernel PCG: vin = base of the vector sets
            const PcgCtrl* __restrict__ ctrl = ctrl2 + (L & 1);
            if (ctrl->done != 0) return;                      // grid-uniform
            vin += (size_t)((ctrl->iters & 1) * kPcgVecs + kPcgU) * n6;
        }
        if (!GTAB) {
            for (int cam = threadIdx.x; cam < C; cam += blockDim.x) {     // vin: plane-major [6][C]
                double u[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) u[k] = vin[(size_t)k * C + cam];
                put_au(cam, u);
            }
        }
    }
    if (!GTAB) {
        for (int e = threadIdx.x; e < C * (kCamRT / 2); e += blockDim.x) {     // R | T: 6 x 16 bytes per camera, coalesced
            const int cam = e / (kCamRT / 2), k = e - cam * (kCamRT / 2);
            reinterpret_cast<double2*>(smem + (size_t)kRcRow * cam)[k] = reinterpret_cast<const double2*>(rt)[e];
        }
        __syncthreads();
    }
    const double* __restrict__ table = GTAB ? rctab : smem;
    // GTAB: the rows of a step are gathered through the wave's LDS slab (rows_request, see k_resjac); a point with more
    // than 64 observations reads its rows from global memory as before
    double* const slab = smem + (size_t)(threadIdx.x >> 6) * kRowSlabDoubles;
    static_assert(kRcRow == kCamRow, "rows_request moves rows of kCamRow doubles");

    // y contribution of one observation from its camera row (nine 16-byte pieces) and its point X
    auto contrib = [&](const double2* __restrict__ row, double X, double Y, double Z, double* y) {
        const double2 r01 = row[0], r23 = row[1], r45 = row[2], r67 = row[3], r8t = row[4], t12 = row[5];
        const double2 a01 = row[6], a2u = row[7], u12 = row[8];
        const double R0 = r01.x, R1 = r01.y, R2 = r23.x, R3 = r23.y, R4 = r45.x, R5 = r45.y, R6 = r67.x, R7 = r67.y,
                     R8 = r8t.x;
        const double vx = X - r8t.y, vy = Y - t12.x, vz = Z - t12.y;
        const double qx = R0 * vx + R1 * vy + R2 * vz;
        const double qy = R3 * vx + R4 * vy + R5 * vz;
        const double qz = R6 * vx + R7 * vy + R8 * vz;
        const double px = K.k[0] * qx + K.k[1] * qy + K.k[2] * qz;
        const double py = K.k[3] * qx + K.k[4] * qy + K.k[5] * qz;
        const double pz = K.k[6] * qx + K.k[7] * qy + K.k[8] * qz;
        const double iz = 1.0 / pz;
        // g = v x a' + u_T
        const double ax = a01.x, ay = a01.y, az = a2u.x;
        const double g0 = vy * az - vz * ay + a2u.y, g1 = vz * ax - vx * az + u12.x, g2 = vx * ay - vy * ax + u12.y;
        y[0] = 0.0; y[1] = 0.0; y[2] = 0.0;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const double pk = (k == 0 ? px : py) * iz;
            const double b0 = (K.k[3 * k + 0] - pk * K.k[6]) * iz;
            const double b1 = (K.k[3 * k + 1] - pk * K.k[7]) * iz;
            const double b2 = (K.k[3 * k + 2] - pk * K.k[8]) * iz;
            const double j0 = b0 * R0 + b1 * R3 + b2 * R6;            // row k of A R
            const double j1 = b0 * R1 + b1 * R4 + b2 * R7;
            const double j2 = b0 * R2 + b1 * R5 + b2 * R8;
            const double tk = -(j0 * g0 + j1 * g1 + j2 * g2);        // (Jc u)_k
            y[0] += j0 * tk; y[1] += j1 * tk; y[2] += j2 * tk;
        }
    };
    auto row_of = [&](int cc) { return reinterpret_cast<const double2*>(table + (size_t)kRcRow * cc); };

    // the point one step ahead; its inverse block too, except with GTAB, where the registers of a second copy across
    // the step would spill: there it is requested at the top of the step that uses it, behind the row traffic
    auto load_point = [&](bool on, int pp, double* X, double* vi) {
        if (on) {
            if (kVinvInRec >= 0) {               // record layout 2: coordinates and inverse block from ONE 128-byte record
                const double2* __restrict__ rp = reinterpret_cast<const double2*>(zout + (size_t)kRec * pp);
                const double2 xy = rp[0], zz = rp[1];
                X[0] = xy.x; X[1] = xy.y; X[2] = zz.x;
                if (!GTAB) {
                    const double2 v0 = rp[4], v1 = rp[5], v2 = rp[6];
                    vi[0] = v0.x; vi[1] = v0.y; vi[2] = v1.x; vi[3] = v1.y; vi[4] = v2.x; vi[5] = v2.y;
                }
            } else {
                const double* __restrict__ Xp = pts + 3 * (size_t)pp;
                X[0] = Xp[0]; X[1] = Xp[1]; X[2] = Xp[2];
                if (!GTAB) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) vi[k] = Vinv[kVinvRow * (size_t)pp + k];
                }
            }
        }
    };
    double X[3] = {0.0, 0.0, 0.0}, vi[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    load_point(cur.y <= 64 && lane < cur.y, p, X, vi);
    bool requested = false;                       // (wave-uniform) the slab holds / will hold the rows of step `cur`
    while (s < s_end) {
        double y[3] = {0.0, 0.0, 0.0};
        if (GTAB && cur.y <= 64) {
            // this step's rows out of the slab and the next step's request BEFORE this iteration's own loads are issued
            if (!requested) rows_request(rctab, c, lane, slab);
            rows_wait();
            if (lane < cur.y) contrib(reinterpret_cast<const double2*>(slab + (size_t)lane * kRcRow), X[0], X[1], X[2], y);
            rows_read_done();
            requested = s + 1 < s_end && nxt.y <= 64;
            if (requested) rows_request(rctab, cn, lane, slab);
        } else {
            requested = false;
        }
        int2 n3 = make_int2(0, 0);
        if (s + 3 < s_end) n3 = st.steps[s + 3];
        n3.x = __builtin_amdgcn_readfirstlane(n3.x);      // wave-uniform: keep the descriptor in scalar registers
        n3.y = __builtin_amdgcn_readfirstlane(n3.y);
        int c2 = 0, p2 = 0;
        if (nn.y <= 64 && lane < nn.y) { c2 = cam_idx[nn.x + lane]; p2 = pt_idx[nn.x + lane]; }
        double Xn[3] = {0.0, 0.0, 0.0}, vin_[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        load_point(nxt.y <= 64 && lane < nxt.y, pn, Xn, vin_);
        if (GTAB && cur.y <= 64 && lane < cur.y) {
#pragma unroll
            for (int k = 0; k < 6; ++k) vi[k] = Vinv[kVinvRow * (size_t)p + k];
        }
        if (cur.y > 64) {                          // one point with more than 64 observations
            const int run_end = cur.x + cur.y;
            const int pp = pt_idx[cur.x];
            const double* __restrict__ Xp = pts + 3 * (size_t)pp;
            const double Xl = Xp[0], Yl = Xp[1], Zl = Xp[2];
            for (int j = cur.x + lane; j < run_end; j += 64) {
                double w[3];
                contrib(row_of(cam_idx[j]), Xl, Yl, Zl, w);
                y[0] += w[0]; y[1] += w[1]; y[2] += w[2];
            }
            y[0] = wave_sum(y[0]); y[1] = wave_sum(y[1]); y[2] = wave_sum(y[2]);
            if (lane == 0) {
                const double* vl = Vinv + kVinvRow * (size_t)pp;
                zout[(size_t)kRec * pp + 3] = vl[0] * y[0] + vl[1] * y[1] + vl[2] * y[2];
                zout[(size_t)kRec * pp + 4] = vl[1] * y[0] + vl[3] * y[1] + vl[4] * y[2];
                zout[(size_t)kRec * pp + 5] = vl[2] * y[0] + vl[4] * y[1] + vl[5] * y[2];
            }
        } else {
            const bool act = lane < cur.y;
            if (!GTAB && act) contrib(row_of(c), X[0], X[1], X[2], y);
            const int key = act ? p : -1 - lane;
            seg_reduce_serial<3>(y, key, lane);
            const int prev = lane_below(key, lane);
            if (act && (lane == 0 || prev != key)) {
                zout[(size_t)kRec * p + 3] = vi[0] * y[0] + vi[1] * y[1] + vi[2] * y[2];
                zout[(size_t)kRec * p + 4] = vi[1] * y[0] + vi[3] * y[1] + vi[4] * y[2];
                zout[(size_t)kRec * p + 5] = vi[2] * y[0] + vi[4] * y[1] + vi[5] * y[2];
            }
        }
        cur = nxt; nxt = nn; nn = n3;
        c = cn; p = pn; cn = c2; pn = p2;
#pragma unroll
        for (int k = 0; k < 3; ++k) X[k] = Xn[k];
        if (!GTAB) {
#pragma unroll
            for (int k = 0; k < 6; ++k) vi[k] = vin_[k];
        }
        ++s;
    }
}

// Pass A with fp32 OPERANDS and fp64 arithmetic (see MixedPrep): the structure of k_point_sweep_rc with 80-byte camera
// rows R | T - o | a' | u_T (five 16-byte pieces instead of nine, in LDS or behind the LDS-DMA slab), the point read
// from its 32-byte fp32 record, z written there as three floats.
template <bool FUSED, bool GTAB = false, bool Z64 = false>
__global__ __launch_bounds__(kSweepThreads) void k_point_sweep_rc32(
    StepTable st, const int* __restrict__ cam_idx, const int* __restrict__ pt_idx,
    const double* __restrict__ camtab, const float* __restrict__ rt32, KMat K, const double* __restrict__ vin,
    const double* __restrict__ Vinv, float* __restrict__ rec32, const double* __restrict__ acc, int C,
    const PcgCtrl* __restrict__ ctrl2, int L, PcgFused pf, const float* __restrict__ rctab32, double* __restrict__ zout64) {
    // Z64: pass B keeps its fp64 records, z goes there (zout64: [P][kRec]) unrounded
    static_assert(!(FUSED && GTAB), "the fused PCG update needs the table in LDS");
    static_assert(kVinvInRec < 0, "the mixed-precision product reads the inverse point blocks from their own array");
    extern __shared__ __align__(16) double smem_d[];
    float* const smem = reinterpret_cast<float*>(smem_d);
    const int n6 = 6 * C;
    const int wg = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    int s = 0, s_end = 0;
    if (wg < st.n_waves) { const int2 w = st.wsteps[wg]; s = w.x; s_end = w.x + w.y; }
    int2 cur = make_int2(0, 0), nxt = make_int2(0, 0), nn = make_int2(0, 0);
    if (s < s_end) cur = st.steps[s];
    if (s + 1 < s_end) nxt = st.steps[s + 1];
    if (s + 2 < s_end) nn = st.steps[s + 2];
    const double* __restrict__ wbc = camtab + cam_wbc_offset(C);                  // w, b, c plane-major [5][C]
    auto put_au = [&](int cam, const double* u) {          // a' and u_T of camera `cam` from its u (6), rounded to fp32
        const double wx = wbc[cam], wy = wbc[(size_t)C + cam], wz = wbc[2 * (size_t)C + cam];
        const double b = wbc[3 * (size_t)C + cam], cc = wbc[4 * (size_t)C + cam];
        const double c0 = wy * u[2] - wz * u[1], c1 = wz * u[0] - wx * u[2], c2 = wx * u[1] - wy * u[0];   // w x u_w
        const double d0 = wy * c2 - wz * c1, d1 = wz * c0 - wx * c2, d2 = wx * c1 - wy * c0;               // w x (w x u_w)
        float4* __restrict__ row = reinterpret_cast<float4*>(smem + (size_t)kRc32Row * cam);
        row[3] = make_float4((float)(u[0] - b * c0 + cc * d0), (float)(u[1] - b * c1 + cc * d1),
                             (float)(u[2] - b * c2 + cc * d2), (float)u[3]);
        row[4] = make_float4((float)u[4], (float)u[5], 0.f, 0.f);
    };
    if (FUSED) {
        double uu[6];
        if (!pcg_fused_update(pf, acc, C, L, uu)) return;
        if ((int)threadIdx.x < C) put_au(threadIdx.x, uu);
    } else {
        if (ctrl2 != nullptr) {                               // two-kernel PCG: vin = base of the vector sets
            const PcgCtrl* __restrict__ ctrl = ctrl2 + (L & 1);
            if (ctrl->done != 0) return;                      // grid-uniform
            vin += (size_t)((ctrl->iters & 1) * kPcgVecs + kPcgU) * n6;
        }
        if (!GTAB) {
            for (int cam = threadIdx.x; cam < C; cam += blockDim.x) {     // vin: plane-major [6][C]
                double u[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) u[k] = vin[(size_t)k * C + cam];
                put_au(cam, u);
            }
        }
    }
    // (the indices of the first two steps are requested here, behind the PCG prologue -- held across it they spill --
    // and in front of the staging loop, which hides their latency)
    int c = 0, p = 0, cn = 0, pn = 0;
    if (cur.y <= 64 && lane < cur.y) { c = cam_idx[cur.x + lane]; p = pt_idx[cur.x + lane]; }
    if (nxt.y <= 64 && lane < nxt.y) { cn = cam_idx[nxt.x + lane]; pn = pt_idx[nxt.x + lane]; }
    if (!GTAB) {
        for (int e = threadIdx.x; e < C * (kRt32 / 4); e += blockDim.x) {      // R | T - o: 3 x 16 bytes per camera, coalesced
            const int cam = e / (kRt32 / 4), k = e - cam * (kRt32 / 4);
            reinterpret_cast<float4*>(smem + (size_t)kRc32Row * cam)[k] = reinterpret_cast<const float4*>(rt32)[e];
        }
        __syncthreads();
    }
    const float* __restrict__ table = GTAB ? rctab32 : smem;
    float* const slab = smem + (size_t)(threadIdx.x >> 6) * kRow32SlabFloats;

    // y contribution of one observation from its camera row (five 16-byte pieces) and its point X - o
    auto contrib = [&](const float4* __restrict__ row, float X, float Y, float Z, double* y) {
        const float4 f0 = row[0], f1 = row[1], f2 = row[2], f3 = row[3], f4 = row[4];
        const double R0 = f0.x, R1 = f0.y, R2 = f0.z, R3 = f0.w, R4 = f1.x, R5 = f1.y, R6 = f1.z, R7 = f1.w, R8 = f2.x;
        const double vx = (double)X - (double)f2.y, vy = (double)Y - (double)f2.z, vz = (double)Z - (double)f2.w;
        const double qx = R0 * vx + R1 * vy + R2 * vz;
        const double qy = R3 * vx + R4 * vy + R5 * vz;
        const double qz = R6 * vx + R7 * vy + R8 * vz;
        const double px = K.k[0] * qx + K.k[1] * qy + K.k[2] * qz;
        const double py = K.k[3] * qx + K.k[4] * qy + K.k[5] * qz;
        const double pz = K.k[6] * qx + K.k[7] * qy + K.k[8] * qz;
        const double iz = 1.0 / pz;
        const double ax = f3.x, ay = f3.y, az = f3.z;             // g = v x a' + u_T
        const double g0 = vy * az - vz * ay + (double)f3.w, g1 = vz * ax - vx * az + (double)f4.x,
                     g2 = vx * ay - vy * ax + (double)f4.y;
        y[0] = 0.0; y[1] = 0.0; y[2] = 0.0;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const double pk = (k == 0 ? px : py) * iz;
            const double b0 = (K.k[3 * k + 0] - pk * K.k[6]) * iz;
            const double b1 = (K.k[3 * k + 1] - pk * K.k[7]) * iz;
            const double b2 = (K.k[3 * k + 2] - pk * K.k[8]) * iz;
            const double j0 = b0 * R0 + b1 * R3 + b2 * R6;            // row k of A R
            const double j1 = b0 * R1 + b1 * R4 + b2 * R7;
            const double j2 = b0 * R2 + b1 * R5 + b2 * R8;
            const double tk = -(j0 * g0 + j1 * g1 + j2 * g2);        // (Jc u)_k
            y[0] += j0 * tk; y[1] += j1 * tk; y[2] += j2 * tk;
        }
    };
    auto row_of = [&](int cc) { return reinterpret_cast<const float4*>(table + (size_t)kRc32Row * cc); };
    auto store_z = [&](int pp, const double* vv, const double* y) {
        if (Z64) {
            double* __restrict__ zd = zout64 + (size_t)kRec * pp;
            zd[3] = vv[0] * y[0] + vv[1] * y[1] + vv[2] * y[2];
            zd[4] = vv[1] * y[0] + vv[3] * y[1] + vv[4] * y[2];
            zd[5] = vv[2] * y[0] + vv[4] * y[1] + vv[5] * y[2];
        } else {
            float* __restrict__ zp = rec32 + (size_t)kRec32 * pp;
            zp[3] = (float)(vv[0] * y[0] + vv[1] * y[1] + vv[2] * y[2]);
            *reinterpret_cast<float2*>(zp + 4) = make_float2((float)(vv[1] * y[0] + vv[3] * y[1] + vv[4] * y[2]),
                                                             (float)(vv[2] * y[0] + vv[4] * y[1] + vv[5] * y[2]));
        }
    };
    auto load_point = [&](bool on, int pp, float* X, double* vi) {
        if (on) {
            const float4 xr = *reinterpret_cast<const float4*>(rec32 + (size_t)kRec32 * pp);
            X[0] = xr.x; X[1] = xr.y; X[2] = xr.z;
            if (!GTAB) {
#pragma unroll
                for (int k = 0; k < 6; ++k) vi[k] = Vinv[kVinvRow * (size_t)pp + k];
            }
        }
    };
    float X[3] = {0.f, 0.f, 0.f};
    double vi[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    load_point(cur.y <= 64 && lane < cur.y, p, X, vi);
    bool requested = false;                       // (wave-uniform) the slab holds / will hold the rows of step `cur`
    while (s < s_end) {
        double y[3] = {0.0, 0.0, 0.0};
        if (GTAB && cur.y <= 64) {
            if (!requested) rows_request32(rctab32, c, lane, slab);
            rows_wait();
            if (lane < cur.y) contrib(reinterpret_cast<const float4*>(slab + (size_t)lane * kRc32Row), X[0], X[1], X[2], y);
            rows_read_done();
            requested = s + 1 < s_end && nxt.y <= 64;
            if (requested) rows_request32(rctab32, cn, lane, slab);
        } else {
            requested = false;
        }
        int2 n3 = make_int2(0, 0);
        if (s + 3 < s_end) n3 = st.steps[s + 3];
        n3.x = __builtin_amdgcn_readfirstlane(n3.x);
        n3.y = __builtin_amdgcn_readfirstlane(n3.y);
        int c2 = 0, p2 = 0;
        if (nn.y <= 64 && lane < nn.y) { c2 = cam_idx[nn.x + lane]; p2 = pt_idx[nn.x + lane]; }
        float Xn[3] = {0.f, 0.f, 0.f};
        double vin_[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        load_point(nxt.y <= 64 && lane < nxt.y, pn, Xn, vin_);
        if (GTAB && cur.y <= 64 && lane < cur.y) {
#pragma unroll
            for (int k = 0; k < 6; ++k) vi[k] = Vinv[kVinvRow * (size_t)p + k];
        }
        if (cur.y > 64) {                          // one point with more than 64 observations
            const int run_end = cur.x + cur.y;
            const int pp = pt_idx[cur.x];
            const float4 xr = *reinterpret_cast<const float4*>(rec32 + (size_t)kRec32 * pp);
            const float Xl = xr.x, Yl = xr.y, Zl = xr.z;
            for (int j = cur.x + lane; j < run_end; j += 64) {
                double w[3];
                contrib(row_of(cam_idx[j]), Xl, Yl, Zl, w);
                y[0] += w[0]; y[1] += w[1]; y[2] += w[2];
            }
            y[0] = wave_sum(y[0]); y[1] = wave_sum(y[1]); y[2] = wave_sum(y[2]);
            if (lane == 0) store_z(pp, Vinv + kVinvRow * (size_t)pp, y);
        } else {
            const bool act = lane < cur.y;
            if (!GTAB && act) contrib(row_of(c), X[0], X[1], X[2], y);
            const int key = act ? p : -1 - lane;
            seg_reduce_serial<3>(y, key, lane);
            const int prev = lane_below(key, lane);
            if (act && (lane == 0 || prev != key)) store_z(p, vi, y);
        }
        cur = nxt; nxt = nn; nn = n3;
        c = cn; p = pn; cn = c2; pn = p2;
#pragma unroll
        for (int k = 0; k < 3; ++k) X[k] = Xn[k];
        if (!GTAB) {
#pragma unroll
            for (int k = 0; k < 6; ++k) vi[k] = vin_[k];
        }
        ++s;
    }
}

// The local form of the PCG update as a kernel of its own (more cameras than the fused launch takes: one thread per
// camera of a 1024-thread workgroup).  Pass B's workgroup of every camera has done that camera's bookkeeping and left
// four partial dot products (PcgLocal); this kernel sums them -- every workgroup all of them, in the same order -- and
// applies  u <- u - alpha m  to its own 1024 cameras.  Same arithmetic as the local branch of pcg_fused_update, shifted
// by one in the launch index: launch L reads u from vector set L & 1 and the control block ctrl2[L & 1], writes set
// (L + 1) & 1 and ctrl2[(L + 1) & 1].  (The two-kernel update it replaces, k_pcg_update, reads 51 doubles per camera
// in every one of its workgroups: 49 us at 5000 cameras.)
__global__ __launch_bounds__(1024) void k_pcg_update_local(double* __restrict__ vecs, const double* __restrict__ part,
                                                           PcgCtrl* __restrict__ ctrl2, int L, int C) {
    __shared__ double red4[16][4];
    const size_t n6 = 6 * (size_t)C;
    const PcgCtrl ci = ctrl2[L & 1];
    PcgCtrl* __restrict__ cout = ctrl2 + ((L + 1) & 1);
    const bool writer = blockIdx.x == 0 && threadIdx.x == 0;
    if (ci.done != 0) {                                       // grid-uniform
        if (writer) *cout = ci;
        return;
    }
    double q4[4] = {0.0, 0.0, 0.0, 0.0};
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
#pragma unroll
        for (int k = 0; k < 4; ++k) q4[k] += part[(size_t)k * C + c];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) q4[k] = wave_sum(q4[k]);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) red4[threadIdx.x >> 6][k] = q4[k];
    }
    __syncthreads();
    double tot[4] = {0.0, 0.0, 0.0, 0.0};
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) {
#pragma unroll
        for (int k = 0; k < 4; ++k) tot[k] += red4[w][k];
    }
    const double delta = tot[0], su = tot[1], sm = tot[2], gamma = tot[3];     // w.u, s.u, s.m, r.u (true gamma_i)
    double rz0 = ci.rz0;
    if (ci.iters == 0 && ci.pad == kPcgInitDeferred) {        // started by k_pcg_init_local: gamma_0 is found here
        rz0 = gamma;
        if (!(gamma > 0.0)) {                                 // zero right-hand side (x = 0 is the solution) / NaN
            if (writer) { PcgCtrl co = ci; co.rz = gamma; co.rz0 = gamma; co.done = gamma == 0.0 ? 1 : 3; *cout = co; }
            return;
        }
    }
    const double beta = ci.iters == 0 ? 0.0 : ci.rz / ci.rz_prev;             // the beta pass B built s and p with
    const double den = delta - (ci.iters == 0 ? 0.0 : beta * gamma / ci.alpha_prev);
    const double alpha = gamma / den;
    if (!(den > 0.0) || !isfinite(alpha)) {                    // S not SPD / NaN: x stays the last good iterate
        if (writer) { PcgCtrl co = ci; co.rz0 = rz0; co.done = 3; *cout = co; }
        return;
    }
    const double rz = gamma - 2.0 * alpha * su + alpha * alpha * sm;          // gamma_{i+1}
    int done = 0;
    if (!(rz > ci.tol2 * rz0)) done = 1;                      // also catches NaN and a cancelled-out (<= 0) value
    else if (ci.iters + 1 >= ci.max_iters) done = 2;
    if (writer) {
        PcgCtrl co = ci;
        co.rz_prev = gamma; co.alpha_prev = alpha; co.rz = rz; co.rz0 = rz0; co.iters = ci.iters + 1; co.done = done;
        *cout = co;
    }
    const int cam = blockIdx.x * blockDim.x + threadIdx.x;
    if (cam >= C) return;
    const double* __restrict__ vold = vecs + (size_t)(L & 1) * kPcgVecs * n6;
    double* __restrict__ vnew = vecs + (size_t)((L + 1) & 1) * kPcgVecs * n6;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const size_t e = (size_t)k * C + cam;
        vnew[kPcgU * n6 + e] = vold[kPcgU * n6 + e] - alpha * vecs[kPcgM * n6 + e];
        if (done != 0) {                                      // the deferred x += alpha p of the last iteration (x is kept
            const double xk = vecs[kPcgX * n6 + e] + alpha * vecs[kPcgP * n6 + e];     // in both sets)
            vecs[kPcgX * n6 + e] = xk;
            vecs[(size_t)kPcgVecs * n6 + kPcgX * n6 + e] = xk;
        }
    }
}

// Pass B: acc_c = sum_{i in c} Jc_i^T ( Jc_i v_c - Jp_i z_p )   (MODE 0; v_c wave-uniform, z gathered), or
//         acc_c = - sum Jc_i^T Jp_i e_p                          (MODE 1; the z slot of the records holds e, written by
//         k_prep), over one camera chunk, with the
// blocks recomputed from the camera row and the gathered point.  ROUND (fp32-storage mode, when pass A is the
// form that reads the stored blocks): the recomputed entries are rounded to float first -- pass A applied the
// STORED (rounded) blocks, and the product has to be that of one symmetric matrix.  Output: plane-major acc[k][C] for a single-chunk camera, partial[chunk][6] otherwise.
//   ctrl_done: PCG control block whose `done` voids this launch (null: unconditional)
//   set:       vector set holding u; < 0: take it from ctrl_done->iters (two-kernel PCG); vin then is the base of
//              the sets.  ctrl_done == null: vin is the plane-major vector itself.
// Local form of the fused PCG (PcgFused::part): what the workgroup of camera c does with its finished product.
struct PcgLocal {
    const double* __restrict__ Dc;
    const double* __restrict__ Minv;
    double* __restrict__ vecs;
    double* __restrict__ part;           // null: not the local form
};


// MIXED (MODE 0, exact-block form): the operands of the mixed-precision product -- R, T - o, a', u_T rounded to fp32
// exactly as pass A's table holds them, the point and z from its 32-byte fp32 record (`rec` then points at rec32).
struct MixedB { const double* __restrict__ rtd; };      // [C][12] = R | T - o, fp32 values held as doubles
template <int MODE, bool ROUND, bool MIXED = false>
__global__ __launch_bounds__(kCamThreads) void k_cam_schur(CamMajor cm, const double* __restrict__ camtab,
                                                           const double* __restrict__ rec, KMat K,
                                                           const double* __restrict__ vin, int C,
                                                           double* __restrict__ acc, double* __restrict__ partial,
                                                           const PcgCtrl* __restrict__ ctrl_done, int set, PcgLocal pl,
                                                           MixedB mxb, CamExchange cx) {
    static_assert(!MIXED || (MODE == 0 && !ROUND), "mixed operands: the product, exact-block form");
    __shared__ double red[kCamWaves][6];
    if (ctrl_done != nullptr) {
        if (ctrl_done->done != 0) return;                       // grid-uniform
        if (MODE == 0) vin += (size_t)((set < 0 ? (ctrl_done->iters & 1) : set) * kPcgVecs + kPcgU) * 6 * C;
    }
    const int4 ch = cm.chunks[blockIdx.x];
    const bool local = MODE == 0 && pl.part != nullptr;     // local form of the fused PCG (see PcgLocal)
    const size_t n6l = 6 * (size_t)C;
    double t[kCamTab];
#pragma unroll
    for (int k = 0; k < kCamTab; ++k) t[k] = camtab[(size_t)ch.x * kCamRow + k];      // wave-uniform: scalar loads
    double vc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (MODE == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) vc[k] = vin[(size_t)k * C + ch.x];
    }
    // Exact blocks (ROUND = false): with j_k = row k of d r/d X, v = X - T and h = v x a' + v_T + z (a' as in
    // k_point_sweep_rc), the two residual-space values are u_k = -j_k . h, and the camera sums follow from
    //     q = sum_k u_k j_k :   translation part  -sum q,   rotation part  -(M - b (M x w) + c ((M x w) x w)),  M = sum q x v
    // -- the rotation Jacobian is applied ONCE per workgroup to the summed M instead of per observation (it is
    // linear in M): ~100 flop per observation instead of ~200.  a[0..2] = sum q x v, a[3..5] = sum q.
    double ap[3] = {0.0, 0.0, 0.0};
    if (MODE == 0 && !ROUND) {
        const double wx = t[12], wy = t[13], wz = t[14];
        const double c0 = wy * vc[2] - wz * vc[1], c1 = wz * vc[0] - wx * vc[2], c2 = wx * vc[1] - wy * vc[0];
        const double d0 = wy * c2 - wz * c1, d1 = wz * c0 - wx * c2, d2 = wx * c1 - wy * c0;
        ap[0] = vc[0] - t[15] * c0 + t[16] * d0; ap[1] = vc[1] - t[15] * c1 + t[16] * d1; ap[2] = vc[2] - t[15] * c2 + t[16] * d2;
    }
    double vT[3] = {vc[3], vc[4], vc[5]};                  // the translation part of the camera vector as the sweep uses it
    if (MIXED) {
#pragma unroll
        for (int k = 0; k < kRt32; ++k) t[k] = mxb.rtd[(size_t)ch.x * kRt32 + k];      // wave-uniform: scalar loads
#pragma unroll
        for (int k = 0; k < 3; ++k) { ap[k] = (double)(float)ap[k]; vT[k] = (double)(float)vT[k]; }
    }
    double a[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int k0 = ch.y + (int)threadIdx.x; k0 < ch.z; k0 += kCamThreads * kCamUnroll) {      // see k_cam_blocks
        int p[kCamUnroll];
        double X[kCamUnroll][3], z[kCamUnroll][3];
#pragma unroll
        for (int u = 0; u < kCamUnroll; ++u) {
            const int k = k0 + u * kCamThreads;
            p[u] = k < ch.z ? cm.pt[k] : -1;
        }
#pragma unroll
        for (int u = 0; u < kCamUnroll; ++u) {
            if (MIXED) {                                  // the point's fp32 record: X Y Z z0 | z1 z2, one 32-byte sector
                const float* __restrict__ rp = reinterpret_cast<const float*>(rec) + (size_t)kRec32 * (p[u] < 0 ? 0 : p[u]);
                const float4 r0 = *reinterpret_cast<const float4*>(rp);
                const float2 r1 = *reinterpret_cast<const float2*>(rp + 4);
                X[u][0] = r0.x; X[u][1] = r0.y; X[u][2] = r0.z;
                z[u][0] = r0.w; z[u][1] = r1.x; z[u][2] = r1.y;
            } else {                                      // the point's record: X Y | Z z0 | z1 z2, three 16-byte loads
                const double2* __restrict__ rp = reinterpret_cast<const double2*>(rec + (size_t)kRec * (p[u] < 0 ? 0 : p[u]));
                const double2 r0 = rp[0], r1 = rp[1], r2 = rp[2];
                X[u][0] = r0.x; X[u][1] = r0.y; X[u][2] = r1.x;
                z[u][0] = r1.y; z[u][1] = r2.x; z[u][2] = r2.y;
            }
        }
#pragma unroll
        for (int u = 0; u < kCamUnroll; ++u) {
            if (p[u] < 0) continue;
            if (ROUND) {
                double jc[12], jp[6], rx, ry;
                observe<true>(t, X[u][0], X[u][1], X[u][2], 0.0, 0.0, K, rx, ry, jc, jp);
#pragma unroll
                for (int q = 0; q < 3; ++q) { jc[q] = (double)(float)jc[q]; jc[6 + q] = (double)(float)jc[6 + q]; }
#pragma unroll
                for (int q = 0; q < 6; ++q) jp[q] = (double)(float)jp[q];
#pragma unroll
                for (int q = 0; q < 3; ++q) { jc[3 + q] = -jp[q]; jc[9 + q] = -jp[3 + q]; }
                double u0 = -(jp[0] * z[u][0] + jp[1] * z[u][1] + jp[2] * z[u][2]);
                double u1 = -(jp[3] * z[u][0] + jp[4] * z[u][1] + jp[5] * z[u][2]);
                if (MODE == 0) {
#pragma unroll
                    for (int q = 0; q < 6; ++q) { u0 += jc[q] * vc[q]; u1 += jc[6 + q] * vc[q]; }
                }
#pragma unroll
                for (int q = 0; q < 6; ++q) a[q] += jc[q] * u0 + jc[6 + q] * u1;
            } else {
                const double vx = X[u][0] - t[9], vy = X[u][1] - t[10], vz = X[u][2] - t[11];
                const double qx = t[0] * vx + t[1] * vy + t[2] * vz;
                const double qy = t[3] * vx + t[4] * vy + t[5] * vz;
                const double qz = t[6] * vx + t[7] * vy + t[8] * vz;
                const double px = K.k[0] * qx + K.k[1] * qy + K.k[2] * qz;
                const double py = K.k[3] * qx + K.k[4] * qy + K.k[5] * qz;
                const double pz = K.k[6] * qx + K.k[7] * qy + K.k[8] * qz;
                const double iz = 1.0 / pz;
                double h0 = z[u][0], h1 = z[u][1], h2 = z[u][2];
                if (MODE == 0) {
                    h0 += vy * ap[2] - vz * ap[1] + vT[0];
                    h1 += vz * ap[0] - vx * ap[2] + vT[1];
                    h2 += vx * ap[1] - vy * ap[0] + vT[2];
                }
                double s0 = 0.0, s1 = 0.0, s2 = 0.0;             // q = sum_k u_k j_k
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const double pk = (k == 0 ? px : py) * iz;
                    const double b0 = (K.k[3 * k + 0] - pk * K.k[6]) * iz;
                    const double b1 = (K.k[3 * k + 1] - pk * K.k[7]) * iz;
                    const double b2 = (K.k[3 * k + 2] - pk * K.k[8]) * iz;
                    const double j0 = b0 * t[0] + b1 * t[3] + b2 * t[6];
                    const double j1 = b0 * t[1] + b1 * t[4] + b2 * t[7];
                    const double j2 = b0 * t[2] + b1 * t[5] + b2 * t[8];
                    const double uk = -(j0 * h0 + j1 * h1 + j2 * h2);
                    s0 += uk * j0; s1 += uk * j1; s2 += uk * j2;
                }
                a[0] += s1 * vz - s2 * vy; a[1] += s2 * vx - s0 * vz; a[2] += s0 * vy - s1 * vx;       // q x v
                a[3] += s0; a[4] += s1; a[5] += s2;
            }
        }
    }
    // local form: the camera's own state, requested by the six threads that need it behind the reduction (after
    // the sweep, so that it does not occupy registers across it: the kernel has to stay at four workgroups per CU)
    double l_u = 0.0, l_dc = 0.0, l_s = 0.0, l_p = 0.0, l_x = 0.0, l_r = 0.0, l_m[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    double l_rz = 0.0, l_rzp = 1.0, l_ap = 0.0;
    int l_it = 0;
    if (local && threadIdx.x < 6) {
        const int k = threadIdx.x;
        const size_t e = (size_t)k * C + ch.x;
        l_u = vin[e]; l_dc = pl.Dc[e];
        l_s = pl.vecs[kPcgS * n6l + e]; l_p = pl.vecs[kPcgP * n6l + e];
        l_x = pl.vecs[kPcgX * n6l + e]; l_r = pl.vecs[kPcgR * n6l + e];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int a_ = k < j ? k : j, b_ = k < j ? j : k;
            l_m[j] = pl.Minv[(size_t)(a_ * 6 - a_ * (a_ - 1) / 2 + (b_ - a_)) * C + ch.x];     // row k of the packed upper triangle
        }
        l_rz = ctrl_done->rz; l_rzp = ctrl_done->rz_prev; l_ap = ctrl_done->alpha_prev; l_it = ctrl_done->iters;
    }
    const double s = cam_block_total<6>(a, red);
    double out = s;
    if (!ROUND) {                                        // rotation Jacobian applied to the summed M (see above)
        __shared__ double tot[6];
        if (threadIdx.x < 6) tot[threadIdx.x] = s;
        __syncthreads();
        if (threadIdx.x < 3) {
            const double wx = t[12], wy = t[13], wz = t[14];
            const double m0 = tot[0], m1 = tot[1], m2 = tot[2];
            const double c0 = m1 * wz - m2 * wy, c1 = m2 * wx - m0 * wz, c2 = m0 * wy - m1 * wx;          // M x w
            const double d0 = c1 * wz - c2 * wy, d1 = c2 * wx - c0 * wz, d2 = c0 * wy - c1 * wx;          // (M x w) x w
            const double r0 = -(m0 - t[15] * c0 + t[16] * d0), r1 = -(m1 - t[15] * c1 + t[16] * d1),
                         r2 = -(m2 - t[15] * c2 + t[16] * d2);
            out = threadIdx.x == 0 ? r0 : (threadIdx.x == 1 ? r1 : r2);
        } else if (threadIdx.x < 6) {
            out = -tot[threadIdx.x];
        }
    }
    if (MODE == 0 && cx.world > 1 && threadIdx.x < 6)        // (every camera is a single chunk in this form)
        out = cam_exchange_value(cx, cam_slots_pcg(cx.world, C, l_it & 1), 6, threadIdx.x, C, ch.x, out, 2u);
    if (threadIdx.x < 6) {
        if (ch.w == 1) acc[(size_t)threadIdx.x * C + ch.x] = out;
        else partial[(size_t)blockIdx.x * 6 + threadIdx.x] = out;
    }
    if (local) {                                            // (every camera is a single chunk in this form)
        __shared__ double ts[6];
        const int k = threadIdx.x;
        double w = 0.0, sk = 0.0;
        if (k < 6) {
            const double beta = l_it == 0 ? 0.0 : l_rz / l_rzp;
            w = out + l_dc * l_u;                           // (S u)_k
            l_x += l_ap * l_p;                              // the deferred updates of the previous iteration
            l_r -= l_ap * l_s;
            sk = w + beta * l_s;
            l_p = l_u + beta * l_p;
            ts[k] = sk;
        }
        __syncthreads();
        if (threadIdx.x < 64) {                             // wave 0: lanes 0..5 carry the camera, the others zeros
            double mk = 0.0;
            if (k < 6) {
#pragma unroll
                for (int j = 0; j < 6; ++j) mk += l_m[j] * ts[j];
            }
            const double d0 = wave_sum(k < 6 ? w * l_u : 0.0), d1 = wave_sum(k < 6 ? sk * l_u : 0.0);
            const double d2 = wave_sum(k < 6 ? sk * mk : 0.0), d3 = wave_sum(k < 6 ? l_r * l_u : 0.0);
            if (k < 6) {
                const size_t e = (size_t)k * C + ch.x;
                pl.vecs[kPcgS * n6l + e] = sk; pl.vecs[kPcgP * n6l + e] = l_p; pl.vecs[kPcgR * n6l + e] = l_r;
                pl.vecs[kPcgM * n6l + e] = mk;
                pl.vecs[kPcgX * n6l + e] = l_x;
                pl.vecs[(size_t)kPcgVecs * n6l + kPcgX * n6l + e] = l_x;        // x lives in both sets (k_backsub)
            }
            if (k < 4) pl.part[(size_t)k * C + ch.x] = k == 0 ? d0 : (k == 1 ? d1 : (k == 2 ? d2 : d3));
        }
    }
}

// Pass B over the XCD-aware chunk table (many points: every camera's list cut at the eight point-range boundaries), ONE
// WAVE PER CHUNK.  A chunk there is short (a camera's ~2000 observations / 8 = 250): as a 256-thread workgroup it used
// one of its four unroll slots, and paid two barriers and an LDS round trip for the six sums of 250 terms.  Here the
// four waves of a workgroup take four chunks of the SAME point range (cameras 4 g .. 4 g + 3, range k: workgroup 8 g + k,
// which runs on XCD k), each wave walks its chunk with four gathers per lane in flight and reduces with DPP / readlane
// alone; no LDS, no barrier.  Row q = (8 g + k) 4 + j of `partial` holds the sums of camera 4 g + j over range k;
// k_cam_combine_w adds a camera's eight rows in range order.
template <bool MIXED>
__global__ __launch_bounds__(kCamThreads) void k_cam_schur_w(CamMajor cm, const double* __restrict__ camtab,
                                                             const double* __restrict__ rec, KMat K,
                                                             const double* __restrict__ vin, int C,
                                                             double* __restrict__ partial,
                                                             const PcgCtrl* __restrict__ ctrl_done, int set, MixedB mxb) {
    if (ctrl_done != nullptr) {
        if (ctrl_done->done != 0) return;                       // grid-uniform
        vin += (size_t)((set < 0 ? (ctrl_done->iters & 1) : set) * kPcgVecs + kPcgU) * 6 * C;
    }
    const int lane = threadIdx.x & 63;
    const int q = __builtin_amdgcn_readfirstlane((int)blockIdx.x * kWaveChunkCams + ((int)threadIdx.x >> 6));   // wave-uniform
    const int4 ch = cm.chunks[q];
    if (ch.x < 0) return;                                       // padding behind the last camera
    double t[kCamTab];
#pragma unroll
    for (int k = 0; k < kCamTab; ++k) t[k] = camtab[(size_t)ch.x * kCamRow + k];      // wave-uniform: scalar loads
    double vc[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) vc[k] = vin[(size_t)k * C + ch.x];
    double ap[3];
    {
        const double wx = t[12], wy = t[13], wz = t[14];
        const double c0 = wy * vc[2] - wz * vc[1], c1 = wz * vc[0] - wx * vc[2], c2 = wx * vc[1] - wy * vc[0];
        const double d0 = wy * c2 - wz * c1, d1 = wz * c0 - wx * c2, d2 = wx * c1 - wy * c0;
        ap[0] = vc[0] - t[15] * c0 + t[16] * d0; ap[1] = vc[1] - t[15] * c1 + t[16] * d1; ap[2] = vc[2] - t[15] * c2 + t[16] * d2;
    }
    double vT[3] = {vc[3], vc[4], vc[5]};
    if (MIXED) {
#pragma unroll
        for (int k = 0; k < kRt32; ++k) t[k] = mxb.rtd[(size_t)ch.x * kRt32 + k];
#pragma unroll
        for (int k = 0; k < 3; ++k) { ap[k] = (double)(float)ap[k]; vT[k] = (double)(float)vT[k]; }
    }
    double a[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int k0 = ch.y + lane; k0 < ch.z; k0 += 64 * kCamUnroll) {
        int p[kCamUnroll];
        double X[kCamUnroll][3], z[kCamUnroll][3];
#pragma unroll
        for (int u = 0; u < kCamUnroll; ++u) {
            const int k = k0 + u * 64;
            p[u] = k < ch.z ? cm.pt[k] : -1;
        }
#pragma unroll
        for (int u = 0; u < kCamUnroll; ++u) {
            if (MIXED) {
                const float* __restrict__ rp = reinterpret_cast<const float*>(rec) + (size_t)kRec32 * (p[u] < 0 ? 0 : p[u]);
                const float4 r0 = *reinterpret_cast<const float4*>(rp);
                const float2 r1 = *reinterpret_cast<const float2*>(rp + 4);
                X[u][0] = r0.x; X[u][1] = r0.y; X[u][2] = r0.z;
                z[u][0] = r0.w; z[u][1] = r1.x; z[u][2] = r1.y;
            } else {
                const double2* __restrict__ rp = reinterpret_cast<const double2*>(rec + (size_t)kRec * (p[u] < 0 ? 0 : p[u]));
                const double2 r0 = rp[0], r1 = rp[1], r2 = rp[2];
                X[u][0] = r0.x; X[u][1] = r0.y; X[u][2] = r1.x;
                z[u][0] = r1.y; z[u][1] = r2.x; z[u][2] = r2.y;
            }
        }
#pragma unroll
        for (int u = 0; u < kCamUnroll; ++u) {
            if (p[u] < 0) continue;
            const double vx = X[u][0] - t[9], vy = X[u][1] - t[10], vz = X[u][2] - t[11];
            const double qx = t[0] * vx + t[1] * vy + t[2] * vz;
            const double qy = t[3] * vx + t[4] * vy + t[5] * vz;
            const double qz = t[6] * vx + t[7] * vy + t[8] * vz;
            const double px = K.k[0] * qx + K.k[1] * qy + K.k[2] * qz;
            const double py = K.k[3] * qx + K.k[4] * qy + K.k[5] * qz;
            const double pz = K.k[6] * qx + K.k[7] * qy + K.k[8] * qz;
            const double iz = 1.0 / pz;
            const double h0 = z[u][0] + vy * ap[2] - vz * ap[1] + vT[0];
            const double h1 = z[u][1] + vz * ap[0] - vx * ap[2] + vT[1];
            const double h2 = z[u][2] + vx * ap[1] - vy * ap[0] + vT[2];
            double s0 = 0.0, s1 = 0.0, s2 = 0.0;             // q = sum_k u_k j_k
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const double pk = (k == 0 ? px : py) * iz;
                const double b0 = (K.k[3 * k + 0] - pk * K.k[6]) * iz;
                const double b1 = (K.k[3 * k + 1] - pk * K.k[7]) * iz;
                const double b2 = (K.k[3 * k + 2] - pk * K.k[8]) * iz;
                const double j0 = b0 * t[0] + b1 * t[3] + b2 * t[6];
                const double j1 = b0 * t[1] + b1 * t[4] + b2 * t[7];
                const double j2 = b0 * t[2] + b1 * t[5] + b2 * t[8];
                const double uk = -(j0 * h0 + j1 * h1 + j2 * h2);
                s0 += uk * j0; s1 += uk * j1; s2 += uk * j2;
            }
            a[0] += s1 * vz - s2 * vy; a[1] += s2 * vx - s0 * vz; a[2] += s0 * vy - s1 * vx;       // q x v
            a[3] += s0; a[4] += s1; a[5] += s2;
        }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) a[k] = wave_sum(a[k]);        // on every lane
    if (lane < 6) {
        const double wx = t[12], wy = t[13], wz = t[14];
        const double m0 = a[0], m1 = a[1], m2 = a[2];
        const double c0 = m1 * wz - m2 * wy, c1 = m2 * wx - m0 * wz, c2 = m0 * wy - m1 * wx;          // M x w
        const double d0 = c1 * wz - c2 * wy, d1 = c2 * wx - c0 * wz, d2 = c0 * wy - c1 * wx;          // (M x w) x w
        const double r0 = -(m0 - t[15] * c0 + t[16] * d0), r1 = -(m1 - t[15] * c1 + t[16] * d1),
                     r2 = -(m2 - t[15] * c2 + t[16] * d2);
        const double out = lane == 0 ? r0 : (lane == 1 ? r1 : (lane == 2 ? r2 : (lane == 3 ? -a[3] : (lane == 4 ? -a[4] : -a[5]))));
        partial[(size_t)q * 6 + lane] = out;
    }
}
// out[c * cs + col * ks] = sum over the eight ranges of camera c's rows (NC values each) of a wave-per-chunk pass, in
// range order (pass B: NC = 6, acc plane-major; K3: 27, [U | g_c] camera-major; rhs pass: 27, acc | sd plane-major)
template <int NC>
__global__ __launch_bounds__(256) void k_cam_combine_w(const double* __restrict__ partial, int C, double* __restrict__ out,
                                                       int cs, int ks, const int* __restrict__ done,
                                                       const double* __restrict__ skip) {
    if (done != nullptr && *done != 0) return;
    if (skip != nullptr && *skip != 0.0) return;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= C * NC) return;
    const int c = e / NC, col = e - c * NC;
    const int g = c / kWaveChunkCams, j = c - g * kWaveChunkCams;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < kWaveChunkRanges; ++k)
        s += partial[(size_t)((g * kWaveChunkRanges + k) * kWaveChunkCams + j) * NC + col];
    out[(size_t)c * cs + (size_t)col * ks] = s;
}

// Reduced right-hand side WITH the diagonal blocks of W Vinv W^T (Schur-diagonal preconditioner): one camera-major
// pass yields, per camera,  out[0..5] = -sum_i W_i e_p  (what k_cam_schur MODE 1 computes) and
// out[6..26] = sum_i W_i Vinv_p W_i^T  (packed upper triangle), W_i = Jc_i^T Jp_i.  Every W_i is formed anyway for the
// first sum; the second costs one more gathered row per observation (Vinv_p, 48 bytes) and no pass of its own.
// Output plane-major [27][C] (acc | sd: one contiguous vector for the all-reduce) or partial[chunk][27].
// 192-thread workgroups: the kernel needs ~150 VGPRs (three waves per SIMD; the record of the next trip is in flight
// while the current one is worked on), and four 3-wave workgroups per CU keep 1024 camera workgroups resident at once
// where three 4-wave ones would take a 1000-camera problem in two rounds.
#ifndef SFMBA_RHS_THREADS
#define SFMBA_RHS_THREADS 192
#endif
constexpr int kRhsThreads = SFMBA_RHS_THREADS;
// minv (one rank, every camera a single chunk): the workgroup also inverts its camera's preconditioner block
// (U + Dc - sd), which is complete the moment its 27 sums are -- k_cam_prep_schur's work without its launch.
struct RhsPrecond {
    const double* __restrict__ Ugc;      // null: Minv is made by k_cam_prep_schur (several ranks or chunks)
    const double* __restrict__ Dc;
    double* __restrict__ Minv;
};
template <bool ROUND>
__global__ __launch_bounds__(kRhsThreads) void k_cam_rhs_diag(CamMajor cm, const double* __restrict__ camtab,
                                                              const double* __restrict__ rec,
                                                              const double* __restrict__ Vinv, KMat K, int C,
                                                              double* __restrict__ out, double* __restrict__ partial,
                                                              RhsPrecond mp, const double* __restrict__ rhsrec, CamExchange cx) {
    __shared__ double red[kRhsThreads / 64][27];
    const int4 ch = cm.chunks[blockIdx.x];
    double t[kCamTab];
#pragma unroll
    for (int k = 0; k < kCamTab; ++k) t[k] = camtab[(size_t)ch.x * kCamRow + k];      // wave-uniform: scalar loads
    double a[27];
#pragma unroll
    for (int q = 0; q < 27; ++q) a[q] = 0.0;
    // One observation per lane and trip (27 sums + two rows of a 6x3 block pair in registers: no room to unroll), software
    // pipelined: the 1000 camera workgroups of a launch start together and walk in lockstep, so a trip whose index load,
    // record gather and arithmetic follow each other costs the two latencies SIX times per launch (a camera's ~1000
    // observations / 192 lanes) with nothing to hide them behind -- 33 us at cfg4, of which 13 are arithmetic.  Here the
    // index is fetched two trips ahead and the record (120 of its 128 bytes: X | e | Vinv) one trip ahead, i.e. requested
    // before the current trip's ~250 instructions issue.  Same observations per lane in the same order: same bits.
    auto gather = [&](int pidx, double* Xo, double* eo, double* vo) {
        const size_t pp = (size_t)(pidx < 0 ? 0 : pidx);
        // one 128-byte record per observation (kRhsRec) when k_prep has written them, else the two gathers
        const double2* __restrict__ rp = reinterpret_cast<const double2*>(rhsrec != nullptr ? rhsrec + kRhsRec * pp : rec + kRec * pp);
        const double2* __restrict__ vp = rhsrec != nullptr ? rp + 4 : reinterpret_cast<const double2*>(Vinv + kVinvRow * pp);
        const double2 r0 = rp[0], r1 = rp[1], r2 = rp[2], v0 = vp[0], v1 = vp[1], v2 = vp[2];
        Xo[0] = r0.x; Xo[1] = r0.y; Xo[2] = r1.x;
        eo[0] = r1.y; eo[1] = r2.x; eo[2] = r2.y;
        vo[0] = v0.x; vo[1] = v0.y; vo[2] = v1.x; vo[3] = v1.y; vo[4] = v2.x; vo[5] = v2.y;
    };
    const int k_first = ch.y + (int)threadIdx.x;
    int p_cur = k_first < ch.z ? cm.pt[k_first] : -1;
    int p_nxt = k_first + kRhsThreads < ch.z ? cm.pt[k_first + kRhsThreads] : -1;
    double X[3], e[3], vi[6];
    gather(p_cur, X, e, vi);
    for (int k0 = ch.y; k0 < ch.z; k0 += kRhsThreads) {            // (workgroup-uniform trip count)
        const int k2 = k0 + (int)threadIdx.x + 2 * kRhsThreads;
        const int p_n2 = k2 < ch.z ? cm.pt[k2] : -1;
        double Xn[3], en[3], vn[6];
        gather(p_nxt, Xn, en, vn);
        if (p_cur >= 0) {
            double jc[12], jp[6], rx, ry;
            observe<true>(t, X[0], X[1], X[2], 0.0, 0.0, K, rx, ry, jc, jp);
            if (ROUND) {                           // as stored in fp32 (see k_cam_schur)
#pragma unroll
                for (int q = 0; q < 3; ++q) { jc[q] = (double)(float)jc[q]; jc[6 + q] = (double)(float)jc[6 + q]; }
#pragma unroll
                for (int q = 0; q < 6; ++q) jp[q] = (double)(float)jp[q];
#pragma unroll
                for (int q = 0; q < 3; ++q) { jc[3 + q] = -jp[q]; jc[9 + q] = -jp[3 + q]; }
            }
            // W = Jc^T Jp has rank two (the two residual rows), so neither W nor W Vinv is formed:
            //     -W e          = -Jc^T (Jp e)
            //     W Vinv W^T    = Jc^T G Jc,   G = Jp Vinv Jp^T  (2 x 2, symmetric)
            // 111 multiply-adds per observation instead of 171, and no 6 x 3 temporaries in registers
            const double s0 = jp[0] * e[0] + jp[1] * e[1] + jp[2] * e[2];
            const double s1 = jp[3] * e[0] + jp[4] * e[1] + jp[5] * e[2];
            const double h00 = vi[0] * jp[0] + vi[1] * jp[1] + vi[2] * jp[2];      // Vinv jp_0 (Vinv packed upper)
            const double h01 = vi[1] * jp[0] + vi[3] * jp[1] + vi[4] * jp[2];
            const double h02 = vi[2] * jp[0] + vi[4] * jp[1] + vi[5] * jp[2];
            const double h10 = vi[0] * jp[3] + vi[1] * jp[4] + vi[2] * jp[5];      // Vinv jp_1
            const double h11 = vi[1] * jp[3] + vi[3] * jp[4] + vi[4] * jp[5];
            const double h12 = vi[2] * jp[3] + vi[4] * jp[4] + vi[5] * jp[5];
            const double g00 = jp[0] * h00 + jp[1] * h01 + jp[2] * h02;
            const double g01 = jp[0] * h10 + jp[1] * h11 + jp[2] * h12;
            const double g11 = jp[3] * h10 + jp[4] * h11 + jp[5] * h12;
            double m0[6], m1[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                a[i] -= jc[i] * s0 + jc[6 + i] * s1;
                m0[i] = g00 * jc[i] + g01 * jc[6 + i];
                m1[i] = g01 * jc[i] + g11 * jc[6 + i];
            }
            int n = 6;
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = i; j < 6; ++j) a[n++] += jc[i] * m0[j] + jc[6 + i] * m1[j];
        }
        p_cur = p_nxt; p_nxt = p_n2;
#pragma unroll
        for (int q = 0; q < 3; ++q) { X[q] = Xn[q]; e[q] = en[q]; }
#pragma unroll
        for (int q = 0; q < 6; ++q) vi[q] = vn[q];
    }
    double s = cam_block_total<27, kRhsThreads / 64>(a, red);
    if (threadIdx.x < 27) {
        if (cx.world > 1) s = cam_exchange_value(cx, cam_slots_rhs(cx.world, C), 27, threadIdx.x, C, ch.x, s, 4u);
        if (ch.w == 1) out[(size_t)threadIdx.x * C + ch.x] = s;
        else partial[(size_t)blockIdx.x * 27 + threadIdx.x] = s;
    }
    if (mp.Ugc != nullptr) {                                  // (ch.w == 1 for every camera on this path)
        __shared__ double sdl[21];
        if (threadIdx.x >= 6 && threadIdx.x < 27) sdl[threadIdx.x - 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            double B[6][6], inv[21];
            int n = 0;
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = i; j < 6; ++j) {
                    B[i][j] = mp.Ugc[(size_t)ch.x * 27 + n] - sdl[n];
                    B[j][i] = B[i][j];
                    ++n;
                }
#pragma unroll
            for (int i = 0; i < 6; ++i) B[i][i] += mp.Dc[(size_t)i * C + ch.x];
            spd6_inverse(B, inv);
#pragma unroll
            for (int q = 0; q < 21; ++q) mp.Minv[(size_t)q * C + ch.x] = inv[q];
        }
    }
}

// The rhs + preconditioner pass over the XCD-aware chunk table, one wave per chunk (see k_cam_blocks_w): row q of
// `partial` = the chunk's 27 sums (6 of -W e, 21 of W Vinv W^T); k_cam_combine_w<27> adds the eight rows of a camera into
// acc | sd and k_cam_prep_schur inverts the preconditioner blocks.
template <bool ROUND>
__global__ __launch_bounds__(kCamThreads) void k_cam_rhs_diag_w(CamMajor cm, const double* __restrict__ camtab,
                                                                const double* __restrict__ rec,
                                                                const double* __restrict__ Vinv, KMat K,
                                                                double* __restrict__ partial,
                                                                const double* __restrict__ rhsrec) {
    const int lane = threadIdx.x & 63;
    const int q = __builtin_amdgcn_readfirstlane((int)blockIdx.x * kWaveChunkCams + ((int)threadIdx.x >> 6));   // wave-uniform
    const int4 ch = cm.chunks[q];
    if (ch.x < 0) return;                                       // padding behind the last camera
    double t[kCamTab];
#pragma unroll
    for (int k = 0; k < kCamTab; ++k) t[k] = camtab[(size_t)ch.x * kCamRow + k];      // wave-uniform: scalar loads
    double a[27];
#pragma unroll
    for (int n = 0; n < 27; ++n) a[n] = 0.0;
    auto gather = [&](int pidx, double* Xo, double* eo, double* vo) {
        const size_t pp = (size_t)(pidx < 0 ? 0 : pidx);
        const double2* __restrict__ rp = reinterpret_cast<const double2*>(rhsrec != nullptr ? rhsrec + kRhsRec * pp : rec + kRec * pp);
        const double2* __restrict__ vp = rhsrec != nullptr ? rp + 4 : reinterpret_cast<const double2*>(Vinv + kVinvRow * pp);
        const double2 r0 = rp[0], r1 = rp[1], r2 = rp[2], v0 = vp[0], v1 = vp[1], v2 = vp[2];
        Xo[0] = r0.x; Xo[1] = r0.y; Xo[2] = r1.x;
        eo[0] = r1.y; eo[1] = r2.x; eo[2] = r2.y;
        vo[0] = v0.x; vo[1] = v0.y; vo[2] = v1.x; vo[3] = v1.y; vo[4] = v2.x; vo[5] = v2.y;
    };
    const int k_first = ch.y + lane;
    int p_cur = k_first < ch.z ? cm.pt[k_first] : -1;
    int p_nxt = k_first + 64 < ch.z ? cm.pt[k_first + 64] : -1;
    double X[3], e[3], vi[6];
    gather(p_cur, X, e, vi);
    for (int k0 = ch.y; k0 < ch.z; k0 += 64) {                     // (wave-uniform trip count; software pipelined as k_cam_rhs_diag)
        const int k2 = k0 + lane + 128;
        const int p_n2 = k2 < ch.z ? cm.pt[k2] : -1;
        double Xn[3], en[3], vn[6];
        gather(p_nxt, Xn, en, vn);
        if (p_cur >= 0) {
            double jc[12], jp[6], rx, ry;
            observe<true>(t, X[0], X[1], X[2], 0.0, 0.0, K, rx, ry, jc, jp);
            if (ROUND) {                           // as stored in fp32 (see k_cam_schur)
#pragma unroll
                for (int n = 0; n < 3; ++n) { jc[n] = (double)(float)jc[n]; jc[6 + n] = (double)(float)jc[6 + n]; }
#pragma unroll
                for (int n = 0; n < 6; ++n) jp[n] = (double)(float)jp[n];
#pragma unroll
                for (int n = 0; n < 3; ++n) { jc[3 + n] = -jp[n]; jc[9 + n] = -jp[3 + n]; }
            }
            const double s0 = jp[0] * e[0] + jp[1] * e[1] + jp[2] * e[2];
            const double s1 = jp[3] * e[0] + jp[4] * e[1] + jp[5] * e[2];
            const double h00 = vi[0] * jp[0] + vi[1] * jp[1] + vi[2] * jp[2];      // Vinv jp_0 (Vinv packed upper)
            const double h01 = vi[1] * jp[0] + vi[3] * jp[1] + vi[4] * jp[2];
            const double h02 = vi[2] * jp[0] + vi[4] * jp[1] + vi[5] * jp[2];
            const double h10 = vi[0] * jp[3] + vi[1] * jp[4] + vi[2] * jp[5];      // Vinv jp_1
            const double h11 = vi[1] * jp[3] + vi[3] * jp[4] + vi[4] * jp[5];
            const double h12 = vi[2] * jp[3] + vi[4] * jp[4] + vi[5] * jp[5];
            const double g00 = jp[0] * h00 + jp[1] * h01 + jp[2] * h02;
            const double g01 = jp[0] * h10 + jp[1] * h11 + jp[2] * h12;
            const double g11 = jp[3] * h10 + jp[4] * h11 + jp[5] * h12;
            double m0[6], m1[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                a[i] -= jc[i] * s0 + jc[6 + i] * s1;
                m0[i] = g00 * jc[i] + g01 * jc[6 + i];
                m1[i] = g01 * jc[i] + g11 * jc[6 + i];
            }
            int n = 6;
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = i; j < 6; ++j) a[n++] += jc[i] * m0[j] + jc[6 + i] * m1[j];
        }
        p_cur = p_nxt; p_nxt = p_n2;
#pragma unroll
        for (int n = 0; n < 3; ++n) { X[n] = Xn[n]; e[n] = en[n]; }
#pragma unroll
        for (int n = 0; n < 6; ++n) vi[n] = vn[n];
    }
    const int col = wave_fold_sum<27>(a, lane);
    if (col >= 0) partial[(size_t)q * 27 + col] = a[0];
}

// ---------------------------------------------------------------------------------------------
// Few cameras (6 C <= kDenseMaxN): the size the reference itself produces -- SceauxCastle has 11 cameras
// (BASELINE.json configs[0-1]).  There the implicit product is all launch latency (two launches per PCG iteration,
// ~10 iterations per outer iteration), while the reduced camera matrix S = U + Dc - W V^-1 W^T is 66 x 66.
// So S is FORMED -- one workgroup per 6x6 block pair (a <= b) over the list of points both cameras see, built once
// per problem; blocks recomputed from the camera rows and the gathered point; no atomics -- and the SAME
// preconditioned conjugate gradients run inside one workgroup with S in LDS: an iteration is a 66 x 66
// matrix-vector product by 512 threads and one block reduction, ~3000 cycles instead of two launches.
// The preconditioner is the inverse of the diagonal 6x6 blocks of S, i.e. exactly the Schur-diagonal
// preconditioner of the implicit path: both paths walk through the same iterates up to rounding.
// (An earlier version factorised S instead: blocked Cholesky with the trailing updates on v_mfma_f64_16x16x4_f64.  33 us at
// 66 unknowns -- the 16-step diagonal tiles and panel substitutions are serial code for one wavefront -- and its
// exact steps carried rounding noise along the seven gauge directions that no camera being fixed leaves open.)
// ---------------------------------------------------------------------------------------------
constexpr int kDenseMaxN = 128;                  // 6 C <= 128: C <= 21
__host__ __device__ constexpr int dense_block_index(int a, int b, int C) {      // a <= b, row-major upper triangle
    return a * C - a * (a - 1) / 2 + (b - a);
}

// blk[a][b] (6x6, row-major) = sum over the points p seen by cameras a and b (with multiplicity) of
// W_a(p) Vinv_p W_b(p)^T,  W_c(p) = Jc^T Jp of camera c at point p.  The workgroup of a diagonal pair (a, a) also
// sums the reduced right-hand-side term of its camera, rhs_a = -sum W_a(p) e_p over the list entries that pair an
// observation with itself (marked ~p by set_problem; e_p in the z half of the point records), into acc[k][a]
// (acc == null: not wanted) -- the pass k_cam_schur<1> would make over the same observations, without its launch.
__global__ __launch_bounds__(kCamThreads) void k_schur_blocks(const int* __restrict__ cov_ptr, const int* __restrict__ cov_pt,
                                                     const int2* __restrict__ blk_ab, const double* __restrict__ camtab,
                                                     const double* __restrict__ rec, const double* __restrict__ Vinv,
                                                     KMat K, int C, double* __restrict__ Sblk, double* __restrict__ acc) {
    __shared__ double red[kCamWaves][36];
    __shared__ double red6[kCamWaves][6];
    const int2 ab = blk_ab[blockIdx.x];
    const bool diag = ab.x == ab.y, rhs = diag && acc != nullptr;          // workgroup-uniform
    double ta[kCamTab], tb[kCamTab];
#pragma unroll
    for (int k = 0; k < kCamTab; ++k) { ta[k] = camtab[(size_t)ab.x * kCamRow + k]; tb[k] = camtab[(size_t)ab.y * kCamRow + k]; }
    double s[36], g[6];
#pragma unroll
    for (int q = 0; q < 36; ++q) s[q] = 0.0;
#pragma unroll
    for (int q = 0; q < 6; ++q) g[q] = 0.0;
    auto w_of = [&](const double* t, double X, double Y, double Z, double (&W)[6][3]) {
        double jc[12], jp[6], rx, ry;
        observe<true>(t, X, Y, Z, 0.0, 0.0, K, rx, ry, jc, jp);
#pragma unroll
        for (int u = 0; u < 6; ++u)
#pragma unroll
            for (int v = 0; v < 3; ++v) W[u][v] = jc[u] * jp[v] + jc[6 + u] * jp[3 + v];
    };
    for (int k = cov_ptr[blockIdx.x] + (int)threadIdx.x; k < cov_ptr[blockIdx.x + 1]; k += kCamThreads) {
        const int pm = cov_pt[k];
        const bool self = pm < 0;                                  // an observation paired with itself
        const int p = self ? ~pm : pm;
        const double2* __restrict__ rp = reinterpret_cast<const double2*>(rec + kRec * (size_t)p);   // X Y | Z e0 | e1 e2
        const double* __restrict__ vi = Vinv + kVinvRow * (size_t)p;
        const double2 r0 = rp[0], r1 = rp[1], r2 = rp[2];
        const double X = r0.x, Y = r0.y, Z = r1.x;
        const double v0 = vi[0], v1 = vi[1], v2 = vi[2], v3 = vi[3], v4 = vi[4], v5 = vi[5];
        double Wa[6][3], Wb[6][3];
        w_of(ta, X, Y, Z, Wa);
        if (!diag) w_of(tb, X, Y, Z, Wb);
        else {
#pragma unroll
            for (int u = 0; u < 6; ++u)
#pragma unroll
                for (int v = 0; v < 3; ++v) Wb[u][v] = Wa[u][v];
        }
        if (rhs && self) {
#pragma unroll
            for (int u = 0; u < 6; ++u) g[u] -= Wa[u][0] * r1.y + Wa[u][1] * r2.x + Wa[u][2] * r2.y;
        }
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const double y0 = Wa[u][0] * v0 + Wa[u][1] * v1 + Wa[u][2] * v2;      // row u of W_a Vinv (Vinv packed upper)
            const double y1 = Wa[u][0] * v1 + Wa[u][1] * v3 + Wa[u][2] * v4;
            const double y2 = Wa[u][0] * v2 + Wa[u][1] * v4 + Wa[u][2] * v5;
#pragma unroll
            for (int v = 0; v < 6; ++v) s[6 * u + v] += y0 * Wb[v][0] + y1 * Wb[v][1] + y2 * Wb[v][2];
        }
    }
    const double tot = cam_block_total<36>(s, red);              // lanes, then the four waves in wave order
    if (threadIdx.x < 36) Sblk[(size_t)blockIdx.x * 36 + threadIdx.x] = tot;
    if (rhs) {
        const double tg = cam_block_total<6>(g, red6);
        if (threadIdx.x < 6) acc[(size_t)threadIdx.x * C + ab.x] = tg;
    }
}

// Solve (U + Dc - blk) dc = -g_c - acc by block-preconditioned CG in LDS (see above).  One workgroup of 512
// threads; S full and symmetric, leading dimension n | 1; four lanes per row of the matrix-vector product, the
// first of them owns x, r, p, s of its unknown in registers.  The recurrences are those of the implicit PCG below
// (single reduction per iteration, Chronopoulos-Gear):
//     u = M^-1 r;  w = S u;  gamma = r.u;  delta = w.u;  beta = gamma / gamma_prev;
//     alpha = gamma / (delta - beta gamma / alpha_prev);  p = u + beta p;  s = w + beta s;  x += alpha p;  r -= alpha s
// so an iteration is three barriers and four LDS round trips; the lane's 32 entries of S stay in registers.
// (Measured with s_memtime at 66 unknowns: the first version -- textbook CG, 256 threads, S read from LDS in a
// 9-trip loop -- spent 5600 cycles per iteration, all of it exposed LDS latency.)  The step goes to x of both PCG vector sets and the control block reports
// iterations and outcome like the implicit PCG does (1 converged, 2 iteration cap, 3 breakdown: x is the last good
// iterate).  Every sum has a fixed order: same input, same bits.
constexpr int kDenseThreads = 512, kDenseLanes = 4;      // (1024 x 8: 128 registers per lane, spills; 256 x 2: slower)
__device__ __forceinline__ double dense_rcp(double d) {         // v_rcp_f64 + one Newton step
    const double y = __builtin_amdgcn_rcp(d);
    return fma(fma(-d, y, 1.0), y, y);
}
// The solve itself (the body of k_dense_pcg): A = the
// workgroup's LDS region (matrix | r u | 6x6 inverses | reduction slots), blk_load(e) = entry e of the block list,
// x_store(camera, k, value) takes the step; Ugc / Dc / acc may live in global memory or in LDS.  Every thread of the
// 512 returns the same control block.
template <class BlkLoad, class XStore>
__device__ __forceinline__ PcgCtrl dense_pcg_body(double* __restrict__ A, BlkLoad blk_load, const double* __restrict__ Ugc,
                                                  const double* __restrict__ Dc, const double* __restrict__ acc, int C,
                                                  double tol, int max_iters, XStore x_store) {
    constexpr int kWaves = kDenseThreads / 64;
    const int n = 6 * C, ld = n | 1;
    double* rv = A + (size_t)n * ld;            // behind the matrix: r, u, the 6x6 inverses, reduction slots
    double* uv = rv + kDenseMaxN;
    double* mi = uv + kDenseMaxN;               // [C][21] packed upper triangles, padded to a multiple of 16 bytes
    double* red = mi + ((21 * kDenseMaxN / 6 + 1) & ~1);      // [2 rounds][waves][gamma, delta]
    const int tid = threadIdx.x;
    const int row = tid / kDenseLanes, sub = tid & (kDenseLanes - 1);
    const bool owner = sub == 0 && row < n;
    auto at = [&](int r, int c) -> double& { return A[(size_t)r * ld + c]; };
    // entry e of the block list -> (a, b, u, v) without a table: blk = a C - a (a - 1) / 2 + (b - a), a <= b
    auto decode = [&](int e, int& a_, int& b_, int& u, int& v) {
        const int blk = e / 36, uvi = e - blk * 36;
        u = uvi / 6; v = uvi - 6 * u;
        const float t = (float)(2 * C + 1);
        int a = (int)((t - sqrtf(t * t - 8.0f * (float)blk)) * 0.5f);
        while (a > 0 && dense_block_index(a, a, C) > blk) --a;
        while (a + 1 < C && dense_block_index(a + 1, a + 1, C) <= blk) ++a;
        a_ = a; b_ = a + (blk - dense_block_index(a, a, C));
    };
    double x = 0.0, r = 0.0, pp = 0.0, ss = 0.0;
    if (owner) {                                                         // right-hand side
        const int c = row / 6, k = row - 6 * c;
        r = -Ugc[(size_t)c * 27 + 21 + k] - acc[(size_t)k * C + c];
        rv[row] = r;
    }
    {   // S from the blocks: entries (6a+u, 6b+v) and (6b+v, 6a+u), a <= b, are -blk[a][b][u][v] (+ U, Dc on the
        // diagonal blocks, whose half u <= v is mirrored); six entries per thread and trip with all their global
        // loads in flight together (the block list was written by the previous launch on other XCDs: every dependent
        // round trip costs a microsecond)
        const int nent = C * (C + 1) / 2 * 36;
        constexpr int kB = 6;
        for (int e0 = tid; e0 < nent; e0 += kB * kDenseThreads) {
            double val[kB], add[kB];
            int rr[kB], cc[kB];
#pragma unroll
            for (int q = 0; q < kB; ++q) {
                const int e = e0 + q * kDenseThreads;
                val[q] = 0.0; add[q] = 0.0; rr[q] = -1; cc[q] = 0;
                if (e < nent) {
                    int a, b, u, v;
                    decode(e, a, b, u, v);
                    if (a == b && u > v) continue;
                    val[q] = blk_load(e);
                    if (a == b) {
                        add[q] = Ugc[(size_t)a * 27 + (u * 6 - u * (u - 1) / 2 + (v - u))];
                        if (u == v) add[q] += Dc[(size_t)u * C + a];
                    }
                    rr[q] = 6 * a + u; cc[q] = 6 * b + v;
                }
            }
#pragma unroll
            for (int q = 0; q < kB; ++q) {
                if (rr[q] < 0) continue;
                const double sv = add[q] - val[q];
                at(rr[q], cc[q]) = sv;
                at(cc[q], rr[q]) = sv;
            }
        }
    }
    __syncthreads();
    if (tid < C) {                               // preconditioner: thread c inverts the diagonal block of camera c
        double B[6][6], out[21];                 // (from LDS: gathered from global memory, its 63 strided loads per
#pragma unroll                                   // lane cost the texture addresser more than the whole fill)
        for (int u = 0; u < 6; ++u)
#pragma unroll
            for (int v = 0; v < 6; ++v) B[u][v] = at(6 * tid + u, 6 * tid + v);
        spd6_inverse(B, out);
#pragma unroll
        for (int k = 0; k < 21; ++k) mi[21 * tid + k] = out[k];
    }
    // this lane's part of its row of S stays in registers for the whole solve
    double sv[kDenseMaxN / kDenseLanes];
#pragma unroll
    for (int q = 0; q < kDenseMaxN / kDenseLanes; ++q) {
        const int j = sub + q * kDenseLanes;
        sv[q] = (row < n && j < n) ? at(row, j) : 0.0;
    }
    __syncthreads();
    const double tol2 = tol * tol;
    double gamma = 0.0, gamma0 = 0.0, inv_gamma_prev = 1.0, inv_alpha_prev = 1.0;
    int it = 0, done = 0;
    const int cam = row / 6, kk = row - 6 * cam;
    for (;;) {
        // u = M^-1 r (r of the camera's six unknowns from LDS), published for the product
        double u = 0.0;
        if (owner) {
            u = minv_row(mi + 21 * cam, rv + 6 * cam, kk);
            uv[row] = u;
        }
        __syncthreads();
        // w = S u: four lanes per row, up to 32 independent products per lane (S from registers, u from LDS:
        // lanes with the same `sub` read the same words), the lanes add up by DPP
        double w = 0.0;
        if (row < n) {
            double s4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < kDenseMaxN / kDenseLanes; ++q) {
                const int j = sub + q * kDenseLanes;
                if (q * kDenseLanes < n) s4[q & 3] = fma(sv[q], uv[j < n ? j : 0], s4[q & 3]);      // (wave-uniform test)
            }
            w = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        }
        static_assert(kDenseLanes == 4, "the lanes of a row are one quad");
        w += dpp_double<0xB1>(w);                                        // quad_perm [1,0,3,2]
        w += dpp_double<0x4E>(w);                                        // quad_perm [2,3,0,1]
        // gamma = r.u, delta = w.u: waves in wave order
        {
            const double g = wave_sum(owner ? r * u : 0.0), d = wave_sum(owner ? w * u : 0.0);
            double* __restrict__ slot = red + (size_t)(it & 1) * 2 * kWaves;
            if ((tid & 63) == 0) { slot[2 * (tid >> 6)] = g; slot[2 * (tid >> 6) + 1] = d; }
            __syncthreads();
            double2 t2[4];                                               // four chains over the waves, fixed order
#pragma unroll
            for (int k = 0; k < 4; ++k) t2[k] = reinterpret_cast<const double2*>(slot)[k];
#pragma unroll
            for (int k = 4; k < kWaves; ++k) {
                const double2 v = reinterpret_cast<const double2*>(slot)[k];
                t2[k & 3].x += v.x; t2[k & 3].y += v.y;
            }
            t2[0].x = (t2[0].x + t2[1].x) + (t2[2].x + t2[3].x);
            t2[0].y = (t2[0].y + t2[1].y) + (t2[2].y + t2[3].y);
            const double ds = t2[0].y;
            gamma = t2[0].x;
            if (it == 0) {
                gamma0 = gamma;
                if (!(gamma0 > 0.0)) { done = gamma0 == 0.0 ? 1 : 3; break; }        // zero right-hand side: x = 0
            } else if (!(gamma > tol2 * gamma0)) { done = 1; break; }                // also catches NaN
            if (it >= max_iters) { done = 2; break; }
            // (reciprocals by v_rcp_f64 + one Newton step: three divisions were a quarter of the iteration)
            const double beta = it == 0 ? 0.0 : gamma * inv_gamma_prev;
            const double den = ds - (it == 0 ? 0.0 : beta * gamma * inv_alpha_prev);
            const double inv_den = dense_rcp(den);
            const double alpha = gamma * inv_den;
            if (!(den > 0.0) || !isfinite(alpha)) { done = 3; break; }               // S not positive definite / NaN
            if (owner) {
                pp = fma(beta, pp, u);
                ss = fma(beta, ss, w);
                x = fma(alpha, pp, x);
                r = fma(-alpha, ss, r);
                rv[row] = r;
            }
            inv_gamma_prev = dense_rcp(gamma); inv_alpha_prev = den * inv_gamma_prev;
            ++it;
        }
        __syncthreads();
    }
    if (owner) x_store(cam, kk, isfinite(x) ? x : 0.0);
    PcgCtrl c0;
    c0.rz = gamma; c0.rz0 = gamma0; c0.tol2 = tol2; c0.rz_prev = 1.0 / inv_gamma_prev; c0.alpha_prev = 1.0 / inv_alpha_prev;
    c0.iters = it; c0.max_iters = max_iters; c0.done = done; c0.pad = 2;
    return c0;
}

__global__ __launch_bounds__(kDenseThreads) void k_dense_pcg(const double* __restrict__ Sblk, const double* __restrict__ Ugc,
                                                             const double* __restrict__ Dc, const double* __restrict__ acc,
                                                             int C, double tol, int max_iters, double* __restrict__ vecs,
                                                             PcgCtrl* __restrict__ ctrl2) {
    extern __shared__ __align__(16) double A[];
    const int n = 6 * C;
    const PcgCtrl c0 = dense_pcg_body(
        A, [&](int e) { return Sblk[e]; }, Ugc, Dc, acc, C, tol, max_iters, [&](int cam, int kk, double xo) {
            vecs[kPcgX * (size_t)n + (size_t)kk * C + cam] = xo;                         // both vector sets: the back
            vecs[(kPcgVecs + kPcgX) * (size_t)n + (size_t)kk * C + cam] = xo;            // substitution picks one by count
        });
    if (threadIdx.x == 0) { ctrl2[0] = c0; ctrl2[1] = c0; }
}

// PCG on the reduced camera system S dc = rhs in the single-reduction (Chronopoulos-Gear) form, so
// that one sweep (w = S u) and ONE small update kernel make an iteration:
//     delta = (w,u); beta = gamma/gamma_prev; alpha = gamma / (delta - beta gamma / alpha_prev)
//     p = u + beta p; s = w + beta s; x += alpha p; r -= alpha s; u = Minv r; gamma' = (r,u)
// One thread per camera (its 6-vectors and 6x6 block are thread private).
//
// A single workgroup doing this is bound by one CU's store issue rate (240 KB of stores: 8 of 16 us,
// measured with in-kernel stamps), so the update runs on kPcgUpdateBlocks workgroups WITHOUT any
// inter-workgroup synchronisation: every workgroup computes both dot products redundantly from the
// complete old vector set, and stores only its own slice of the cameras into the other set.  Control
// blocks alternate as well (read ctrl2[L&1], written ctrl2[(L+1)&1] by block 0 only), so no workgroup
// can observe a value written during its own launch.

// Start: rhs = -g_c - acc (acc = -sum W e from pass B, MODE 1), x = 0, r = rhs, u = Minv r,
// p = s = 0 in set 0; ctrl2[0] initialised.  Single workgroup.
__global__ __launch_bounds__(1024) void k_pcg_init(const double* __restrict__ Ugc,
                                                   const double* __restrict__ acc,
                                                   const double* __restrict__ Minv, int C,
                                                   double* __restrict__ vecs, const double* __restrict__ tol_dev,
                                                   int max_iters, PcgCtrl* __restrict__ ctrl2) {
    const double tol = *tol_dev;                 // the forcing term k_prep left for this iteration
    __shared__ double red[16];
    const size_t n6 = 6 * (size_t)C;
    double* __restrict__ xk = vecs + kPcgX * n6;
    double* __restrict__ rk = vecs + kPcgR * n6;
    double* __restrict__ pk = vecs + kPcgP * n6;
    double* __restrict__ sk = vecs + kPcgS * n6;
    double* __restrict__ uk = vecs + kPcgU * n6;
    double s[1] = {0.0};
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        double rr[6], m[21];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const size_t e = (size_t)k * C + c;
            rr[k] = -Ugc[(size_t)c * 27 + 21 + k] - acc[e];
            xk[e] = 0.0; pk[e] = 0.0; sk[e] = 0.0;
            rk[e] = rr[k];
        }
#pragma unroll
        for (int n = 0; n < 21; ++n) m[n] = Minv[(size_t)n * C + c];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const double z = minv_row(m, rr, k);
            uk[(size_t)k * C + c] = z;
            vecs[kPcgUcm * n6 + 6 * (size_t)c + k] = z;
            s[0] += z * rr[k];
        }
    }
    block_sum<1>(s, red);
    if (threadIdx.x == 0) {
        PcgCtrl c0;
        c0.rz = s[0]; c0.rz0 = s[0]; c0.tol2 = tol * tol; c0.rz_prev = 1.0; c0.alpha_prev = 1.0;
        c0.iters = 0; c0.max_iters = max_iters;
        c0.done = (s[0] > 0.0) ? 0 : (s[0] == 0.0 ? 1 : 3);
        c0.pad = 0;
        ctrl2[0] = c0;
        ctrl2[1] = c0;
    }
}

// The same start for the LOCAL form whose update is k_pcg_update_local (more than 1024 cameras), on as many workgroups as
// the cameras need and without any reduction: that form takes gamma from the true r.u of the iterate every iteration,
// so gamma_0 = rz0 is simply what its first update finds (ctrl.pad = kPcgInitDeferred says so).  The single-workgroup
// k_pcg_init took 58 us at 5000 cameras -- a tenth of an outer iteration of an eighth-size shard.
__global__ __launch_bounds__(64) void k_pcg_init_local(const double* __restrict__ Ugc, const double* __restrict__ acc,
                                                       const double* __restrict__ Minv, int C, double* __restrict__ vecs,
                                                       const double* __restrict__ tol_dev, int max_iters,
                                                       PcgCtrl* __restrict__ ctrl2) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n6 = 6 * (size_t)C;
    if (c < C) {
        double rr[6], m[21];
#pragma unroll
        for (int n = 0; n < 21; ++n) m[n] = Minv[(size_t)n * C + c];
#pragma unroll
        for (int k = 0; k < 6; ++k) rr[k] = -Ugc[(size_t)c * 27 + 21 + k] - acc[(size_t)k * C + c];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const size_t e = (size_t)k * C + c;
            const double z = minv_row(m, rr, k);
            vecs[kPcgX * n6 + e] = 0.0; vecs[(size_t)kPcgVecs * n6 + kPcgX * n6 + e] = 0.0;
            vecs[kPcgP * n6 + e] = 0.0; vecs[kPcgS * n6 + e] = 0.0;
            vecs[kPcgR * n6 + e] = rr[k];
            vecs[kPcgU * n6 + e] = z;
            vecs[kPcgUcm * n6 + 6 * (size_t)c + k] = z;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        PcgCtrl c0;
        const double tol = *tol_dev;
        c0.rz = 0.0; c0.rz0 = 0.0; c0.tol2 = tol * tol; c0.rz_prev = 1.0; c0.alpha_prev = 1.0;
        c0.iters = 0; c0.max_iters = max_iters; c0.done = 0; c0.pad = kPcgInitDeferred;
        ctrl2[0] = c0;
        ctrl2[1] = c0;
    }
}

// One PCG step after passes A and B of launch L produced acc = (S - Dc) u.  A finished solve turns every
// later sweep/update into a no-op (done is copied forward), so the host may enqueue iterations without
// reading back.
__global__ __launch_bounds__(1024) void k_pcg_update(const double* __restrict__ acc_in,
                                                     const double* __restrict__ Dc,
                                                     const double* __restrict__ Minv, int C,
                                                     double* __restrict__ vecs,
                                                     PcgCtrl* __restrict__ ctrl2, int L) {
    __shared__ double red[16];
    __shared__ double sh_bcast[2];
    const PcgCtrl* __restrict__ cin = ctrl2 + (L & 1);
    PcgCtrl* __restrict__ cout = ctrl2 + ((L + 1) & 1);
    const PcgCtrl ci = *cin;
    if (ci.done != 0) {
        if (blockIdx.x == 0 && threadIdx.x == 0) *cout = ci;
        return;
    }
    const size_t n6 = 6 * (size_t)C;
    const int set = ci.iters & 1;
    const double* __restrict__ vin = vecs + (size_t)set * kPcgVecs * n6;
    double* __restrict__ vout = vecs + (size_t)(set ^ 1) * kPcgVecs * n6;
    const double* __restrict__ u_in = vin + kPcgU * n6;
    // cameras whose results this workgroup stores
    const int slice = ((C + (int)gridDim.x - 1) / (int)gridDim.x + 63) & ~63;
    const int own_lo = (int)blockIdx.x * slice, own_hi = own_lo + slice;

    double d[1] = {0.0};
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const size_t e = (size_t)k * C + c;
            const double ue = u_in[e];
            d[0] += (acc_in[e] + Dc[e] * ue) * ue;
        }
    }
    block_sum<1>(d, red);
    if (threadIdx.x == 0) sh_bcast[0] = d[0];
    __syncthreads();
    const double delta = sh_bcast[0];
    const double gamma = ci.rz;
    const double beta = ci.iters == 0 ? 0.0 : gamma / ci.rz_prev;
    const double den = delta - (ci.iters == 0 ? 0.0 : beta * gamma / ci.alpha_prev);
    const double alpha = gamma / den;
    if (!(den > 0.0) || !isfinite(alpha)) {                      // S not SPD / NaN: uniform in the grid
        if (blockIdx.x == 0 && threadIdx.x == 0) { PcgCtrl co = ci; co.done = 3; *cout = co; }
        return;
    }
    double t[1] = {0.0};
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        double rr[6], uu[6], pp[6], ss[6], xx[6], m[21];
#pragma unroll
        for (int n = 0; n < 21; ++n) m[n] = Minv[(size_t)n * C + c];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const size_t e = (size_t)k * C + c;
            const double ue = u_in[e];
            const double we = acc_in[e] + Dc[e] * ue;
            pp[k] = ue + beta * vin[kPcgP * n6 + e];
            ss[k] = we + beta * vin[kPcgS * n6 + e];
            xx[k] = vin[kPcgX * n6 + e] + alpha * pp[k];
            rr[k] = vin[kPcgR * n6 + e] - alpha * ss[k];
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            uu[k] = minv_row(m, rr, k);
            t[0] += uu[k] * rr[k];
        }
        if (c >= own_lo && c < own_hi) {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const size_t e = (size_t)k * C + c;
                vout[kPcgX * n6 + e] = xx[k]; vout[kPcgR * n6 + e] = rr[k];
                vout[kPcgP * n6 + e] = pp[k]; vout[kPcgS * n6 + e] = ss[k];
                vout[kPcgU * n6 + e] = uu[k];
                vout[kPcgUcm * n6 + 6 * (size_t)c + k] = uu[k];
            }
        }
    }
    block_sum<1>(t, red);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        PcgCtrl co = ci;
        co.rz_prev = ci.rz;
        co.alpha_prev = alpha;
        co.rz = t[0];
        co.iters = ci.iters + 1;
        if (!(t[0] > ci.tol2 * ci.rz0)) co.done = 1;         // also catches NaN
        else if (co.iters >= ci.max_iters) co.done = 2;
        *cout = co;
    }
}

// Back-substitution dp = Vinv (-g_p - sum_i W_i^T dc) per point, fused with everything the 2-D
// trust-region model needs of the new step p = [dc; dp] (SCIPY trf.py:481-485): t2_i = J p with the Gram
// sums G12 = sum t1.t2, G22 = sum t2.t2 (t1 = J D^2 g from k_jdot), and the four dot products
// q5 = g.p, q6 = |p s|^2, q7 = (D^2 g).p, q8 = |p|^2 -- the point part where each dp is produced (run heads),
// the camera part spread over the workgroups.  part[block] = (G12, G22, q5..q8 points, q5..q8 cameras), kBacksubCols wide.
constexpr int kBacksubCols = 10;
// The last step of a PCG solve in the LOCAL fused form, done HERE instead of by one more launch of pass A (fu.part != null;
// LDS_VEC only): when the host replays a record it trusts count for count, the launch of pass A whose prologue would only
// find the solve finished -- alpha of the last iteration, x += alpha p, the control block -- is left out (8 us per outer
// iteration) and every workgroup of this kernel forms alpha from pass B's partial dot products itself
// (pcg_local_step: the same sums in the same order) while it stages the step into its LDS.  Workgroup 0 writes the
// control block as that launch would have, and the finished x into the step vector.  Should the solve NOT be finished (the record was
// too short), the control block says so, k_tr_step cancels the trial as for any short guess, this launch returns at once
// and the host enqueues the owed pass A / pass B pair: nothing the pair reads has been touched.
struct FinalUpdate {
    const double* __restrict__ part;     // [4][C] partial dot products of the cameras (null: x is final already)
    double* __restrict__ vecs;           // PCG vector sets (x and p of the local form live in set 0, x in both)
    PcgCtrl* __restrict__ ctrl2;
    int L;                               // the launch of pass A that was left out
};
template <bool LDS_VEC, bool JFREE = false>
__global__ __launch_bounds__(kSweepThreads) void k_backsub(
    const int2* __restrict__ ranges, int n_ranges, ObsArrays o, const double* __restrict__ dc_planes,
    double* __restrict__ dc, const double* __restrict__ Vinv, const double* __restrict__ gp,
    const double* __restrict__ t1, double* __restrict__ dp, double* __restrict__ part, int C,
    const PcgCtrl* __restrict__ ctrl2, int L, const double* __restrict__ gvec,
    const double* __restrict__ si, const double* __restrict__ sg, Recompute rc, FinalUpdate fu) {
    static_assert(!(LDS_VEC && JFREE), "J-free: the LDS holds the camera table, the step vector comes from L2");
    extern __shared__ __align__(16) double smem[];
    __shared__ double red[kBacksubCols * kWavesPerSweepBlock];
    if (LDS_VEC && fu.part != nullptr) {
        const int n6 = 6 * C, cam = threadIdx.x;
        const PcgCtrl ci = fu.ctrl2[fu.L & 1];
        PcgCtrl co = ci;
        double alpha = 0.0;
        double xe[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, pe[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (cam < C) {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                xe[k] = fu.vecs[kPcgX * n6 + (size_t)k * C + cam];
                pe[k] = fu.vecs[kPcgP * n6 + (size_t)k * C + cam];
            }
        }
        if (ci.done == 0) alpha = pcg_local_step(fu.part, C, ci, co);     // (else: finished earlier than the record says)
        if (blockIdx.x == 0 && threadIdx.x == 0) fu.ctrl2[(fu.L + 1) & 1] = co;
        if (co.done == 0) return;                                         // grid-uniform: the record was too short
        const bool add = ci.done == 0 && co.done != 3;                    // (3: x stays the last good iterate)
        if (cam < C) {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const double xk = add ? xe[k] + alpha * pe[k] : xe[k];
                smem[6 * cam + k] = xk;
                // (the finished x goes to the step vector only: every workgroup of this launch reads x and p of the
                // vector sets, so nobody may write them here; nothing reads x there before the next solve resets it)
                if (blockIdx.x == 0) dc[6 * cam + k] = xk;
            }
        }
        __syncthreads();
    } else {
    if (ctrl2 != nullptr)                      // dc_planes = base of the PCG vector sets: take x of the final set
        dc_planes += (size_t)((ctrl2[L & 1].iters & 1) * kPcgVecs + kPcgX) * 6 * C;
    // dc_planes: PCG solution, plane-major [6][C]; dc: camera-major [C][6] copy (already written by
    // k_transpose when the LDS table is not used)
    if (LDS_VEC) {
        for (int i = threadIdx.x; i < 6 * C; i += blockDim.x) {
            const int k = i / C, c = i - k * C;
            const double val = dc_planes[i];
            smem[6 * c + k] = val;
            if (blockIdx.x == 0) dc[6 * c + k] = val;
        }
        __syncthreads();
    }
    }
    if (JFREE) stage_cam_table(rc.camtab, C, smem);
    const double* __restrict__ vv = LDS_VEC ? smem : dc;
    const int wg = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    int pos = 0, end = 0;
    if (wg < n_ranges) { const int2 rg = ranges[wg]; pos = rg.x; end = rg.y; }
    auto blocks_of = [&](int j, int pj, double* jc_, double* jp_) {       // the blocks of observation j (of point pj)
        if (JFREE) recompute_blocks(rc, smem, o.cam_idx[j], pj, jc_, jp_);
        else load_blocks(o, j, jc_, jp_);
    };
    double g12 = 0.0, g22 = 0.0;
    double qs[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};       // q5..q8: [0..3] points, [4..7] cameras
    auto dots = [&](double* q, size_t e, double pe) {              // element e of the parameter vector, step pe
        const double s = si[e];
        q[0] += gvec[e] * pe; q[1] += (pe * s) * (pe * s); q[2] += sg[e] * pe; q[3] += pe * pe;
    };

    auto jcv = [&](const double* jc, int c, double& t0, double& t1v) {
        // 48-byte rows, 16-byte aligned in both placements: three 128-bit reads
        const double2* a2 = reinterpret_cast<const double2*>(vv + 6 * c);
        const double2 a01 = a2[0], a23 = a2[1], a45 = a2[2];
        const double a[6] = {a01.x, a01.y, a23.x, a23.y, a45.x, a45.y};
        t0 = 0.0; t1v = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) { t0 += jc[k] * a[k]; t1v += jc[6 + k] * a[k]; }
    };
    // point entries: g is g_p itself and D^2 g = g / s^2 exactly as k_update_scale formed it, so only the
    // three scale entries have to be fetched
    auto point_dots = [&](int p, const double* s3, double z0, double z1, double z2) {
        const double z[3] = {z0, z1, z2};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double s_ = s3[k], ge = gp[3 * (size_t)p + k], pe = z[k];
            qs[0] += ge * pe; qs[1] += (pe * s_) * (pe * s_); qs[2] += (ge / (s_ * s_)) * pe; qs[3] += pe * pe;
        }
    };
    auto solve_point = [&](int p, const double* y, double& z0, double& z1, double& z2) {
        const double* vi = Vinv + kVinvRow * (size_t)p;
        const double b0 = -gp[3 * (size_t)p] - y[0], b1 = -gp[3 * (size_t)p + 1] - y[1],
                     b2 = -gp[3 * (size_t)p + 2] - y[2];
        z0 = vi[0] * b0 + vi[1] * b1 + vi[2] * b2;
        z1 = vi[1] * b0 + vi[3] * b1 + vi[4] * b2;
        z2 = vi[2] * b0 + vi[4] * b1 + vi[5] * b2;
        dp[3 * (size_t)p] = z0; dp[3 * (size_t)p + 1] = z1; dp[3 * (size_t)p + 2] = z2;
    };
    auto gram = [&](int i, const double* jp, double t0, double t1v, double z0, double z1, double z2) {
        const double a0 = t0 + jp[0] * z0 + jp[1] * z1 + jp[2] * z2;
        const double a1 = t1v + jp[3] * z0 + jp[4] * z1 + jp[5] * z2;
        const double2 tt = load_pair(t1, o.f32, i);
        g12 += tt.x * a0 + tt.y * a1;
        g22 += a0 * a0 + a1 * a1;
    };

    while (pos < end) {
        const int i = pos + lane;
        const bool in = i < end;
        const int p = in ? o.pt_idx[i] : o.pt_idx[pos];
        const int sb = o.pt_ptr[p], se = o.pt_ptr[p + 1];
        const bool complete = in && (se <= pos + 64);
        const int n_take = __popcll(__ballot(complete));
        double jc[12], jp[6];
        if (n_take == 0) {
            const int run_end = __shfl(se, 0);
            const int pp = __shfl(p, 0);
            double y[3] = {0.0, 0.0, 0.0};
            for (int j = pos + lane; j < run_end; j += 64) {
                blocks_of(j, pp, jc, jp);
                double t0, t1v;
                jcv(jc, o.cam_idx[j], t0, t1v);
                y[0] += jp[0] * t0 + jp[3] * t1v; y[1] += jp[1] * t0 + jp[4] * t1v;
                y[2] += jp[2] * t0 + jp[5] * t1v;
            }
            y[0] = wave_sum(y[0]); y[1] = wave_sum(y[1]); y[2] = wave_sum(y[2]);
            double z0, z1, z2;
            solve_point(pp, y, z0, z1, z2);              // every lane writes the same values
            if (lane == 0) point_dots(pp, si + 6 * (size_t)C + 3 * (size_t)pp, z0, z1, z2);
            for (int j = pos + lane; j < run_end; j += 64) {
                blocks_of(j, pp, jc, jp);
                double t0, t1v;
                jcv(jc, o.cam_idx[j], t0, t1v);
                gram(j, jp, t0, t1v, z0, z1, z2);
            }
            pos = run_end;
            continue;
        }
        const bool act = lane < n_take;
        double t0 = 0.0, t1v = 0.0, z0 = 0.0, z1 = 0.0, z2 = 0.0;
        double y[3] = {0.0, 0.0, 0.0};
        double s3[3] = {1.0, 1.0, 1.0};        // scale entries of the lane's point, requested with the blocks so
        if (act) {                             // that the run head does not wait for them after the reduction
            blocks_of(i, p, jc, jp);
#pragma unroll
            for (int k = 0; k < 3; ++k) s3[k] = si[6 * (size_t)C + 3 * (size_t)p + k];
            jcv(jc, o.cam_idx[i], t0, t1v);
            y[0] = jp[0] * t0 + jp[3] * t1v; y[1] = jp[1] * t0 + jp[4] * t1v;
            y[2] = jp[2] * t0 + jp[5] * t1v;
        }
        seg_reduce_serial<3>(y, act ? sb : -1 - lane, lane);
        if (act && i == sb) { solve_point(p, y, z0, z1, z2); point_dots(p, s3, z0, z1, z2); }
        const int head = act ? lane - (i - sb) : lane;
        z0 = __shfl(z0, head); z1 = __shfl(z1, head); z2 = __shfl(z2, head);
        if (act) gram(i, jp, t0, t1v, z0, z1, z2);
        pos += n_take;
    }
    {   // camera slice (dc is camera-major like x), a few elements per workgroup so that none straggles;
        // after the sweep, so that its four sums are not live across it
        const int per = (6 * C + (int)gridDim.x - 1) / (int)gridDim.x;
        const int e = (int)blockIdx.x * per + (int)threadIdx.x;
        if ((int)threadIdx.x < per && e < 6 * C) dots(qs + 4, (size_t)e, vv[e]);
    }
    double row[kBacksubCols];
    row[0] = wave_sum(g12);
    row[1] = wave_sum(g22);
#pragma unroll
    for (int k = 0; k < 8; ++k) row[2 + k] = wave_sum(qs[k]);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < kBacksubCols; ++k) red[kBacksubCols * (threadIdx.x >> 6) + k] = row[k];
    }
    __syncthreads();
    if (threadIdx.x < kBacksubCols) {                   // one thread per column, waves added in fixed order
        double a = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) a += red[kBacksubCols * w + threadIdx.x];
        part[(size_t)kBacksubCols * blockIdx.x + threadIdx.x] = a;
    }
}

// out [cols][rows] <- in [rows][cols]; camera-major [C][6] <- plane-major [6][C] with rows = 6, cols = C.
// ctrl2 != null: `in` is the base of the PCG vector sets, take x of the final set.
__global__ void k_transpose(const double* __restrict__ in, int rows, int cols, double* __restrict__ out,
                            const PcgCtrl* __restrict__ ctrl2, int L) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    if (ctrl2 != nullptr) in += (size_t)((ctrl2[L & 1].iters & 1) * kPcgVecs + kPcgX) * rows * cols;
    const int r = i / cols, c = i - r * cols;
    out[(size_t)c * rows + r] = in[i];
}

// streaming-store ceiling probe: 16 B per lane, grid-stride
// ---------------------------------------------------------------------------------------------
// Direct all-reduce over peer-mapped memory (xGMI).  The vectors this solver reduces are small (a few
// scalars up to 27 C doubles) and there are a dozen of them per outer iteration, so latency is all that
// matters.  Every rank owns a staging buffer that its peers map through hipIpc:
//     flags[2][W] (one 128-byte line each)   data[2][W][stride]
// One launch per collective: (1) every workgroup copies its slice of the local vector into slot
// [parity][rank] of EVERY rank's buffer (remote stores travel the xGMI links in parallel); the last
// workgroup to finish raises flag [parity][rank] = seq on every rank; (2) every workgroup waits until
// all W flags of its own buffer carry seq and sums its slice over the W slots IN RANK ORDER, so all
// ranks obtain bitwise the same result.  Parity alternates per performed call: a rank can only reach call
// n+2 after every peer has finished reading call n.  The wait gives up after `timeout` ticks of the
// 100 MHz wall clock and raises *error, so the grid always drains.
// All ranks must issue the same sequence of calls (they do: the host loop is replicated).
constexpr int kP2pMaxRanks = 16;
constexpr int kP2pFlagStride = 16;            // uint64 words between two flags (128 bytes)
constexpr int kP2pMaxBlocks = 32;             // waiting workgroups must all be resident
struct P2pArgs {
    double* data[kP2pMaxRanks];               // data region of every rank's staging buffer (own included)
    unsigned long long* flags[kP2pMaxRanks];
    int rank, world;
    long long stride;                         // doubles per slot
    unsigned long long* seq;                  // local: number of collectives performed so far (device-side, so that
                                              // collectives cancelled on the device do not advance the parity)
    const int* cancel;                        // optional: non-zero -> this collective is void on EVERY rank (it
                                              // follows a PCG launch that did nothing); return at once
    unsigned* ticket;                         // local: arrival counter of this launch's workgroups
    unsigned* error;                          // local: set to 1 on timeout
    long long timeout;
    unsigned long long max_mask;              // bit e set: element e (< 64) is reduced with max although op is sum
                                              // (lets one collective carry sums and a maximum)
    // Single-workgroup collectives (a few scalars) can also do the work around them, saving three launches per
    // outer iteration: the final sums that produce the scalars (rider), and the hand-off post behind them.
    Piggyback rider;                          // part == null: none
    Mailbox post;                             // host == null: none
    const double* skip;                       // speculative trial cancelled: no collective, but still post
};

#ifdef SFMBA_P2P_NOFENCE      // experiment only: what the system-scope fences of the collectives cost
#ifndef SFMBA_EXPERIMENT
#error "SFMBA_P2P_NOFENCE builds collectives that are formally unordered: measurement builds only (-DSFMBA_EXPERIMENT)"
#endif
#define P2P_FENCE() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#else
#define P2P_FENCE() __threadfence_system()
#endif
// Arrival ticket of a collective's workgroups.  Acquire + release (agent scope): every workgroup's slot stores are
// fenced (system-scope release) BEFORE its ticket, and the last arriver's read of the ticket ACQUIRES them, so that its
// release store of the peer flag orders all workgroups' slots -- not only its own -- before the flag.  (Round 3 used a
// relaxed RMW here: correct on gfx950 only because system-scope stores are acknowledged at vmcnt(0).)
__device__ __forceinline__ unsigned p2p_take_ticket(unsigned* ticket) {
    return __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
}
// The collective itself, for the workgroups of one launch (any block size; slices by blockIdx).
__device__ __forceinline__ void p2p_allreduce_body(double* __restrict__ vec, int count, int op, const P2pArgs& a) {
    __shared__ unsigned s_last;
    const int tid = threadIdx.x;
    const unsigned long long seq = *a.seq + 1ull;             // the last workgroup to arrive publishes it
    const int par = (int)(seq & 1ull);
    const int per = (count + (int)gridDim.x - 1) / (int)gridDim.x;
    const int lo = (int)blockIdx.x * per, hi = min(count, lo + per);
    for (int q = 0; q < a.world; ++q) {
        double* dst = a.data[q] + ((size_t)par * a.world + a.rank) * a.stride;
        for (int e = lo + tid; e < hi; e += blockDim.x)
            __hip_atomic_store(dst + e, vec[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    P2P_FENCE();
    __syncthreads();
    if (tid == 0) s_last = (p2p_take_ticket(a.ticket) == gridDim.x - 1) ? 1u : 0u;
    __syncthreads();
    if (s_last != 0u) {                          // (every workgroup fenced its stores before its ticket, the last one
        if (tid < a.world)                       // ACQUIRES them with its ticket; the flag's release store publishes all)
            __hip_atomic_store(a.flags[tid] + ((size_t)par * a.world + a.rank) * kP2pFlagStride, seq,
                               __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (tid == 0) { *a.ticket = 0u; *a.seq = seq; }
    }
    if (tid < a.world) {                         // relaxed polls; ONE acquire fence for all threads behind the barrier
        const unsigned long long* f = a.flags[a.rank] + ((size_t)par * a.world + tid) * kP2pFlagStride;
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
            if (wall_clock64() - t0 > a.timeout) { atomicCAS(a.error, 0u, 1u); break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    P2P_FENCE();
    const double* slots = a.data[a.rank] + (size_t)par * a.world * a.stride;
    for (int e = lo + tid; e < hi; e += blockDim.x) {
        double s = __hip_atomic_load(slots + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const bool use_max = op == 1 || (e < 64 && ((a.max_mask >> e) & 1ull) != 0ull);
        for (int q = 1; q < a.world; ++q) {
            const double v = __hip_atomic_load(slots + (size_t)q * a.stride + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            s = use_max ? fmax(s, v) : s + v;
        }
        vec[e] = s;
    }
}

__global__ __launch_bounds__(256) void k_p2p_allreduce(double* __restrict__ vec, int count, int op, P2pArgs a) {
    if (a.skip != nullptr && *a.skip != 0.0) {                // identical on all ranks: sequence number untouched
        if (a.post.host != nullptr && blockIdx.x == 0 && threadIdx.x < 64) post_mailbox(a.post);
        return;
    }
    if (a.cancel != nullptr && *a.cancel != 0) return;        // grid-uniform, identical on all ranks
    if (*a.error != 0u) {                                     // an earlier collective of this solve gave up: do not wait
        if (a.post.host != nullptr && blockIdx.x == 0 && threadIdx.x < 64) post_mailbox(a.post);   // again, let the
        return;                                               // host see the error word at the next hand-off
    }
    const int tid = threadIdx.x;
    if (a.rider.part != nullptr) {                            // (single workgroup) final sums that feed this collective
        finish_in_block(a.rider);
        __syncthreads();
    }
    p2p_allreduce_body(vec, count, op, a);
    if (a.post.host != nullptr) {                             // (single workgroup) the reduced scalars go to the host
        __syncthreads();
        if (tid < 64) post_mailbox(a.post);
    }
}

// ---------------------------------------------------------------------------------------------
// The local form of the PCG when pass B's workgroup does NOT hold the finished product of its camera: sharded solves
// (the product is complete only after the all-reduce over the ranks) and cameras cut into several chunks.  The
// per-camera bookkeeping of the iteration -- what the tail of k_cam_schur does on a single rank -- then runs once per
// camera RIGHT BEHIND the reduction: inside the direct all-reduce kernel (k_p2p_pcg: the workgroup that has just summed
// a camera's six entries over the ranks carries on with that camera), or in k_pcg_tail behind an RCCL / callback
// collective or k_cam_combine.  Pass A's prologue is the light one of the local form in every case (16 doubles per
// camera); nothing is redone per workgroup.  (Round 2 fell back to the general prologue in these cases: every one of
// the 256 workgroups of pass A redid the whole update, 12 us of a 20 us launch on an eighth-size shard, or, past 1024
// cameras, k_pcg_update's eight workgroups each read every vector: 49 us at 5000 cameras.)
// One thread per camera; the arithmetic of the tail of k_cam_schur.
__device__ __forceinline__ void pcg_tail_camera(int c, int C, const double* __restrict__ acc,
                                                const double* __restrict__ u_planes, const PcgLocal& pl, const PcgCtrl& cd) {
    const size_t n6 = 6 * (size_t)C;
    const double beta = cd.iters == 0 ? 0.0 : cd.rz / cd.rz_prev, ap = cd.alpha_prev;
    double u[6], sk[6];
    double d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const size_t e = (size_t)k * C + c;
        u[k] = u_planes[e];
        const double w = acc[e] + pl.Dc[e] * u[k];                  // (S u)_k
        const double p_old = pl.vecs[kPcgP * n6 + e], s_old = pl.vecs[kPcgS * n6 + e];
        const double x = pl.vecs[kPcgX * n6 + e] + ap * p_old;      // the deferred updates of the previous iteration
        const double r = pl.vecs[kPcgR * n6 + e] - ap * s_old;
        sk[k] = w + beta * s_old;
        pl.vecs[kPcgS * n6 + e] = sk[k];
        pl.vecs[kPcgP * n6 + e] = u[k] + beta * p_old;
        pl.vecs[kPcgR * n6 + e] = r;
        pl.vecs[kPcgX * n6 + e] = x;
        pl.vecs[(size_t)kPcgVecs * n6 + kPcgX * n6 + e] = x;        // x lives in both sets (k_backsub)
        d0 += w * u[k]; d1 += sk[k] * u[k]; d3 += r * u[k];
    }
    double m[21];
#pragma unroll
    for (int n = 0; n < 21; ++n) m[n] = pl.Minv[(size_t)n * C + c];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double mk = minv_row(m, sk, k);
        pl.vecs[kPcgM * n6 + (size_t)k * C + c] = mk;
        d2 += sk[k] * mk;
    }
    pl.part[c] = d0; pl.part[(size_t)C + c] = d1; pl.part[2 * (size_t)C + c] = d2; pl.part[3 * (size_t)C + c] = d3;
}

// Behind a collective of another transport, or k_cam_combine: acc holds the finished product.
//   ctrl: the control block pass A of this launch wrote (k_cam_schur's ctrl_done); set < 0: u's set from ctrl->iters
__global__ __launch_bounds__(64) void k_pcg_tail(const double* __restrict__ acc, const double* __restrict__ vin, int C,
                                                 const PcgCtrl* __restrict__ ctrl, int set, PcgLocal pl) {
    const PcgCtrl cd = *ctrl;
    if (cd.done != 0) return;                                 // grid-uniform
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    pcg_tail_camera(c, C, acc, vin + (size_t)((set < 0 ? (cd.iters & 1) : set) * kPcgVecs + kPcgU) * 6 * C, pl, cd);
}

// The direct all-reduce of the product (6 C doubles, plane-major) sliced BY CAMERA, each workgroup carrying on with the
// bookkeeping of the cameras it has just reduced.  Protocol of k_p2p_allreduce (same staging slots, flags, parity,
// ticket, time-out).
constexpr int kP2pPcgThreads = 256;
__global__ __launch_bounds__(kP2pPcgThreads) void k_p2p_pcg(double* __restrict__ acc, const double* __restrict__ vin, int C,
                                                            const PcgCtrl* __restrict__ ctrl, int set, PcgLocal pl, P2pArgs a) {
    __shared__ unsigned s_last;
    const PcgCtrl cd = *ctrl;
    if (cd.done != 0) return;                                 // identical on all ranks: no collective either
    const int tid = threadIdx.x;
    const int per = (C + (int)gridDim.x - 1) / (int)gridDim.x;
    const int lo = (int)blockIdx.x * per, n = max(0, min(C, lo + per) - lo);           // this workgroup's cameras
    if (*a.error == 0u) {
        const unsigned long long seq = *a.seq + 1ull;
        const int par = (int)(seq & 1ull);
        for (int q = 0; q < a.world; ++q) {
            double* dst = a.data[q] + ((size_t)par * a.world + a.rank) * a.stride;
            for (int t = tid; t < 6 * n; t += blockDim.x) {
                const int k = t / n, e = k * C + lo + (t - k * n);
                __hip_atomic_store(dst + e, acc[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        P2P_FENCE();
        __syncthreads();
        if (tid == 0) s_last = (p2p_take_ticket(a.ticket) == gridDim.x - 1) ? 1u : 0u;
        __syncthreads();
        if (s_last != 0u) {
            if (tid < a.world)
                __hip_atomic_store(a.flags[tid] + ((size_t)par * a.world + a.rank) * kP2pFlagStride, seq,
                                   __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            if (tid == 0) { *a.ticket = 0u; *a.seq = seq; }
        }
        if (tid < a.world) {
            const unsigned long long* f = a.flags[a.rank] + ((size_t)par * a.world + tid) * kP2pFlagStride;
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
                if (wall_clock64() - t0 > a.timeout) { atomicCAS(a.error, 0u, 1u); break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        P2P_FENCE();
        const double* slots = a.data[a.rank] + (size_t)par * a.world * a.stride;
        for (int t = tid; t < 6 * n; t += blockDim.x) {
            const int k = t / n, e = k * C + lo + (t - k * n);
            double s = __hip_atomic_load(slots + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            for (int q = 1; q < a.world; ++q)                  // rank order: bitwise the same sum on every rank
                s += __hip_atomic_load(slots + (size_t)q * a.stride + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            acc[e] = s;
        }
    }
    __syncthreads();                                          // this workgroup's entries of acc are final
    if (tid < n)
        pcg_tail_camera(lo + tid, C, acc, vin + (size_t)((set < 0 ? (cd.iters & 1) : set) * kPcgVecs + kPcgU) * 6 * C, pl, cd);
}

// Zero up to eight device arrays in one launch (set_problem's initialisations: seven hipMemsetAsync calls cost the host
// more than the whole set-up of a SceauxCastle-scale problem).  Sizes in 16-byte units; blockIdx.y = array.
struct ZeroJob { void* p[8]; int64_t n16[8]; };
__global__ __launch_bounds__(256) void k_zero_many(ZeroJob z) {
    double2* __restrict__ a = static_cast<double2*>(z.p[blockIdx.y]);
    const int64_t n = z.n16[blockIdx.y];
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
        a[e] = make_double2(0.0, 0.0);
}

__global__ __launch_bounds__(1024) void k_fill16(double* __restrict__ a, int64_t n2, double v) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n2; e += (int64_t)gridDim.x * blockDim.x)
        *reinterpret_cast<double2*>(a + 2 * e) = make_double2(v, v + 1.0);
}

}  // namespace sfmba
