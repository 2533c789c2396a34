// small_solve.hpp -- the WHOLE trust-region solve of a small problem in one launch.
//
// The reference calls bundle adjustment ~55 times per reconstruction on problems of at most 11 cameras / 3k points /
// 10k observations (/root/reference/sfm_lite/sfm.py:59-71, 243-281; BASELINE.json configs[0-1]).  At that size the
// multi-launch path (12-13 launches per outer iteration, sfmba.hip solve_impl) is all latency: ~100 us per outer
// iteration of which the arithmetic is a few microseconds.  Here one resident kernel runs the complete loop of
// SCIPY/optimize/_lsq/trf.py:401-560 (as restated by oracle/ba_oracle.py trf_schur) -- evaluation, normal blocks, column
// scale, Cauchy product, regularisation, Schur blocks, the in-LDS PCG of k_dense_pcg, back-substitution, 2-D
// trust-region step, trial evaluation, accept / reject, termination -- and hands the host ONE result block.
//
// Placement: workgroups are dealt to the eight XCDs round-robin, so the workgroups with blockIdx % 8 == 0 of an
// 8 G-wide launch all sit on ONE XCD and share ONE L2 (the others return at once).  Among them a grid barrier needs
// neither an L2 write-back nor an invalidation: a store is in the L2 once vmcnt == 0 (the vector L1 is write-through),
// and every load of data another workgroup may have written is an `sc1` load (relaxed agent-scope atomic load), which
// bypasses the reader's L1 and is served by that L2 (MI355X_MICROARCH.md, fence table: "sc1 loads bypass L1 only").
// (`buffer_inv sc0` is NOT an acquire for another CU's data -- the first version of this kernel used it after a probe
// whose 32 KB working set happened to evict the L1 on its own, and read stale rows from the second iteration on.)
// The kernel checks the placement (HW_REG_XCC_ID of every workgroup) behind a first, agent-scope barrier and keeps
// agent-scope release / acquire fences in its barriers when it does not hold; every wait is bounded.
//
// Work decomposition: thread gt owns observation gt of the point-major order (its residual and 2x6 / 2x3 blocks stay in
// registers for the whole outer iteration), observation gt of the camera-major order (blocks recomputed), point gt, and
// the entries gt, gt + T, ... of the pair lists (k_schur_blocks' lists).  EVERY sum over observations is the same
// mechanism: values in registers -> segmented reduction inside the wave (seg_reduce_serial) -> the first lane of every
// run writes one row of a segment table (its index is known on the host: segments are cut at run starts and at wave
// starts) -> barrier -> whoever needs the run's total adds its rows in order.  No atomics: same input, same bits.
// Scalars (cost, radius, counters, the 2-D model) are computed redundantly and identically by every thread from the same
// partial rows in the same order; the camera-sized vectors live in every workgroup's LDS.
#pragma once

namespace sfmba {

constexpr int kSmallThreads = kDenseThreads;   // 512: the shape of the in-LDS PCG
constexpr int kSmallWaves = kSmallThreads / 64;
constexpr int kSmallMaxG = 32;                 // one workgroup per CU of one XCD
constexpr int kSmallPartCols = 16;
constexpr int kSmallHist = 112;                // PCG iterations of the first outer iterations (sfmba_get_pcg_history)
constexpr int kSmallOutHead = 16;
constexpr int kSmallOut = kSmallOutHead + kSmallHist;
constexpr int kSmallBarWords = 64;             // [0] arrivals, [1] error, [8 .. 8 + G) XCC ids

struct SmallSeg {
    const int* wseg;          // [passes * waves]: index of the first segment of (pass, wave)
    const int* segptr;        // [runs + 1]: the segments of run q are [segptr[q], segptr[q + 1])
};

struct SmallArgs {
    int C, P, N, E, G, n_blk;
    const int* cam_idx; const int* pt_idx; const double2* uv;            // point-major (the order of set_problem)
    const int* cm_cam; const int* cm_pt; const double2* cm_uv;           // camera-major
    const int* cov_key; const int* cov_pt; const int2* blk_ab;           // pair-major: block pair of every list entry
    SmallSeg sp, sc, ss;                                                  // runs = points, cameras, block pairs
    double* segP; double* segY; double* segC; double* segR; double* segS; // rows: 9, 3 (ids of sp) | 27, 6 (sc) | 36 (ss)
    double* xa; double* xb;                                               // xa holds x0; the result block says which holds x
    double* Vinv; double* e; double* g; double* si; double* sg; double* p; double* r;
    double* part;                                                         // [G][kSmallPartCols]
    unsigned* bar;
    double* out;
    KMat K;
    double ftol, xtol, gtol, reg_min, pcg_tol;                            // pcg_tol: the tolerance the PCG runs to
    int max_nfev, max_iter, pcg_max_iters;
    int force_agent;          // test hook: agent-scope barriers although the workgroups share an XCD
};

// ---- grid barrier of the participating workgroups ---------------------------------------------------------------------
struct SmallSync {
    unsigned* ctr;
    unsigned* err;
    unsigned round, G;
    bool same_xcd;
    int* ok;                  // LDS
};
__device__ __forceinline__ unsigned small_ld(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// every load of a double that another workgroup may have written since this CU last read its line
__device__ __forceinline__ double ldg(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // global_load_dwordx2 ... sc1
}
__device__ __forceinline__ bool small_barrier(SmallSync& s) {
    ++s.round;
    if (s.same_xcd) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's stores are in the L2
    __syncthreads();
    if (threadIdx.x == 0) {
        if (!s.same_xcd) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(s.ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = s.round * s.G;
        bool ok = true;
        unsigned spins = 0;
        while ((int)(small_ld(s.ctr) - target) < 0) {
            if (++spins > (1u << 21)) { __hip_atomic_store(s.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = false; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (!s.same_xcd) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (small_ld(s.err) != 0u) ok = false;                              // somebody else gave up: everybody leaves
        *s.ok = ok ? 1 : 0;
    }
    __syncthreads();
    // one XCD: nothing to invalidate -- shared data is read with sc1 loads (ldg) only
    if (!s.same_xcd) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    else asm volatile("" ::: "memory");
    return *s.ok != 0;
}

// ---- segmented sums ---------------------------------------------------------------------------------------------------
// v: this lane's values, key: its run (runs are contiguous over the lanes; a lane without work passes valid = false and
// a key nobody shares).  The first lane of every run in the wave writes the wave's sum over that run to row
// first_seg + (number of run starts below it).
template <int NV>
__device__ __forceinline__ void seg_emit(double (&v)[NV], int key, bool valid, int first_seg, double* rows, int lane) {
    seg_reduce_serial<NV>(v, key, lane);
    const int prev = lane_below(key, lane);
    const bool head = valid && (lane == 0 || prev != key);
    const unsigned long long m = __ballot(head);
    if (head) {
        const int s = first_seg + __popcll(m & ((1ull << lane) - 1ull));
        double* o = rows + (size_t)s * NV;
#pragma unroll
        for (int n = 0; n < NV; ++n) o[n] = v[n];
    }
}
template <int NV>
__device__ __forceinline__ double seg_total(const double* rows, int s0, int s1, int n) {
    // eight rows requested together (an sc1 load is a round trip to the L2: ~0.7 us each when they wait for one another),
    // added in row order
    double acc = 0.0;
    for (int s = s0; s < s1; s += 8) {
        double v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = s + j < s1 ? ldg(rows + (size_t)(s + j) * NV + n) : 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j];
    }
    return acc;
}

// sums over the workgroup, result on EVERY thread (waves added in wave order); red: (kSmallWaves) * NQ doubles
template <int NQ>
__device__ __forceinline__ void wg_sum(double (&v)[NQ], double* red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < NQ; ++q) v[q] = wave_sum(v[q]);
    __syncthreads();                                    // (red may still be read from the previous reduction)
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) red[w * NQ + q] = v[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < kSmallWaves; ++k) s += red[k * NQ + q];
        v[q] = s;
    }
}
__device__ __forceinline__ double wg_max(double v, double* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double m = 0.0;
#pragma unroll
    for (int k = 0; k < kSmallWaves; ++k) m = fmax(m, red[k]);
    return m;
}

__device__ __forceinline__ double trial_coord(double x, double sg, double p, double c1, double c2) {
    return fma(c2, p, fma(c1, sg, x));                 // x + c1 D^2 g + c2 p (trf.py:495-497), one rounding sequence everywhere
}

__host__ __device__ constexpr size_t small_lds_doubles(int C) {
    // camera rows (current | trial), [U|g_c], camera parameters (current | trial), si g sg p Dc acc, reduction slots,
    // then the region of the in-LDS PCG
    const size_t n6 = 6 * (size_t)C, n = n6;
    size_t head = 2 * (size_t)C * kCamRow + 27 * (size_t)C + 2 * n6 + 6 * n6 + (size_t)kSmallWaves * 12;
    head = (head + 1) & ~(size_t)1;
    return head + n * (n | 1) + 2 * kDenseMaxN + ((21 * kDenseMaxN / 6 + 1) & ~1) + 4 * (kDenseThreads / 64);
}

__global__ __launch_bounds__(kSmallThreads) void k_small_solve(SmallArgs a) {
    if ((blockIdx.x & 7) != 0) return;                  // the other seven XCDs
    extern __shared__ __align__(16) double sm[];
    __shared__ int s_ok;
    const int me = blockIdx.x >> 3, tid = threadIdx.x, lane = tid & 63;
    const int G = a.G, C = a.C, P = a.P, N = a.N, n6 = 6 * C;
    const int gt = me * kSmallThreads + tid, gw = gt >> 6, T = G * kSmallThreads, Wtot = T >> 6;
    double* tab = sm;                                   // [C][kCamRow] rows of the current iterate
    double* tabn = tab + (size_t)C * kCamRow;           // ... of the trial point
    double* Ugc = tabn + (size_t)C * kCamRow;           // [C][27]
    double* xc = Ugc + 27 * (size_t)C;                  // camera parameters, current | trial
    double* xcn = xc + n6;
    double* sic = xcn + n6;                             // scale, gradient, D^2 g, step of the cameras (camera-major)
    double* gc = sic + n6;
    double* sgc = gc + n6;
    double* pc = sgc + n6;
    double* Dc = pc + n6;                               // plane-major [6][C], as dense_pcg_body reads it
    double* accp = Dc + n6;                             // plane-major: reduced right-hand-side term
    double* red = accp + n6;
    double* A = sm + (((size_t)(red - sm) + (size_t)kSmallWaves * 12 + 1) & ~(size_t)1);

    SmallSync sy{a.bar, a.bar + 1, 0u, (unsigned)G, false, &s_ok};
    if (small_ld(a.bar + 1) != 0u) return;              // an earlier workgroup of this launch has already given up
    if (tid == 0) a.bar[8 + me] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;      // hwreg(HW_REG_XCC_ID, 0, 4)
    if (!small_barrier(sy)) return;
    {
        const unsigned x0 = small_ld(a.bar + 8);
        bool same = true;
        for (int w = 1; w < G; ++w) same = same && small_ld(a.bar + 8 + w) == x0;
        sy.same_xcd = same && a.force_agent == 0;
    }

    // ---- what this thread owns ------------------------------------------------------------------------------------
    const bool own = gt < N, ptv = gt < P;
    int ci = 0, pi = 0, cmc = 0, cmp = 0;
    double2 uvv = make_double2(0.0, 0.0), cmuv = make_double2(0.0, 0.0);
    if (own) {
        ci = a.cam_idx[gt]; pi = a.pt_idx[gt]; uvv = a.uv[gt];
        cmc = a.cm_cam[gt]; cmp = a.cm_pt[gt]; cmuv = a.cm_uv[gt];
    }
    const int wsegP = a.sp.wseg[gw], wsegC = a.sc.wseg[gw];
    int ps0 = 0, ps1 = 0;
    if (ptv) { ps0 = a.sp.segptr[gt]; ps1 = a.sp.segptr[gt + 1]; }
    double* xcur = a.xa;
    double* xnew = a.xb;
    for (int t = tid; t < n6; t += kSmallThreads) { xc[t] = a.xa[t]; sic[t] = 0.0; }
    __syncthreads();

    __shared__ long long s_time[16];                    // (diagnostic) 100 MHz ticks per phase, workgroup 0
    if (tid < 16) s_time[tid] = 0;
    long long t_mark = wall_clock64();
    auto lap = [&](int k) {
        if (tid == 0) { const long long t = wall_clock64(); s_time[k] += t - t_mark; t_mark = t; }
    };
    int n_eval = 0;                                     // parity of the cost column (see the hazard note at `evaluate`)
    // Residual + blocks of the owner's observation, point sums (rows of segP), camera sums (rows of segC), the
    // workgroup's part of sum r^2.  trial: the point is x + c1 D^2 g + c2 p, also written to `xnew`.
    // Hazard: a workgroup that has passed the barrier behind this evaluation may start the NEXT evaluation (a rejected
    // step) while a slower one still reads the cost column: the column alternates.
    auto evaluate = [&](const double* xcl, double* tabl, bool trial, double c1, double c2, double& rr0, double& rr1,
                        double* jcn, double* jpn) {
        if (tid < C) cam_row_values(xcl + 6 * tid, tabl + (size_t)kCamRow * tid);
        __syncthreads();
        auto point_of = [&](int p, double& X, double& Y, double& Z) {
            const size_t e = (size_t)n6 + 3 * (size_t)p;
            X = ldg(xcur + e); Y = ldg(xcur + e + 1); Z = ldg(xcur + e + 2);
            if (trial) {
                X = trial_coord(X, ldg(a.sg + e), ldg(a.p + e), c1, c2);
                Y = trial_coord(Y, ldg(a.sg + e + 1), ldg(a.p + e + 1), c1, c2);
                Z = trial_coord(Z, ldg(a.sg + e + 2), ldg(a.p + e + 2), c1, c2);
            }
        };
        double cl[1] = {0.0};
        double v9[9];
#pragma unroll
        for (int n = 0; n < 9; ++n) v9[n] = 0.0;
        if (own) {
            double X, Y, Z;
            point_of(pi, X, Y, Z);
            observe<true>(tabl + (size_t)kCamRow * ci, X, Y, Z, uvv.x, uvv.y, a.K, rr0, rr1, jcn, jpn);
            cl[0] = rr0 * rr0 + rr1 * rr1;
            v9[0] = jpn[0] * jpn[0] + jpn[3] * jpn[3]; v9[1] = jpn[0] * jpn[1] + jpn[3] * jpn[4];
            v9[2] = jpn[0] * jpn[2] + jpn[3] * jpn[5]; v9[3] = jpn[1] * jpn[1] + jpn[4] * jpn[4];
            v9[4] = jpn[1] * jpn[2] + jpn[4] * jpn[5]; v9[5] = jpn[2] * jpn[2] + jpn[5] * jpn[5];
            v9[6] = jpn[0] * rr0 + jpn[3] * rr1; v9[7] = jpn[1] * rr0 + jpn[4] * rr1; v9[8] = jpn[2] * rr0 + jpn[5] * rr1;
        }
        seg_emit<9>(v9, own ? pi : -1 - lane, own, wsegP, a.segP, lane);
        double w27[27];
#pragma unroll
        for (int n = 0; n < 27; ++n) w27[n] = 0.0;
        if (own) {                                      // the same thread's observation of the camera-major order
            double X, Y, Z, q0, q1, jcc[12], jpp[6];
            point_of(cmp, X, Y, Z);
            observe<true>(tabl + (size_t)kCamRow * cmc, X, Y, Z, cmuv.x, cmuv.y, a.K, q0, q1, jcc, jpp);
            int n = 0;
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = i; j < 6; ++j) w27[n++] = jcc[i] * jcc[j] + jcc[6 + i] * jcc[6 + j];
#pragma unroll
            for (int i = 0; i < 6; ++i) w27[21 + i] = jcc[i] * q0 + jcc[6 + i] * q1;
        }
        seg_emit<27>(w27, own ? cmc : -1 - lane, own, wsegC, a.segC, lane);
        wg_sum<1>(cl, red);
        if (tid == 0) a.part[(size_t)me * kSmallPartCols + (n_eval & 1)] = cl[0];
        ++n_eval;
        if (trial) {
            if (ptv) {
                double X, Y, Z;
                point_of(gt, X, Y, Z);
                const size_t e = (size_t)n6 + 3 * (size_t)gt;
                xnew[e] = X; xnew[e + 1] = Y; xnew[e + 2] = Z;
            }
            if (me == 0 && tid < n6) xnew[tid] = xcl[tid];
        }
    };
    // sum of one column of the partial rows: lane w takes workgroup w's row, the wave adds (every wave of every
    // workgroup the same loads and the same order: the same bits everywhere)
    auto total = [&](int col) {
        const double v = lane < G ? ldg(a.part + (size_t)lane * kSmallPartCols + col) : 0.0;
        return wave_sum(v);
    };
    auto finish = [&](int status, int nfev, int njev, int iteration, int pcg_total, double cost0, double cost, double g_norm,
                      double step_norm, double reg, int n_hist, int breakdowns) {
        if (me == 0 && tid == 0) {
            double* o = a.out;
            o[0] = (double)status; o[1] = (double)nfev; o[2] = (double)njev; o[3] = (double)iteration;
            o[4] = (double)pcg_total; o[5] = cost0; o[6] = cost; o[7] = g_norm; o[8] = step_norm; o[9] = reg;
            o[10] = xcur == a.xa ? 0.0 : 1.0; o[11] = (double)n_hist; o[12] = (double)breakdowns;
            o[13] = sy.same_xcd ? 1.0 : 0.0; o[14] = (double)sy.round; o[15] = 1.0;           // [15]: block is complete
            for (int k = 0; k < 16; ++k) o[kSmallOutHead + kSmallHist / 2 + k] = 0.01 * (double)s_time[k];   // us
        }
    };

    // ---- f0, J0 (least_squares.py:838, 903-912) ---------------------------------------------------------------------
    double r0 = 0.0, r1 = 0.0, jc[12], jp[6];           // the owner's observation at the CURRENT iterate
    double rn0 = 0.0, rn1 = 0.0, jcn[12], jpn[6];       // ... at the trial point
#pragma unroll
    for (int k = 0; k < 12; ++k) { jc[k] = 0.0; jcn[k] = 0.0; }
#pragma unroll
    for (int k = 0; k < 6; ++k) { jp[k] = 0.0; jpn[k] = 0.0; }
    evaluate(xc, tab, false, 0.0, 0.0, r0, r1, jc, jp);
    if (!small_barrier(sy)) return;
    lap(0);
    double cost = 0.5 * total(0);
    const double cost0 = cost;
    if (!isfinite(cost)) { finish(-2, 1, 1, 0, 0, cost0, cost, 0.0, 0.0, 0.0, 0, 0); return; }

    int nfev = 1, njev = 1, iteration = 0, status = -1, pcg_total = 0, n_hist = 0, breakdowns = 0;
    double Delta = 0.0, step_norm = 0.0, actual_reduction = 0.0, g_norm = 0.0, reg = 0.0;
    bool need_lin = true, first = true;
    double gpr[3] = {0.0, 0.0, 0.0}, sr[3] = {0.0, 0.0, 0.0}, Vr[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};   // this thread's point
    double qp[5] = {0.0, 0.0, 0.0, 0.0, 0.0}, qcm[5] = {0.0, 0.0, 0.0, 0.0, 0.0};   // q0..q4: point slice | camera slice

    for (;;) {                                          // trf.py:450
        if (need_lin) {
            // ---- normal blocks of the accepted point from the rows its evaluation left; column scale of
            // x_scale='jac' (common.py:598-610), gradient, D^2 g, q0..q4 ------------------------------------------------
            double q[4] = {0.0, 0.0, 0.0, 0.0}, qmax = 0.0;
            if (ptv) {
                double s9[9];
#pragma unroll
                for (int n = 0; n < 9; ++n) s9[n] = 0.0;
#pragma unroll
                for (int n = 0; n < 9; ++n) s9[n] = seg_total<9>(a.segP, ps0, ps1, n);
#pragma unroll
                for (int k = 0; k < 6; ++k) Vr[k] = s9[k];
                const int dv[3] = {0, 3, 5};
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const size_t e = (size_t)n6 + 3 * (size_t)gt + k;
                    const double ge = s9[6 + k];
                    double s = sqrt(s9[dv[k]]);
                    if (first) { if (s == 0.0) s = 1.0; } else { s = fmax(s, sr[k]); }
                    const double sge = ge / (s * s), xe = ldg(xcur + e);
                    a.si[e] = s; a.g[e] = ge; a.sg[e] = sge;
                    gpr[k] = ge; sr[k] = s;
                    qmax = fmax(qmax, fabs(ge));
                    q[0] += (ge / s) * (ge / s); q[1] += (xe * s) * (xe * s); q[2] += xe * xe; q[3] += sge * sge;
                }
            }
            for (int t = tid; t < 27 * C; t += kSmallThreads) {
                const int c = t / 27, n = t - 27 * c;
                Ugc[t] = seg_total<27>(a.segC, a.sc.segptr[c], a.sc.segptr[c + 1], n);
            }
            __syncthreads();
            double qc[4] = {0.0, 0.0, 0.0, 0.0}, qcmax = 0.0;
            if (tid < n6) {
                const int diagU[6] = {0, 6, 11, 15, 18, 20};
                const int c = tid / 6, k = tid - 6 * c;
                const double ge = Ugc[c * 27 + 21 + k];
                double s = sqrt(Ugc[c * 27 + diagU[k]]);
                if (first) { if (s == 0.0) s = 1.0; } else { s = fmax(s, sic[tid]); }
                const double sge = ge / (s * s), xe = xc[tid];
                sic[tid] = s; gc[tid] = ge; sgc[tid] = sge;
                if (me == 0) { a.si[tid] = s; a.g[tid] = ge; a.sg[tid] = sge; }
                qcmax = fabs(ge);
                qc[0] = (ge / s) * (ge / s); qc[1] = (xe * s) * (xe * s); qc[2] = xe * xe; qc[3] = sge * sge;
            }
            wg_sum<4>(qc, red);
            qcmax = wg_max(qcmax, red);
            qcm[0] = qcmax;
#pragma unroll
            for (int k = 0; k < 4; ++k) qcm[1 + k] = qc[k];
            wg_sum<4>(q, red);
            qmax = wg_max(qmax, red);
            if (tid == 0) {
                double* row = a.part + (size_t)me * kSmallPartCols;
                row[2] = qmax; row[3] = q[0]; row[4] = q[1]; row[5] = q[2]; row[6] = q[3];
            }
            lap(1);
            if (!small_barrier(sy)) return;
            lap(2);
            qp[0] = wave_max(lane < G ? ldg(a.part + (size_t)lane * kSmallPartCols + 2) : 0.0);
#pragma unroll
            for (int k = 0; k < 4; ++k) qp[1 + k] = total(3 + k);
            if (first) {
                Delta = sqrt(qp[2] + qcm[2]);           // |x0 * scale_inv|, trf.py:428
                if (Delta == 0.0) Delta = 1.0;
            }
            first = false;
            need_lin = false;
        }
        // ---- loop head, trf.py:450-459 --------------------------------------------------------------------------------
        g_norm = fmax(qp[0], qcm[0]);
        if (g_norm < a.gtol) status = 1;
        if (status != -1 || nfev >= a.max_nfev || (a.max_iter > 0 && iteration >= a.max_iter)) break;
        const double a11 = qp[1] + qcm[1];              // |g_h|^2

        // ---- t1 = J D^2 g per observation (registers), G11 = |t1|^2 ------------------------------------------------------
        double t1x = 0.0, t1y = 0.0;
        {
            double gl[1] = {0.0};
            if (own) {
                const double* sc6 = sgc + 6 * ci;
                const size_t e = (size_t)n6 + 3 * (size_t)pi;
                const double b0 = ldg(a.sg + e), b1 = ldg(a.sg + e + 1), b2 = ldg(a.sg + e + 2);
#pragma unroll
                for (int k = 0; k < 6; ++k) { t1x += jc[k] * sc6[k]; t1y += jc[6 + k] * sc6[k]; }
                t1x += jp[0] * b0 + jp[1] * b1 + jp[2] * b2;
                t1y += jp[3] * b0 + jp[4] * b1 + jp[5] * b2;
                gl[0] = t1x * t1x + t1y * t1y;
            }
            wg_sum<1>(gl, red);
            if (tid == 0) a.part[(size_t)me * kSmallPartCols + 7] = gl[0];
        }
        if (!small_barrier(sy)) return;
        lap(3);
        const double G11 = total(7);
        // ---- regularisation (trf.py:471-475), Vinv and e per point, Dc per camera ----------------------------------------
        reg = reg_from_a11(a11, G11, Delta, a.reg_min);
        double vinv[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, er[3] = {0.0, 0.0, 0.0};
        if (ptv) {
            double m6[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) m6[k] = Vr[k];
            m6[0] += reg * sr[0] * sr[0]; m6[3] += reg * sr[1] * sr[1]; m6[5] += reg * sr[2] * sr[2];
            chol3_inverse(m6, vinv);
            er[0] = vinv[0] * gpr[0] + vinv[1] * gpr[1] + vinv[2] * gpr[2];
            er[1] = vinv[1] * gpr[0] + vinv[3] * gpr[1] + vinv[4] * gpr[2];
            er[2] = vinv[2] * gpr[0] + vinv[4] * gpr[1] + vinv[5] * gpr[2];
#pragma unroll
            for (int k = 0; k < 6; ++k) a.Vinv[(size_t)gt * 6 + k] = vinv[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) a.e[(size_t)gt * 3 + k] = er[k];
        }
        if (tid < n6) { const int c = tid / 6, k = tid - 6 * c; Dc[k * C + c] = reg * sic[tid] * sic[tid]; }
        if (!small_barrier(sy)) return;
        lap(4);
        // ---- blocks of W Vinv W^T per camera pair (k_schur_blocks' lists), reduced right-hand side per camera ---------------
        for (int k0 = 0; k0 < a.E; k0 += T) {
            const int k = k0 + gt;
            const bool valid = k < a.E;
            double s36[36];
#pragma unroll
            for (int n = 0; n < 36; ++n) s36[n] = 0.0;
            int key = -1 - lane;
            if (valid) {
                key = a.cov_key[k];
                const int pm = a.cov_pt[k], p = pm < 0 ? ~pm : pm;
                const int2 ab = a.blk_ab[key];
                const size_t e = (size_t)n6 + 3 * (size_t)p;
                const double X = ldg(xcur + e), Y = ldg(xcur + e + 1), Z = ldg(xcur + e + 2);
                const double* vi = a.Vinv + (size_t)p * 6;
                const double v0 = ldg(vi), v1 = ldg(vi + 1), v2 = ldg(vi + 2), v3 = ldg(vi + 3), v4 = ldg(vi + 4), v5 = ldg(vi + 5);
                double ja[12], pa[6], jb[12], pb[6], rx, ry;
                observe<true>(tab + (size_t)kCamRow * ab.x, X, Y, Z, 0.0, 0.0, a.K, rx, ry, ja, pa);
                if (ab.x != ab.y) observe<true>(tab + (size_t)kCamRow * ab.y, X, Y, Z, 0.0, 0.0, a.K, rx, ry, jb, pb);
                else {
#pragma unroll
                    for (int q = 0; q < 12; ++q) jb[q] = ja[q];
#pragma unroll
                    for (int q = 0; q < 6; ++q) pb[q] = pa[q];
                }
                // W_a Vinv W_b^T = Jc_a^T (Jp_a Vinv Jp_b^T) Jc_b: the 2 x 2 core first (W = Jc^T Jp has rank two)
                const double h00 = v0 * pb[0] + v1 * pb[1] + v2 * pb[2], h01 = v1 * pb[0] + v3 * pb[1] + v4 * pb[2],
                             h02 = v2 * pb[0] + v4 * pb[1] + v5 * pb[2];          // Vinv (row 0 of Jp_b)
                const double h10 = v0 * pb[3] + v1 * pb[4] + v2 * pb[5], h11 = v1 * pb[3] + v3 * pb[4] + v4 * pb[5],
                             h12 = v2 * pb[3] + v4 * pb[4] + v5 * pb[5];
                const double g00 = pa[0] * h00 + pa[1] * h01 + pa[2] * h02, g01 = pa[0] * h10 + pa[1] * h11 + pa[2] * h12;
                const double g10 = pa[3] * h00 + pa[4] * h01 + pa[5] * h02, g11 = pa[3] * h10 + pa[4] * h11 + pa[5] * h12;
#pragma unroll
                for (int v = 0; v < 6; ++v) {
                    const double m0 = g00 * jb[v] + g01 * jb[6 + v], m1 = g10 * jb[v] + g11 * jb[6 + v];
#pragma unroll
                    for (int u = 0; u < 6; ++u) s36[6 * u + v] = ja[u] * m0 + ja[6 + u] * m1;
                }
            }
            seg_emit<36>(s36, key, valid, a.ss.wseg[(k0 / T) * Wtot + gw], a.segS, lane);
        }
        {
            double v6[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            if (own) {
                const size_t e = (size_t)n6 + 3 * (size_t)cmp;
                double jcc[12], jpp[6], rx, ry;
                observe<true>(tab + (size_t)kCamRow * cmc, ldg(xcur + e), ldg(xcur + e + 1), ldg(xcur + e + 2), 0.0, 0.0, a.K, rx, ry, jcc, jpp);
                const double* ep = a.e + (size_t)cmp * 3;
                const double e0 = ldg(ep), e1 = ldg(ep + 1), e2 = ldg(ep + 2);
                const double s0 = jpp[0] * e0 + jpp[1] * e1 + jpp[2] * e2, s1 = jpp[3] * e0 + jpp[4] * e1 + jpp[5] * e2;
#pragma unroll
                for (int i = 0; i < 6; ++i) v6[i] = -(jcc[i] * s0 + jcc[6 + i] * s1);        // -W e = -Jc^T (Jp e)
            }
            seg_emit<6>(v6, own ? cmc : -1 - lane, own, wsegC, a.segR, lane);
        }
        lap(5);
        if (!small_barrier(sy)) return;
        lap(6);
        // ---- S dc = -g_c - acc by the PCG of k_dense_pcg, in every workgroup (the same bits everywhere) -------------------
        if (tid < n6) {
            const int c = tid / 6, k = tid - 6 * c;
            accp[k * C + c] = seg_total<6>(a.segR, a.sc.segptr[c], a.sc.segptr[c + 1], k);
        }
        __syncthreads();
        const PcgCtrl hc = dense_pcg_body(
            A, [&](int e) { const int blk = e / 36; return seg_total<36>(a.segS, a.ss.segptr[blk], a.ss.segptr[blk + 1], e - 36 * blk); },
            Ugc, Dc, accp, C, a.pcg_tol, a.pcg_max_iters, [&](int cam, int kk, double xo) { pc[6 * cam + kk] = xo; });
        __syncthreads();
        lap(7);
        pcg_total += hc.iters;
        if (hc.done == 3) ++breakdowns;
        if (me == 0 && tid == 0 && n_hist < kSmallHist / 2) a.out[kSmallOutHead + n_hist] = (double)hc.iters;
        ++n_hist;
        // ---- back-substitution dp = Vinv (-g_p - sum W^T dc), model products ---------------------------------------------
        double u0 = 0.0, u1 = 0.0;                      // Jc dc of the owner's observation
        {
            double y3[3] = {0.0, 0.0, 0.0};
            if (own) {
                const double* d6 = pc + 6 * ci;
#pragma unroll
                for (int k = 0; k < 6; ++k) { u0 += jc[k] * d6[k]; u1 += jc[6 + k] * d6[k]; }
                y3[0] = jp[0] * u0 + jp[3] * u1; y3[1] = jp[1] * u0 + jp[4] * u1; y3[2] = jp[2] * u0 + jp[5] * u1;
            }
            seg_emit<3>(y3, own ? pi : -1 - lane, own, wsegP, a.segY, lane);
        }
        if (!small_barrier(sy)) return;
        lap(8);
        double qb[4] = {0.0, 0.0, 0.0, 0.0}, qbc[4] = {0.0, 0.0, 0.0, 0.0};          // q5..q8: points | cameras
        if (ptv) {
            const double b0 = -gpr[0] - seg_total<3>(a.segY, ps0, ps1, 0), b1 = -gpr[1] - seg_total<3>(a.segY, ps0, ps1, 1),
                         b2 = -gpr[2] - seg_total<3>(a.segY, ps0, ps1, 2);
            const double z[3] = {vinv[0] * b0 + vinv[1] * b1 + vinv[2] * b2, vinv[1] * b0 + vinv[3] * b1 + vinv[4] * b2,
                                 vinv[2] * b0 + vinv[4] * b1 + vinv[5] * b2};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                a.p[(size_t)n6 + 3 * (size_t)gt + k] = z[k];
                const double s_ = sr[k], ge = gpr[k], pe = z[k];
                qb[0] += ge * pe; qb[1] += (pe * s_) * (pe * s_); qb[2] += (ge / (s_ * s_)) * pe; qb[3] += pe * pe;
            }
        }
        if (tid < n6) {
            const double pe = pc[tid], s_ = sic[tid];
            qbc[0] = gc[tid] * pe; qbc[1] = (pe * s_) * (pe * s_); qbc[2] = sgc[tid] * pe; qbc[3] = pe * pe;
            if (me == 0) a.p[tid] = pe;
        }
        wg_sum<4>(qbc, red);
        wg_sum<4>(qb, red);
        if (tid == 0) {
            double* row = a.part + (size_t)me * kSmallPartCols;
            row[8] = qb[0]; row[9] = qb[1]; row[10] = qb[2]; row[11] = qb[3];
        }
        if (!small_barrier(sy)) return;
        lap(9);
        {
            double gg[2] = {0.0, 0.0};
            if (own) {
                const size_t e = (size_t)n6 + 3 * (size_t)pi;
                const double z0 = ldg(a.p + e), z1 = ldg(a.p + e + 1), z2 = ldg(a.p + e + 2);
                const double t2x = u0 + jp[0] * z0 + jp[1] * z1 + jp[2] * z2, t2y = u1 + jp[3] * z0 + jp[4] * z1 + jp[5] * z2;
                gg[0] = t1x * t2x + t1y * t2y;
                gg[1] = t2x * t2x + t2y * t2y;
            }
            wg_sum<2>(gg, red);
            if (tid == 0) { a.part[(size_t)me * kSmallPartCols + 12] = gg[0]; a.part[(size_t)me * kSmallPartCols + 13] = gg[1]; }
        }
        if (!small_barrier(sy)) return;
        lap(10);
        // ---- the 2-D subspace model (trf.py:481-485) and the step loop (trf.py:488-526) -------------------------------------
        const double G12 = total(12), G22 = total(13);
        const TrModel model = tr_build_model(G11, G12, G22, a11, total(8) + qbc[0], total(9) + qbc[1], qp[4] + qcm[4],
                                             total(10) + qbc[2], total(11) + qbc[3]);
        const double x_norm = sqrt(qp[3] + qcm[3]);
        actual_reduction = -1.0;
        double cost_new = cost;
        while (actual_reduction <= 0.0 && nfev < a.max_nfev) {
            const TrStep st = tr_solve_step(model, Delta);
            if (tid < n6) xcn[tid] = trial_coord(xc[tid], sgc[tid], pc[tid], st.c1, st.c2);
            __syncthreads();
            lap(11);
            const int col = n_eval & 1;
            evaluate(xcn, tabn, true, st.c1, st.c2, rn0, rn1, jcn, jpn);
            lap(12);
            if (!small_barrier(sy)) return;
            lap(13);
            ++nfev;
            cost_new = 0.5 * total(col);
            if (!isfinite(cost_new)) {                  // trf.py:504-506
                Delta = 0.25 * st.step_h_norm;
                continue;
            }
            actual_reduction = cost - cost_new;
            double ratio;
            const double Delta_new = update_tr_radius(Delta, actual_reduction, st.predicted, st.step_h_norm,
                                                      st.step_h_norm > 0.95 * Delta, &ratio);
            step_norm = st.step_norm;
            const int term = check_termination(actual_reduction, cost, step_norm, x_norm, ratio, a.ftol, a.xtol);
            if (term != 0) { status = term; break; }
            Delta = Delta_new;
        }
        if (actual_reduction > 0.0) {                   // trf.py:528
            double* t_ = xcur; xcur = xnew; xnew = t_;
            t_ = xc; xc = xcn; xcn = t_;
            t_ = tab; tab = tabn; tabn = t_;
            r0 = rn0; r1 = rn1;
#pragma unroll
            for (int k = 0; k < 12; ++k) jc[k] = jcn[k];
#pragma unroll
            for (int k = 0; k < 6; ++k) jp[k] = jpn[k];
            cost = cost_new;
            ++njev;
            need_lin = true;
        } else {
            step_norm = 0.0;
            actual_reduction = 0.0;
        }
        ++iteration;
    }
    if (status == -1) status = 0;
    if (own) reinterpret_cast<double2*>(a.r)[gt] = make_double2(r0, r1);      // result.fun: f at the final x
    if (xcur != a.xa) {                                 // the host reads x from xa: every thread copies what it wrote itself
        if (ptv) {
            const size_t e = (size_t)n6 + 3 * (size_t)gt;
            a.xa[e] = ldg(xcur + e); a.xa[e + 1] = ldg(xcur + e + 1); a.xa[e + 2] = ldg(xcur + e + 2);
        }
        if (me == 0 && tid < n6) a.xa[tid] = xc[tid];
    }
    finish(status, nfev, njev, iteration, pcg_total, cost0, cost, g_norm, step_norm, reg, n_hist, breakdowns);
}

}  // namespace sfmba
