// sfmba.hip -- C-ABI (include/sfmba.h) and the host controller of the MI355X bundle-adjustment path.
//
// The controller restates scipy's trf_no_bounds (SCIPY/optimize/_lsq/trf.py:401-560, the code behind
// the least_squares(method='trf', x_scale='jac') call of /root/reference/sfm_lite/sfm.py:266-268):
// same scaling, same 1-D Cauchy regularisation, same 2-D subspace trust-region step, same radius
// update and termination tests.  Two things differ by design: the Jacobian is analytic (K1) instead
// of forward-differenced, and the damped Gauss-Newton step of trf.py:480 is obtained from the Schur
// complement on the cameras with block-Jacobi PCG (K4-K6) instead of LSMR on the full system.
// Only scalars and 2x2 systems are handled on the host; every n- or m-vector stays in HBM.
#include "../../include/sfmba.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <rccl/rccl.h>   // types only: librccl.so.1 is dlopen()ed by sfmba_comm_init
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "ba_kernels.hpp"
#include "tr2d.hpp"

using namespace sfmba;

namespace {

constexpr size_t kLdsBytes = 160 * 1024;       // gfx950: 160 KiB LDS per workgroup
constexpr size_t kLdsDynMax = kLdsBytes - 2048; // dynamic part; every kernel here has <= 2 KiB static LDS (k_backsub: 1280 B)
// Scalars at the end of the exchange arena (32 doubles):
//   summed over ranks [0..11]: 0 sum r^2 | 1 G11 | 2 G12 | 3 G22 | 4..7 q5..q8 | 8..11 q1..q4 of the point
//                              slice (grouped so that each phase all-reduces one contiguous run of fresh values)
//   max over ranks    [12]   : q0 = max|g| of the point slice
//   device-local      [13]   : regularisation term of this iteration (k_prep)
//                     [14,15]: |g_h|^2 of the previous iterate, forcing term of this iteration's PCG (k_prep)
//   never exchanged   [16..24]: q0..q8 of the camera slice (replicated on every rank)
constexpr int kScalSlots = 32;
// Per-workgroup partial sums live in two halves of one buffer: A (k_update_scale, the cost of
// k_resjac) and B (k_jdot, k_backsub).  Consecutive producers alternate halves, so the final sums of one
// producer can ride along with the NEXT producer's launch (Piggyback) without a race on the rows.
constexpr int kPartRows = 2048;
constexpr int kMaxSlot = 12, kRegSlot = 13, kGhPrevSlot = 14, kEtaSlot = 15, kCamSlot = 16;
constexpr int kPointSlot[9] = {12, 8, 9, 10, 11, 4, 5, 6, 7};     // slot of q0..q8 of the point slice

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t ensure(size_t n) {
        if (n <= bytes && p) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        if (n == 0) n = 8;
        hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) bytes = n;
        return e;
    }
    // grow-only like ensure(), but the first `keep` bytes survive a reallocation (device-to-device copy) and the new
    // block has headroom, so that a problem that grows call by call does not reallocate every time
    hipError_t ensure_keep(size_t n, size_t keep) {
        if (n <= bytes && p) return hipSuccess;
        if (keep == 0 || !p) return ensure(n + n / 2);
        void* q = nullptr;
        const size_t cap = n + n / 2;
        hipError_t e = hipMalloc(&q, cap);
        if (e != hipSuccess) return e;
        e = hipMemcpy(q, p, std::min(keep, bytes), hipMemcpyDeviceToDevice);
        (void)hipFree(p);
        p = q; bytes = cap;
        return e;
    }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

// a piece of a larger device allocation (the structure tables of a problem share one buffer and one upload)
struct DevView {
    void* p = nullptr;
    template <class T> T* as() const { return static_cast<T*>(p); }
};

// pinned host memory, grow-only, optionally keeping a prefix across a reallocation
struct PinnedBuf {
    void* p = nullptr;
    size_t bytes = 0;
    ~PinnedBuf() { if (p) (void)hipHostFree(p); }
    hipError_t ensure(size_t n, size_t keep) {
        if (n <= bytes && p) return hipSuccess;
        void* q = nullptr;
        const size_t cap = n + n / 2 + 64;
        hipError_t e = hipHostMalloc(&q, cap, hipHostMallocDefault);
        if (e != hipSuccess) return e;
        if (p && keep) memcpy(q, p, std::min(keep, bytes));
        if (p) (void)hipHostFree(p);
        p = q; bytes = cap;
        return hipSuccess;
    }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

// A few persistent host threads for set_problem's passes over the observation arrays (spawning threads per
// call cost more than the passes themselves at a million observations).  run(parts, fn) calls fn(0..parts-1),
// part 0 on the calling thread, and returns when all are done.
class HostPool {
public:
    static constexpr int kMax = 16;
    ~HostPool() {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; ++gen_; }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    int parts_for(int64_t n) const {
        const int hw = (int)std::thread::hardware_concurrency();
        return (int)std::max<int64_t>(1, std::min<int64_t>(std::min(kMax, hw > 0 ? hw : 1), n / 65536));
    }
    template <class Fn>
    void run(int parts, Fn fn) {
        if (parts <= 1) { fn(0); return; }
        while ((int)th_.size() < parts - 1) {
            const int id = (int)th_.size() + 1;
            th_.emplace_back([this, id] { worker(id); });
        }
        std::function<void(int)> f = fn;
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = &f; job_parts_ = parts; pending_ = parts - 1; ++gen_;
        }
        cv_.notify_all();
        fn(0);
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [this] { return pending_ == 0; });
        job_ = nullptr;
    }
private:
    void worker(int id) {
        unsigned long long seen = 0;
        for (;;) {
            std::function<void(int)>* f = nullptr;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                if (id >= job_parts_ || job_ == nullptr) continue;
                f = job_;
            }
            (*f)(id);
            {
                std::lock_guard<std::mutex> lk(m_);
                if (--pending_ == 0) done_.notify_one();
            }
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    std::function<void(int)>* job_ = nullptr;
    int job_parts_ = 0, pending_ = 0;
    unsigned long long gen_ = 0;
    bool stop_ = false;
};

// RCCL entry points, resolved at run time so that libsfmba.so has no load-time dependency on RCCL
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi* rccl_api() {
    static RcclApi api;
    static bool tried = false;
    if (!tried) {
        tried = true;
        void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (lib) {
            api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
            api.CommInitRank = (decltype(api.CommInitRank))dlsym(lib, "ncclCommInitRank");
            api.AllReduce = (decltype(api.AllReduce))dlsym(lib, "ncclAllReduce");
            api.CommDestroy = (decltype(api.CommDestroy))dlsym(lib, "ncclCommDestroy");
            api.GetErrorString = (decltype(api.GetErrorString))dlsym(lib, "ncclGetErrorString");
            if (api.GetUniqueId && api.CommInitRank && api.AllReduce && api.CommDestroy && api.GetErrorString)
                api.lib = lib;
        }
    }
    return api.lib ? &api : nullptr;
}

double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

struct sfmba_handle {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    int n_cu = 256;

    // problem
    bool have_problem = false;
    int64_t C = 0, P = 0, N = 0, ld = 0, n = 0;
    int64_t N_total = 0;
    KMat K{};
    bool permuted = false;
    std::vector<int64_t> order;              // sorted position -> caller's observation index
    int n_ranges = 0;
    bool f32 = false;                        // fp32 storage of uv, r, t1 and the Jacobian (arithmetic stays fp64)
    bool f32_next = false;                   // takes effect at the next sfmba_set_problem
    std::vector<int64_t> fixed_next;         // sfmba_set_fixed_cameras: cameras held still from the next sfmba_set_problem on
    int64_t n_fixed = 0;                     // ... of the current problem
    bool lds_tab = true, lds_vec = true;     // camera table (K1, K2) / camera vector (sweeps) staged in LDS
    bool sweep_rc = false;                   // pass A recomputes the blocks from an LDS table (k_point_sweep_rc)
    bool sweep_rc_g = false;                 // ... from a table in global memory (more cameras than the LDS holds)
    DevBuf rctab;                            // [C][18], k_rc_table
    // mixed-precision Schur product (ba_kernels.hpp: MixedPrep): fp32 operands, fp64 arithmetic
    bool jfree = false;                      // J-free iteration (debug option jfree; needs the camera table in LDS)
    bool mixed = false;                      // pass A
    bool mixed_b = false;                    // ... and pass B (fp32 point records)
    DevBuf rt32, rec32, rctab32, rtd;        // [C][12], [P][8], [C][20] floats; [C][12] doubles
    Origin origin{0.0, 0.0, 0.0};            // coordinates are rounded relative to it (set with every uploaded x)
    bool dense = false;                      // reduced camera matrix formed and factorised (6 C <= kDenseMaxN) instead of PCG
    DevView cov_ptr, cov_pt, blk_ab;         // dense path: per block pair (a <= b) the points both cameras see
    DevBuf Sblk;
    DevBuf tables;                           // ranges | wsteps | steps | chunk table | chunk offsets | pair lists: ONE upload
    int n_blk = 0;
    // test / diagnostic hooks, set through sfmba_debug_option only (nothing reads the environment)
    struct Debug {
        int pcg_fused = -1;                  // 0: two-kernel PCG although the fused launch would fit
        int tab_lds = -1, vec_lds = -1;      // 0: camera table / camera vector read from L2 although LDS would fit
        int sweep_rc = -1;                   // 0: pass A reads the stored Jacobian although the recomputing form would fit
        int dense = -1;                      // 0: PCG although the dense reduced-camera path would apply
        int precond = -1;                    // 0: block-Jacobi preconditioner from U + Dc instead of the Schur diagonal
        int pcg_local = -1;                  // 0: the fused PCG keeps its whole update in pass A's prologue
        int xcd_chunks = -1;                 // 1 / 0: camera lists cut at the eight point-range boundaries (one chunk per XCD) whatever the size
        int rhsrec = -1;                     // 1 / 0: the rhs + preconditioner pass gathers its own 128-byte records whatever the size
        int cost_rider = -1;                 // 0: the trial cost is summed and posted by a k_finish launch of its own
        int packed_upload = -1;              // 0: the observation arrays are uploaded as int32 / fp64 although they would pack
        int jfree = -1;                      // 1: the J-free iteration (measurement): K1 does not write the Jacobian, k_jdot and
                                             // k_backsub recompute its blocks from the LDS camera table
        int cm_device = -1;                  // 0: the camera-major order is sorted on the host and its permutation uploaded
        int pcg_inline = -1;                 // 0: sharded solves keep the collective of the product as a launch of its own
        int xcd_cam = -1;                    // 0: K3 and the rhs pass keep the one-chunk-per-camera table where pass B takes the XCD-aware one
        int pcg_skip_last = -1;              // 0: the pass B behind the launch the record says is the last one is enqueued all the same
        int pcg_mixed_b = -1;                // 0: pass B keeps fp64 point records although pass A runs on fp32 operands
        int pcg_mixed = -1;                  // 1 / 0: fp32 operands in the implicit Schur product whatever the storage mode
        int pcg_split = -1;                  // 1: the local form with its tail in a kernel of its own (k_pcg_tail) on a
                                             // single rank too; 0: sharded / multi-chunk solves keep the round-2 forms
                                             // (whole update in every workgroup of pass A, or k_pcg_update)
        int cam_chunk = 0;                   // > 0: chunk length of the camera-major kernels
        int pcg_guess_bias = 0;              // added to the number of speculatively enqueued PCG iterations
        int trace_pcg = 0, trace_stalls = 0, trace_timing = 0;   // stderr diagnostics
        int wait_deadline_s = 120;           // a hand-off that does not arrive within this many seconds fails the solve (-3)
        int p2p_delay_ms = 0;                // test: sleep this long before the first collective of a solve
        int p2p_timeout_ms = 0;              // test: > 0 overrides both time-outs of the direct all-reduce
    } dbg;

    DevBuf cam_idx, pt_idx, pt_ptr, uv;
    DevView ranges, wsteps, steps;
    int n_steps = 0;
    // camera-major order of the same observations (structure only): per-camera sums without atomics
    DevBuf cm_perm, cm_pt, cm_uv, cam_partial;
    DevBuf sort_hist, fixed_dev;             // device-side camera-major sort: [kSortSlices][C] counts -> offsets; held cameras
    DevView cam_ptr_dev;                     // [C + 1]
    DevView cam_chunks, cam_chunk_ptr;
    int n_chunks = 0;
    // a second chunk table for pass B of the Schur product alone (many points): every camera's list cut at the eight
    // point-range boundaries, chunk 8 c + k on XCD k (see set_problem)
    DevView cam_chunks_b, cam_chunk_ptr_b;
    int n_chunks_b = 0;
    bool xcd_b = false;
    std::vector<int4> host_chunks_b;
    std::vector<int> host_chunk_ptr_b;
    bool cam_multi = false;                  // some camera has more than one chunk: k_cam_combine runs
    DevBuf xa, xb, tabA, tabB, r, J, t1;     // ONE Jacobian / residual buffer set
    DevBuf rhsrec;                           // [P][kRhsRec]: what k_cam_rhs_diag gathers (written by k_prep)
    DevBuf V, Vinv, gp, e, recA, recB;       // rec: point records X Y Z | z (k_fill_rec), one per parameter vector
    DevBuf edge;                             // pieces of the point rows cut by K1's tiles (PointBlocksOut)
    DevBuf g, si, sg, p;                     // n-vectors; p = [dc | dp]
    DevBuf Dc, Minv, vecs, vtmp, vcm;               // camera-sized, plane-major [k][C]; vecs = 2 sets x (x r p s u)
    DevBuf part, ctrl;
    DevBuf arena_own;
    double* arena = nullptr;                 // [acc 6C | sd 21C | Ugc 27C | 32 scalars]; acc | sd are plane-major [27][C]
    int64_t arena_doubles = 0;
    sfmba_allreduce_fn ar_fn = nullptr;
    void* ar_ctx = nullptr;
    sfmba_print_fn print_fn = nullptr;       // verbose = 2 lines go here (null: stdout of the C library)
    void* print_ctx = nullptr;
    ncclComm_t comm = nullptr;               // native RCCL communicator (sfmba_comm_init)
    int64_t n_collectives = 0, n_launches = 0;
    // direct all-reduce over peer-mapped staging buffers (k_p2p_allreduce); preferred over RCCL / the
    // callback for every vector that fits a slot
    struct P2p {
        bool ready = false;
        int rank = 0, world = 0;
        int64_t stride = 0;                  // doubles per slot
        void* own = nullptr;                 // this rank's staging buffer (flags, then data)
        void* opened[kP2pMaxRanks] = {};     // peers' buffers as mapped here (null for own)
        double* data[kP2pMaxRanks] = {};
        unsigned long long* flags[kP2pMaxRanks] = {};
        double* camdata[kP2pMaxRanks] = {};  // [2][W][6 C]: the product's per-camera exchange inside pass B (CamExchange)
        unsigned* words = nullptr;           // [0] ticket, [1] error, [2..3] uint64 count of performed collectives
        int64_t calls = 0;
        bool first_in_solve = true;          // the next collective is the rendezvous of a solve (long timeout)
        // agreed over the link itself at attach (every rank holds the same two values):
        int shared_device = 1;               // ranks whose handle sits on the same physical GPU as this one's (rehearsals)
        bool any_multi = false;              // some rank's shard has a camera of several chunks / the XCD-aware chunk table
    } p2p;
    double* h_scal = nullptr;                // pinned
    // set_problem: converted arrays of the current problem in pinned memory (upload source, and what the next
    // call is compared with), host copies of the structure tables, worker threads
    struct Stage { PinnedBuf uv, ci, pi, perm, ptr, uvf, tables, ci16, uv16; } stage;
    DevBuf ci16_dev, uv16_dev;               // packed upload of the observation arrays (k_unpack_obs)
    struct Prev { bool valid = false; bool f32 = false; bool packed = false; int64_t N = 0, P = 0; } prev;
    std::vector<int2> host_ranges, host_wsteps, host_steps;
    std::vector<int4> host_chunks;
    std::vector<int> host_chunk_ptr;
    HostPool pool;
    int64_t obs_reused = 0, obs_uploaded = 0;    // of the last sfmba_set_problem
    double* mbox = nullptr;                  // coherent pinned block the device posts the hand-off into (Mailbox)
    double* mbox_dev = nullptr;              // its device-visible address
    unsigned long long mbox_seq = 0;
    Mailbox post{};                          // set while the launch that ends a hand-off is enqueued; else empty
    double* h_x = nullptr;                   // pinned staging of the parameter vector
    size_t h_x_doubles = 0;
    // Large problems (>= 2M parameters): the result leaves the device while the solve still runs -- every trial point is
    // copied to a pinned mirror of its buffer on a second stream, behind the launch that wrote it and beside the
    // evaluation of that point, so that the accepted x of the LAST iteration is already on the host when the solve ends
    // (24 MB = 0.45 ms at 3M parameters).  Not below that size: there the runtime performs the copy as a blit KERNEL,
    // which shares the CUs with the residual + Jacobian kernel it overlaps (K1 30 -> 32 us at 306k parameters, 60 us
    // under rocprofv3).  (Also tried there: the result in four chunks whose staging copies overlap the later chunks'
    // DMA -- 145 us against 130-150 us for one copy + a four-thread staging copy: four blit launches, no gain.)
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_written = nullptr, ev_copied[2] = {nullptr, nullptr};
    PinnedBuf mirror[2];
    unsigned long long mirror_tag[2] = {0, 0}, copy_count = 0, x_tag = 0;
    bool mirror_on = false;
    const double* skip = nullptr;         // device flag gating speculative trial launches (sfmba_solve); else null
    double pcg_tol = 0.0; int pcg_cap = 0; // options of the running PCG (fused launch 0 writes the control block)
    bool pcg_fused = false;               // PCG update fused into the launch of pass A (v in LDS, C <= 1024)
    bool pcg_local = false;               // ... with the per-camera bookkeeping in pass B (one rank, single-chunk cameras)
    bool pcg_local2 = false;              // the same bookkeeping with the light update as a kernel of its own (> 1024 cameras)
    bool pcg_b_owed = false;              // the pass B behind the last enqueued pass A was left out (pcg_enqueue)
    bool pcg_a_owed = false;              // ... and that pass A too: k_backsub does its update (FinalUpdate)
    bool pcg_inline = false;              // sharded, direct link, single-chunk cameras: the product's all-reduce runs inside
                                          // pass B, camera by camera (CamExchange), and the iteration is the local form
    bool pcg_split = false;               // local form whose per-camera tail runs behind the reduction (k_p2p_pcg / k_pcg_tail)
                                          // instead of inside pass B: sharded solves, cameras of several chunks
    DevBuf pcg_part;                      // [4][C] partial dot products of the local form
    int pcg_hint = 0;                     // largest PCG iteration count a solve on this handle has needed
    std::vector<int> pcg_hist;            // PCG iterations of outer iteration k in the previous solve on this handle: the
                                          // reference solves a slightly grown problem from a nearby start call after
                                          // call (sfm.py:59-71), and the counts repeat; with a record the speculative
                                          // batch is that count (+1 launch for the fused update), without the spare
    int64_t hist_C = 0, hist_P = 0, hist_N = 0;   // the problem pcg_hist was recorded on
    unsigned long long hist_sig = 0;         // ... and its content: problem generation + a checksum of the start vector
    unsigned long long problem_gen = 0;      // raised by every sfmba_set_problem whose arrays differ from the previous ones
    bool use_rhsrec = false;                 // the rhs + preconditioner pass gathers its own 128-byte records (many points)
    bool solved = false;
    bool transport_dropped = false;          // sfmba_set_problem tore down an active transport: the next compute call
                                             // fails until one is set up again (or single-rank use is acknowledged)
    std::vector<const void*> lds_ready;      // kernels already opted in to 160 KiB dynamic LDS
    int last_pcg_iters = 0;

    double* x = nullptr;                     // current / trial parameter vectors (alias xa/xb)
    double* x_new = nullptr;
    double* tab = nullptr;
    double* tab_new = nullptr;
    double* rec = nullptr;                   // point records of x / x_new (swapped together with them)
    double* rec_new = nullptr;

    double* acc() const { return arena; }    // product of the implicit Schur complement / reduced rhs term (6C)
    double* sd() const { return arena + 6 * C; }       // diagonal blocks of W Vinv W^T (Schur-diagonal preconditioner)
    double* Ugc() const { return arena + 27 * C; }
    double* scal() const { return arena + 54 * C; }
    int pcg_L = 0;                           // launches (sweep+update pairs) since pcg_start
    int red_bc = 1, red_grid = 2;            // block split of k_update_scale's reduction (cameras | points)
    int scale_pts = 1;                       // points per thread of k_update_scale
    double* partB() const { return part.as<double>() + (size_t)kPartRows * kNQ; }
    bool pending_scale_sums = false;         // k_update_scale ran, its final sums ride with the next k_jdot
};

namespace {

int fail(sfmba_handle* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf;
    return code;
}

#define HIPCHK(h, call)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(h, e_ == hipErrorOutOfMemory ? -4 : -3, "%s failed: %s (%s:%d)", #call,  \
                        hipGetErrorString(e_), __FILE__, __LINE__);                             \
    } while (0)

// after every kernel launch: count it (sfmba_get_counters) and pick up a launch failure
#define LAUNCHED(h)                            \
    do {                                       \
        ++(h)->n_launches;                     \
        HIPCHK(h, hipGetLastError());          \
    } while (0)

#define CHK(expr)                  \
    do {                           \
        int rc_ = (expr);          \
        if (rc_ != 0) return rc_;  \
    } while (0)

int enter(sfmba_handle* h) {
    if (!h) return -1;
    h->err.clear();
    HIPCHK(h, hipSetDevice(h->device));
    return 0;
}

// All-reduce `count` doubles of the exchange arena in place over the ranks, on the handle's stream:
// natively with RCCL when a communicator is set, else through the host callback, else a no-op.
void p2p_release(sfmba_handle* h);
void p2p_close_peers(sfmba_handle* h);
bool multi_rank(const sfmba_handle* h) { return h->p2p.ready || h->comm != nullptr || h->ar_fn != nullptr; }
const unsigned* p2p_error_word(const sfmba_handle* h) { return h->p2p.ready ? h->p2p.words + 1 : nullptr; }

constexpr long long kP2pFirstTicks = 6000000000ll, kP2pSteadyTicks = 3000000000ll;      // 60 s, 30 s at 100 MHz
constexpr size_t kP2pFlagBytes = sizeof(unsigned long long) * 2 * kP2pMaxRanks * kP2pFlagStride;

void p2p_fill_args(sfmba_handle* h, P2pArgs& a) {
    auto& p = h->p2p;
    for (int q = 0; q < p.world; ++q) { a.data[q] = p.data[q]; a.flags[q] = p.flags[q]; }
    a.rank = p.rank; a.world = p.world; a.stride = p.stride;
    a.seq = reinterpret_cast<unsigned long long*>(p.words + 2);
    a.ticket = p.words; a.error = p.words + 1;
    // Ticks of the 100 MHz wall clock.  The FIRST collective of a solve is the rendezvous of ranks that entered
    // sfmba_solve at different times (Python skew, first-use code-object loads; no barrier is required before a solve):
    // it waits up to 60 s.  Steady state: 30 s -- the devices run in lockstep there, but a host thread that stalls between
    // two launches (a cold page of a runtime library on a fresh machine, a descheduled process) holds its peers up for as
    // long; the bound only decides how soon a dead peer is noticed (round 4: 3 s gave one spurious failure in a first solve
    // on a fresh box).
    a.timeout = p.first_in_solve ? kP2pFirstTicks : kP2pSteadyTicks;
    if (h->dbg.p2p_timeout_ms > 0) a.timeout = 100000ll * h->dbg.p2p_timeout_ms;       // test hook
    p.first_in_solve = false;
}

// sharded over the direct link, every camera a single chunk: per-camera sums are all-reduced by the workgroup that
// forms them (CamExchange) instead of by a collective launch behind the kernel
// -- on EVERY rank (the chunking is a property of the shard; ranks that disagreed would wait for each other in different
// kernels), and practically never between ranks that SHARE one GPU (p2p_inline_ok).  With a GPU per rank a waiting
// workgroup costs its own device a slot and nothing else.  On a shared device the ranks compete for the CUs: a camera
// workgroup that waits for its peer holds registers on its CU, and the peer -- if it is one kernel behind, which two
// processes drift apart by easily -- first has to run pass A, whose 1024-thread workgroups need the whole register file of
// a CU.  Three hundred waiting workgroups sit on every CU of the card, pass A of the other rank then never starts, and
// both time out (seen at 2 x 300 and 2 x 1000 cameras, one run in four; the collective LAUNCHES never had the problem:
// they are one or a few workgroups).  So rehearsals on one device take the exchange inside the kernels only while the
// waiting workgroups of all other ranks leave half the CUs alone -- small tests -- and the collective launches otherwise.
constexpr int64_t kSharedDeviceCams = 128;         // (ranks - 1) x cameras: at most half of the 256 CUs hold a waiting workgroup
bool p2p_inline_ok(const sfmba_handle* h) {
    const auto& p = h->p2p;
    return p.ready && !p.any_multi && !(p.shared_device > 1 && h->C * (p.shared_device - 1) > kSharedDeviceCams);
}
// K3 and the rhs pass over the XCD-aware chunk table, one wave per chunk (k_cam_blocks_w, k_cam_rhs_diag_w)
bool xcd_cam(const sfmba_handle* h) { return h->xcd_b && h->dbg.xcd_cam != 0; }
bool cam_inline(const sfmba_handle* h) { return p2p_inline_ok(h) && !h->cam_multi && !xcd_cam(h) && h->dbg.pcg_inline != 0; }

CamExchange cam_exchange(sfmba_handle* h) {
    CamExchange cx{};
    auto& p = h->p2p;
    for (int q = 0; q < p.world; ++q) cx.data[q] = p.camdata[q];
    cx.rank = p.rank; cx.world = p.world; cx.C = (int)h->C; cx.error = p.words + 1;
    cx.timeout = p.first_in_solve ? kP2pFirstTicks : kP2pSteadyTicks;      // (as p2p_fill_args)
    if (h->dbg.p2p_timeout_ms > 0) cx.timeout = 100000ll * h->dbg.p2p_timeout_ms;
    p.first_in_solve = false;
    return cx;
}

int p2p_allreduce(sfmba_handle* h, double* ptr, int64_t count, int op, const int* cancel,
                  unsigned long long max_mask = 0, const Piggyback* rider = nullptr, const Mailbox* post = nullptr,
                  const double* skip = nullptr) {
    auto& p = h->p2p;
    P2pArgs a{};
    p2p_fill_args(h, a);
    a.cancel = cancel;
    a.max_mask = max_mask;
    if (rider) a.rider = *rider;
    if (post) a.post = *post;
    a.skip = skip;
    if ((rider || post) && count > 512) return fail(h, -1, "rider / post need a single-workgroup collective");
    const int grid = (int)std::min<int64_t>(kP2pMaxBlocks, std::max<int64_t>(1, (count + 511) / 512));
    hipLaunchKernelGGL(k_p2p_allreduce, dim3(grid), dim3(256), 0, h->stream, ptr, (int)count, op, a);
    LAUNCHED(h);
    ++p.calls;
    return 0;
}

// `cancel` (device pointer, may be null): when it reads non-zero the collective is void on every rank -- it
// follows a PCG launch that did nothing.  Only the direct path can act on it; RCCL and the callback reduce
// the stale vector, which is harmless.
int exchange(sfmba_handle* h, double* ptr, int64_t count, int op, const int* cancel = nullptr) {
    if (h->p2p.ready && count <= h->p2p.stride) {
        CHK(p2p_allreduce(h, ptr, count, op, cancel));
        ++h->n_collectives;
        return 0;
    }
    if (h->comm) {
        RcclApi* api = rccl_api();
        const ncclResult_t rc = api->AllReduce(ptr, ptr, (size_t)count, ncclDouble, op == 0 ? ncclSum : ncclMax,
                                               h->comm, h->stream);
        if (rc != ncclSuccess) return fail(h, -5, "ncclAllReduce failed: %s", api->GetErrorString(rc));
        ++h->n_collectives;
        return 0;
    }
    if (!h->ar_fn) return 0;
    if (h->ar_fn(h->ar_ctx, ptr, count, op) != 0) return fail(h, -5, "all-reduce callback failed");
    ++h->n_collectives;
    return 0;
}

// Dynamic LDS above 64 KiB needs the per-kernel opt-in; it is set ONCE per kernel (to the full
// 160 KiB) because hipFuncSetAttribute costs tens of microseconds to a millisecond per call.
template <class Kern>
int set_lds(sfmba_handle* h, Kern k, size_t bytes) {
    if (bytes <= 64 * 1024) return 0;
    const void* fn = reinterpret_cast<const void*>(k);
    if (std::find(h->lds_ready.begin(), h->lds_ready.end(), fn) != h->lds_ready.end()) return 0;
    HIPCHK(h, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsDynMax));
    h->lds_ready.push_back(fn);
    return 0;
}

// inverse point blocks: an array of their own, or (record layout 2) a slot of the CURRENT parameter vector's point
// records -- written by k_prep / k_point_prep for h->rec before every consumer of the iteration
double* vinv_ptr(const sfmba_handle* h) { return kVinvInRec < 0 ? h->Vinv.as<double>() : h->rec + kVinvInRec; }

MixedPrep mixed_prep(const sfmba_handle* h, const double* tab) {
    if (!h->mixed) return MixedPrep{nullptr, nullptr, nullptr, nullptr, Origin{0.0, 0.0, 0.0}};
    return MixedPrep{tab, h->rt32.as<float>(), h->rec32.as<float>(), h->rtd.as<double>(), h->origin};
}
// fp32 operands of (x, tab) outside a solve (inside, k_prep writes them)
int launch_mixed_prep(sfmba_handle* h, const double* x, const double* tab) {
    if (!h->mixed) return 0;
    const int bc = (int)((h->C + 255) / 256), bp = (int)((h->P + 255) / 256);
    hipLaunchKernelGGL(k_mixed_prep, dim3(bc + bp), dim3(256), 0, h->stream, mixed_prep(h, tab), x + 6 * h->C, (int)h->C, (int)h->P, bc);
    LAUNCHED(h);
    return 0;
}

ObsArrays obs_arrays(const sfmba_handle* h) {
    return ObsArrays{h->cam_idx.as<int>(), h->pt_idx.as<int>(), h->pt_ptr.as<int>(),
                     h->J.as<double>(), h->ld, h->f32 ? 1 : 0};
}

StepTable step_table(const sfmba_handle* h) {
    return StepTable{h->wsteps.as<int2>(), h->steps.as<int2>(), h->n_ranges};
}

int grid_1d(int64_t n, int block, int cap) {
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

// Host/device hand-off: poll the stream instead of sleeping in hipStreamSynchronize.  The solver
// hands control back to the host two to three times per outer iteration for ~30 us of GPU work each;
// a blocking wait that parks the thread costs up to a millisecond per wake-up on an idle host.
void report_stall(const sfmba_handle* h, const char* where, double seconds) {   // trace_stalls: waits longer than 2 ms
    if (h->dbg.trace_stalls && seconds > 2e-3) fprintf(stderr, "sfmba: waited %.2f ms in %s\n", 1e3 * seconds, where);
}

int wait_stream(sfmba_handle* h) {
    const double t0 = now_s();
    for (;;) {
        const hipError_t e = hipStreamQuery(h->stream);
        if (e == hipSuccess) { report_stall(h, "wait_stream", now_s() - t0); return 0; }
        if (e != hipErrorNotReady) return fail(h, -3, "hipStreamQuery failed: %s", hipGetErrorString(e));
        if (now_s() - t0 > 0.05) break;          // long wait: stop burning the core
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    report_stall(h, "wait_stream (blocking)", now_s() - t0);
    return 0;
}

// The hand-off of an outer iteration: the device posts scalars + PCG control block into the mailbox and
// raises its sequence number (post_mailbox); the host polls that word and nothing else: every
// hipStreamQuery on a busy stream leaves a marker packet in the queue, and a dozen of them behind the
// speculative launches cost 5.8 us before the first kernel of the next iteration.  The post of an iteration
// arrives a few hundred microseconds after the host has finished enqueueing it; only past 5 ms is the stream
// queried (every millisecond), to notice a failed launch instead of spinning forever.
int p2p_timed_out(sfmba_handle* h, unsigned code);
int mailbox_arrived(sfmba_handle* h, const char* where, double t0) {
    report_stall(h, where, now_s() - t0);
    if (h->mbox[kMboxErr] != 0.0)          // published with every post: a direct all-reduce gave up waiting for a peer
        return p2p_timed_out(h, (unsigned)h->mbox[kMboxErr]);
    return 0;
}

// the error word of the direct link (k_p2p_allreduce: 1; cam_exchange_value: which exchange, value and camera) as text
int p2p_timed_out(sfmba_handle* h, unsigned code) {
    if ((code & 0x80000000u) == 0u) return fail(h, -5, "a direct all-reduce timed out waiting for a peer rank");
    static const char* const kinds[8] = {"?", "?", "Schur product (pass B)", "camera blocks (K3)", "reduced right-hand side", "?", "?", "?"};
    return fail(h, -5, "a direct all-reduce timed out waiting for a peer rank: the per-camera exchange of the %s, value %u of "
                       "camera %u", kinds[(code >> 28) & 7u], (code >> 23) & 31u, code & 0x7FFFFFu);
}

int wait_mailbox(sfmba_handle* h, unsigned long long seq) {
    unsigned long long* word = reinterpret_cast<unsigned long long*>(h->mbox + kMboxSeq);
    const double t0 = now_s();
    double t_check = t0 + 5e-3;
    for (int spin = 0;; ++spin) {
        if (__atomic_load_n(word, __ATOMIC_ACQUIRE) == seq) return mailbox_arrived(h, "wait_mailbox", t0);
        __builtin_ia32_pause();
        if ((spin & 15) != 15) continue;
        const double t = now_s();
        if (t < t_check) continue;
        const hipError_t e = hipStreamQuery(h->stream);
        if (e == hipSuccess) {                               // everything enqueued has run: the post is visible
            if (__atomic_load_n(word, __ATOMIC_ACQUIRE) == seq) return mailbox_arrived(h, "wait_mailbox (stream idle)", t0);
            return fail(h, -3, "hand-off mailbox was not written");
        }
        if (e != hipErrorNotReady) return fail(h, -3, "hipStreamQuery failed: %s", hipGetErrorString(e));
        if (t - t0 > h->dbg.wait_deadline_s)   // a kernel that never drains, a peer that died inside a library collective
            return fail(h, -3, "hand-off not posted within %d s: the device queue is stuck", h->dbg.wait_deadline_s);
        t_check = t + 1e-3;
    }
}

// ---- kernel launch wrappers --------------------------------------------------------------------

// camera table and point records of a freshly uploaded parameter vector
int launch_cam_table(sfmba_handle* h, const double* x, double* tab, double* rec) {
    hipLaunchKernelGGL(k_cam_table, dim3((h->C + 255) / 256), dim3(256), 0, h->stream, x, (int)h->C, tab);
    LAUNCHED(h);
    hipLaunchKernelGGL(k_fill_rec, dim3((unsigned)((3 * h->P + 255) / 256)), dim3(256), 0, h->stream, x + 6 * h->C, (int)h->P, rec);
    LAUNCHED(h);
    return 0;
}

// residual (+Jacobian) sweep at x (camera table must be current); leaves the
// sum r^2 partials in `part` and returns the number of partials.  With ev0/ev1 the dispatch itself
// is bracketed (hipExtLaunchKernelGGL: start/stop taken from the kernel's own dispatch, as rocprofv3
// does), so the measured duration is the kernel's and not host launch latency.
PointBlocksOut point_blocks_out(sfmba_handle* h) {
    return PointBlocksOut{h->V.as<double>(), h->gp.as<double>(), h->edge.as<double>()};
}

template <bool LDS, bool JAC, bool STORE_R, bool F32, bool BLOCKS>
int launch_resjac_b(sfmba_handle* h, const double* x, const double* tab, int grid, size_t lds,
                    hipEvent_t ev0, hipEvent_t ev1) {
    const double* pts = x + 6 * h->C;
    const PointBlocksOut pb = BLOCKS ? point_blocks_out(h) : PointBlocksOut{nullptr, nullptr, nullptr};
    auto kern = k_resjac<LDS, JAC, STORE_R, F32, BLOCKS>;
    if constexpr (JAC && BLOCKS && LDS && STORE_R) {           // J-free iteration: the blocks feed the point sums only
        if (h->jfree) kern = k_resjac<LDS, JAC, STORE_R, F32, BLOCKS, false>;
    }
    CHK(set_lds(h, kern, lds));
    if (ev0) {
        hipExtLaunchKernelGGL(kern, dim3(grid), dim3(kSweepThreads), (uint32_t)lds, h->stream, ev0, ev1, 0u, tab, pts,
                              (const int*)h->cam_idx.as<int>(), (const int*)h->pt_idx.as<int>(),
                              (const double*)h->uv.as<double>(), h->r.as<double>(), h->J.as<double>(),
                              (int)h->N, h->ld, (int)h->C, h->K, h->part.as<double>(), h->skip, pb);
    } else {
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kSweepThreads), lds, h->stream, tab, pts,
                           h->cam_idx.as<int>(), h->pt_idx.as<int>(), h->uv.as<double>(),
                           h->r.as<double>(), h->J.as<double>(), (int)h->N,
                           h->ld, (int)h->C, h->K, h->part.as<double>(), h->skip, pb);
    }
    LAUNCHED(h);
    return 0;
}
template <bool LDS, bool JAC, bool STORE_R, bool F32>
int launch_resjac_v(sfmba_handle* h, const double* x, const double* tab, int grid, size_t lds,
                    hipEvent_t ev0, hipEvent_t ev1, bool blocks) {
    if constexpr (JAC) {
        if (blocks) return launch_resjac_b<LDS, JAC, STORE_R, F32, true>(h, x, tab, grid, lds, ev0, ev1);
    }
    return launch_resjac_b<LDS, JAC, STORE_R, F32, false>(h, x, tab, grid, lds, ev0, ev1);
}

// `blocks`: the Jacobian launch also leaves V_p, g_p of x (the point half of the normal equations; the camera half
// and the pieces of runs cut by tile boundaries follow in launch_normal_blocks)
template <bool JAC, bool STORE_R>
int launch_resjac(sfmba_handle* h, const double* x, const double* tab, int* nparts,
                  hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr, bool blocks = JAC) {
    const int grid = grid_1d(h->N, kSweepThreads, h->n_cu);
    *nparts = grid;
    const size_t lds = (size_t)h->C * kCamRow * sizeof(double);
    if (h->lds_tab)
        return h->f32 ? launch_resjac_v<true, JAC, STORE_R, true>(h, x, tab, grid, lds, ev0, ev1, blocks)
                      : launch_resjac_v<true, JAC, STORE_R, false>(h, x, tab, grid, lds, ev0, ev1, blocks);
    const size_t slabs = sizeof(double) * kRowSlabDoubles * kWavesPerSweepBlock;       // rows gathered through the LDS
    return h->f32 ? launch_resjac_v<false, JAC, STORE_R, true>(h, x, tab, grid, slabs, ev0, ev1, blocks)
                  : launch_resjac_v<false, JAC, STORE_R, false>(h, x, tab, grid, slabs, ev0, ev1, blocks);
}

// sum of `nparts` partial rows of width nq into the exchange scalars starting at slot `slot`
int launch_finish(sfmba_handle* h, const double* part, int nparts, int nq, int slot) {
    FinishJob job{};
    job.row0[0] = 0; job.nrows[0] = nparts;
    for (int k = 0; k < kFinishCols; ++k) { job.slot[0][k] = slot + k; job.slot[1][k] = -1; }
    hipLaunchKernelGGL(k_finish, dim3(1), dim3(64 * nq), 0, h->stream, part, job, nq, 0, h->scal(), h->skip,
                       h->post);
    LAUNCHED(h);
    return 0;
}

// the two-slice (cameras | points) reduction of k_update_scale; the point slice
// publishes only quantities [q_lo, q_hi] (the others keep their already rank-reduced values)
FinishJob slices_job(const sfmba_handle* h, int q_lo, int q_hi) {
    FinishJob job{};
    job.row0[0] = 0;         job.nrows[0] = h->red_bc;
    job.row0[1] = h->red_bc; job.nrows[1] = h->red_grid - h->red_bc;
    for (int k = 0; k < kFinishCols; ++k) { job.slot[0][k] = -1; job.slot[1][k] = -1; }
    for (int k = 0; k < kNQ; ++k) {
        job.slot[0][k] = kCamSlot + k;
        job.slot[1][k] = (k >= q_lo && k <= q_hi) ? kPointSlot[k] : -1;
    }
    return job;
}
Piggyback slices_rider(sfmba_handle* h, int q_lo, int q_hi) {
    return Piggyback{h->part.as<double>(), h->scal(), slices_job(h, q_lo, q_hi), 2, kNQ, 1};
}

int launch_finish_slices(sfmba_handle* h, int q_lo, int q_hi) {
    const FinishJob job = slices_job(h, q_lo, q_hi);
    hipLaunchKernelGGL(k_finish, dim3(2), dim3(64 * kNQ), 0, h->stream, h->part.as<double>(), job, kNQ, 1, h->scal(),
                       (const double*)nullptr, Mailbox{});
    LAUNCHED(h);
    return 0;
}

CamMajor cam_major(const sfmba_handle* h) {
    return CamMajor{h->cam_chunks.as<int4>(), h->cm_pt.as<int>(), h->cm_uv.as<double>()};
}

// chunk rows of cameras with several chunks -> out[c * cs + col * ks]
int launch_cam_combine(sfmba_handle* h, int ncols, double* out, int cs, int ks, const double* skip, const int* done,
                       bool table_b = false) {
    if (!(table_b ? h->xcd_b : h->cam_multi)) return 0;
    hipLaunchKernelGGL(k_cam_combine, dim3((int)((h->C * ncols + 255) / 256)), dim3(256), 0, h->stream,
                       (table_b ? h->cam_chunk_ptr_b : h->cam_chunk_ptr).as<int>(), h->cam_partial.as<double>(), (int)h->C, ncols,
                       out, cs, ks, skip, done);
    LAUNCHED(h);
    return 0;
}

// K3 at (x, tab): [U_c | g_c] over the camera-major order, blocks recomputed from the camera table and the point
// records.  V_p, g_p were left by the residual+Jacobian launch at the same x (launch_resjac with `blocks`), except
// for the runs its 64-observation tiles cut: their pieces are added by a few extra workgroups of this launch.
// `cost_parts` > 0: the launch also sums that many cost partials (left in `part` by the residual launch before it) into
// scalar slot 0 and posts the hand-off `mb` (one more rider workgroup instead of a k_finish launch)
template <bool F32>
int launch_normal_blocks_v(sfmba_handle* h, const double* x, const double* tab, const double* rec, int cost_parts,
                           const Mailbox& mb) {
    (void)x;
    const int tiles = (int)((h->N + 63) / 64);
    const int riders = (tiles + kCamThreads - 1) / kCamThreads;
    CamExchange cx{};
    if (cam_inline(h)) { cx = cam_exchange(h); ++h->p2p.calls; ++h->n_collectives; }
    Piggyback fin{};
    if (cost_parts > 0) {
        fin = Piggyback{h->part.as<double>(), h->scal(), FinishJob{}, 1, 1, 0};
        fin.job.row0[0] = 0; fin.job.nrows[0] = cost_parts;
        for (int k = 0; k < kFinishCols; ++k) { fin.job.slot[0][k] = k; fin.job.slot[1][k] = -1; }
    }
    if (xcd_cam(h)) {                                           // many points: one wave per chunk of the XCD-aware table
        const CamMajor cmb{h->cam_chunks_b.as<int4>(), h->cm_pt.as<int>(), h->cm_uv.as<double>()};
        const int wgrid = h->n_chunks_b / kWaveChunkCams;
        hipLaunchKernelGGL((k_cam_blocks_w<F32>), dim3(wgrid + riders + (cost_parts > 0 ? 1 : 0)), dim3(kCamThreads), 0, h->stream,
                           cmb, tab, rec, h->K, h->cam_partial.as<double>(), h->skip, wgrid,
                           (const int*)h->pt_idx.as<int>(), (int)h->N, point_blocks_out(h), fin, mb);
        LAUNCHED(h);
        hipLaunchKernelGGL(k_cam_combine_w<27>, dim3((unsigned)((27 * h->C + 255) / 256)), dim3(256), 0, h->stream,
                           (const double*)h->cam_partial.as<double>(), (int)h->C, h->Ugc(), 27, 1, (const int*)nullptr,
                           (const double*)h->skip);
        LAUNCHED(h);
        return exchange(h, h->Ugc(), 27 * h->C, 0);
    }
    hipLaunchKernelGGL((k_cam_blocks<F32>), dim3(h->n_chunks + riders + (cost_parts > 0 ? 1 : 0)), dim3(kCamThreads), 0, h->stream,
                       cam_major(h), tab, rec, h->K, h->Ugc(), h->cam_partial.as<double>(), h->skip, (int)h->n_chunks,
                       (const int*)h->pt_idx.as<int>(), (int)h->N, point_blocks_out(h), fin, mb, cx);
    LAUNCHED(h);
    CHK(launch_cam_combine(h, 27, h->Ugc(), 27, 1, h->skip, nullptr));
    if (cx.world > 1 || cam_inline(h)) return 0;               // [U | g_c] was summed over the ranks camera by camera
    return exchange(h, h->Ugc(), 27 * h->C, 0);
}
int launch_normal_blocks(sfmba_handle* h, const double* x, const double* tab, const double* rec, int cost_parts = 0,
                         const Mailbox& mb = Mailbox{}) {
    return h->f32 ? launch_normal_blocks_v<true>(h, x, tab, rec, cost_parts, mb)
                  : launch_normal_blocks_v<false>(h, x, tab, rec, cost_parts, mb);
}

// Pass A of the implicit Schur product (z_p for every point).  Inside the two-kernel PCG: vin = base of the
// vector sets, ctrl2 / L select the set on the device; standalone (test entry): vin = the vector itself,
// plane-major when it is staged in LDS, camera-major otherwise; ctrl2 = nullptr.
int launch_point_sweep(sfmba_handle* h, const double* vin, const PcgCtrl* ctrl2, int L) {
    const int grid = (h->n_ranges + kWavesPerSweepBlock - 1) / kWavesPerSweepBlock;
    if (h->mixed && h->sweep_rc) {        // fp32 operands, table in LDS
        const size_t lds = sizeof(float) * kRc32Row * (size_t)h->C;
        auto kern = h->mixed_b ? k_point_sweep_rc32<false, false, false> : k_point_sweep_rc32<false, false, true>;
        CHK(set_lds(h, kern, lds));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kSweepThreads), lds, h->stream, step_table(h), (const int*)h->cam_idx.as<int>(),
                           (const int*)h->pt_idx.as<int>(), (const double*)h->tab, (const float*)h->rt32.as<float>(), h->K, vin,
                           (const double*)vinv_ptr(h), h->rec32.as<float>(), (const double*)h->acc(), (int)h->C, ctrl2, L,
                           PcgFused{}, (const float*)nullptr, h->rec);
        LAUNCHED(h);
        return 0;
    }
    if (h->mixed && h->sweep_rc_g) {      // fp32 operands, table in global memory
        hipLaunchKernelGGL(k_rc_table32, dim3((unsigned)((h->C + 255) / 256)), dim3(256), 0, h->stream, (const double*)h->tab,
                           (const float*)h->rt32.as<float>(), vin, ctrl2, L, (int)h->C, h->rctab32.as<float>());
        LAUNCHED(h);
        const size_t slabs = sizeof(float) * kRow32SlabFloats * kWavesPerSweepBlock;
        auto kern_g = h->mixed_b ? k_point_sweep_rc32<false, true, false> : k_point_sweep_rc32<false, true, true>;
        CHK(set_lds(h, kern_g, slabs));
        hipLaunchKernelGGL(kern_g, dim3(grid), dim3(kSweepThreads), slabs, h->stream, step_table(h),
                           (const int*)h->cam_idx.as<int>(), (const int*)h->pt_idx.as<int>(), (const double*)h->tab,
                           (const float*)h->rt32.as<float>(), h->K, vin, (const double*)vinv_ptr(h), h->rec32.as<float>(),
                           (const double*)h->acc(), (int)h->C, ctrl2, L, PcgFused{}, (const float*)h->rctab32.as<float>(), h->rec);
        LAUNCHED(h);
        return 0;
    }
    if (h->sweep_rc) {                    // recomputing form: vin plane-major (or the base of the vector sets)
        const size_t lds = sizeof(double) * kRcRow * (size_t)h->C;
        auto kern = k_point_sweep_rc<false>;
        CHK(set_lds(h, kern, lds));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kSweepThreads), lds, h->stream, step_table(h), (const int*)h->cam_idx.as<int>(),
                           (const int*)h->pt_idx.as<int>(), (const double*)h->tab, (const double*)(h->x + 6 * h->C), h->K, vin,
                           (const double*)vinv_ptr(h), h->rec, (const double*)h->acc(), (int)h->C, ctrl2, L,
                           PcgFused{}, (const double*)nullptr);
        LAUNCHED(h);
        return 0;
    }
    if (h->sweep_rc_g) {                  // the same with its table in global memory: one small launch builds it
        hipLaunchKernelGGL(k_rc_table, dim3((unsigned)((h->C + 255) / 256)), dim3(256), 0, h->stream, (const double*)h->tab, vin,
                           ctrl2, L, (int)h->C, h->rctab.as<double>());
        LAUNCHED(h);
        const size_t slabs = sizeof(double) * kRowSlabDoubles * kWavesPerSweepBlock;    // rows gathered through the LDS
        auto kern_g = k_point_sweep_rc<false, true>;
        CHK(set_lds(h, kern_g, slabs));
        hipLaunchKernelGGL(kern_g, dim3(grid), dim3(kSweepThreads), slabs, h->stream, step_table(h),
                           (const int*)h->cam_idx.as<int>(), (const int*)h->pt_idx.as<int>(), (const double*)h->tab,
                           (const double*)(h->x + 6 * h->C), h->K, vin, (const double*)vinv_ptr(h), h->rec,
                           (const double*)h->acc(), (int)h->C, ctrl2, L, PcgFused{}, (const double*)h->rctab.as<double>());
        LAUNCHED(h);
        return 0;
    }
    if (h->lds_vec) {
        const size_t lds = sizeof(double) * 6 * (size_t)h->C;
        auto kern = k_point_sweep<true, false>;
        CHK(set_lds(h, kern, lds));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kSweepThreads), lds, h->stream, step_table(h), obs_arrays(h), vin,
                           vinv_ptr(h), h->rec, (const double*)h->acc(), (int)h->C, ctrl2, L, PcgFused{});
    } else {
        hipLaunchKernelGGL((k_point_sweep<false, false>), dim3(grid), dim3(kSweepThreads), 0, h->stream, step_table(h),
                           obs_arrays(h), vin, vinv_ptr(h), h->rec, (const double*)h->acc(),
                           (int)h->C, ctrl2, L, PcgFused{});
    }
    LAUNCHED(h);
    return 0;
}

// pass A fused with the PCG update of the previous product (one launch)
int launch_pcg_fused(sfmba_handle* h, int L) {
    const int grid = (h->n_ranges + kWavesPerSweepBlock - 1) / kWavesPerSweepBlock;
    PcgFused pf{h->Dc.as<double>(), h->Minv.as<double>(), h->Ugc(), h->vecs.as<double>(), h->ctrl.as<PcgCtrl>(),
                h->pcg_tol, h->pcg_cap, (const double*)(h->scal() + kEtaSlot),
                h->pcg_local ? h->pcg_part.as<double>() : (double*)nullptr};
    if (h->mixed && h->sweep_rc) {
        const size_t lds32 = sizeof(float) * kRc32Row * (size_t)h->C;
        auto kern32 = h->mixed_b ? k_point_sweep_rc32<true, false, false> : k_point_sweep_rc32<true, false, true>;
        CHK(set_lds(h, kern32, lds32));
        hipLaunchKernelGGL(kern32, dim3(grid), dim3(kSweepThreads), lds32, h->stream, step_table(h),
                           (const int*)h->cam_idx.as<int>(), (const int*)h->pt_idx.as<int>(), (const double*)h->tab,
                           (const float*)h->rt32.as<float>(), h->K, (const double*)h->vecs.as<double>(),
                           (const double*)vinv_ptr(h), h->rec32.as<float>(), (const double*)h->acc(), (int)h->C,
                           (const PcgCtrl*)h->ctrl.as<PcgCtrl>(), L, pf, (const float*)nullptr, h->rec);
        LAUNCHED(h);
        return 0;
    }
    if (h->sweep_rc) {
        const size_t lds_rc = sizeof(double) * kRcRow * (size_t)h->C;
        auto kern_rc = k_point_sweep_rc<true>;
        CHK(set_lds(h, kern_rc, lds_rc));
        hipLaunchKernelGGL(kern_rc, dim3(grid), dim3(kSweepThreads), lds_rc, h->stream, step_table(h),
                           (const int*)h->cam_idx.as<int>(), (const int*)h->pt_idx.as<int>(), (const double*)h->tab,
                           (const double*)(h->x + 6 * h->C), h->K, (const double*)h->vecs.as<double>(),
                           (const double*)vinv_ptr(h), h->rec, (const double*)h->acc(), (int)h->C,
                           (const PcgCtrl*)h->ctrl.as<PcgCtrl>(), L, pf, (const double*)nullptr);
        LAUNCHED(h);
        return 0;
    }
    const size_t lds = sizeof(double) * 6 * (size_t)h->C;
    auto kern = k_point_sweep<true, true>;
    CHK(set_lds(h, kern, lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kSweepThreads), lds, h->stream, step_table(h), obs_arrays(h),
                       (const double*)h->vecs.as<double>(), vinv_ptr(h), h->rec,
                       (const double*)h->acc(), (int)h->C, (const PcgCtrl*)h->ctrl.as<PcgCtrl>(), L, pf);
    LAUNCHED(h);
    return 0;
}

// Pass B (camera-major): MODE 0  acc = sum Jc^T (Jc v - Jp z) with z from pass A; MODE 1  acc = -sum Jc^T Jp e.
// ctrl_done / set: see k_cam_schur.
template <int MODE>
int launch_cam_schur(sfmba_handle* h, const double* vin, const PcgCtrl* ctrl_done, int set, bool local = false,
                     bool inline_exchange = false) {
    const PcgLocal pl{h->Dc.as<double>(), h->Minv.as<double>(), h->vecs.as<double>(),
                      local ? h->pcg_part.as<double>() : (double*)nullptr};
    CamExchange cx{};
    if (MODE == 0 && inline_exchange) { cx = cam_exchange(h); ++h->p2p.calls; ++h->n_collectives; }
    // MODE 0 with the XCD-aware table: chunk 8 c + k runs on XCD k and gathers records of point range k only
    const bool tb = MODE == 0 && h->xcd_b;
    const CamMajor cm = tb ? CamMajor{h->cam_chunks_b.as<int4>(), h->cm_pt.as<int>(), h->cm_uv.as<double>()} : cam_major(h);
    const int grid = tb ? h->n_chunks_b : h->n_chunks;
    const MixedB mxb{h->rtd.as<double>()};
    if (tb) {                                       // one wave per chunk of the XCD-aware table
        const bool round = h->f32 && !h->sweep_rc && !h->sweep_rc_g;
        if (round) return fail(h, -1, "XCD-aware chunks need the recomputing form of pass A");
        const int wgrid = h->n_chunks_b / kWaveChunkCams;
        if (h->mixed_b)
            hipLaunchKernelGGL((k_cam_schur_w<true>), dim3(wgrid), dim3(kCamThreads), 0, h->stream, cm, (const double*)h->tab,
                               reinterpret_cast<const double*>(h->rec32.as<float>()), h->K, vin, (int)h->C,
                               h->cam_partial.as<double>(), ctrl_done, set, mxb);
        else
            hipLaunchKernelGGL((k_cam_schur_w<false>), dim3(wgrid), dim3(kCamThreads), 0, h->stream, cm, (const double*)h->tab,
                               (const double*)h->rec, h->K, vin, (int)h->C, h->cam_partial.as<double>(), ctrl_done, set, mxb);
        LAUNCHED(h);
        hipLaunchKernelGGL(k_cam_combine_w<6>, dim3((unsigned)((6 * h->C + 255) / 256)), dim3(256), 0, h->stream,
                           (const double*)h->cam_partial.as<double>(), (int)h->C, h->acc(), 1, (int)h->C,
                           ctrl_done ? &ctrl_done->done : (const int*)nullptr, (const double*)nullptr);
        LAUNCHED(h);
        return 0;
    }
    if (h->f32 && !h->sweep_rc && !h->sweep_rc_g)   // pass A applies the stored fp32 blocks: pass B rounds its own the same way
        hipLaunchKernelGGL((k_cam_schur<MODE, true>), dim3(grid), dim3(kCamThreads), 0, h->stream, cm,
                           (const double*)h->tab, (const double*)h->rec, h->K, vin, (int)h->C, h->acc(),
                           h->cam_partial.as<double>(), ctrl_done, set, pl, mxb, cx);
    else if (MODE == 0 && h->mixed_b)               // fp32 operands: the same rounded R, T - o, X - o, a', u_T as pass A
        hipLaunchKernelGGL((k_cam_schur<0, false, true>), dim3(grid), dim3(kCamThreads), 0, h->stream, cm,
                           (const double*)h->tab, reinterpret_cast<const double*>(h->rec32.as<float>()), h->K, vin, (int)h->C,
                           h->acc(), h->cam_partial.as<double>(), ctrl_done, set, pl, mxb, cx);
    else
        hipLaunchKernelGGL((k_cam_schur<MODE, false>), dim3(grid), dim3(kCamThreads), 0, h->stream, cm,
                           (const double*)h->tab, (const double*)h->rec, h->K, vin, (int)h->C, h->acc(),
                           h->cam_partial.as<double>(), ctrl_done, set, pl, mxb, cx);
    LAUNCHED(h);
    return launch_cam_combine(h, 6, h->acc(), 1, (int)h->C, nullptr, ctrl_done ? &ctrl_done->done : nullptr, tb);
}

// The local form with its tail split off (k_p2p_pcg / k_pcg_tail): sharded solves and cameras of several chunks; on a
// single rank with single-chunk cameras only on request (debug option, tests).
bool pcg_split_mode(const sfmba_handle* h) {
    if (h->dbg.pcg_local == 0 || h->dbg.precond == 0) return false;
    if (!(h->pcg_fused || h->sweep_rc_g)) return false;         // (the forms that have a local prologue / k_pcg_update_local)
    return (multi_rank(h) || h->cam_multi || h->xcd_b) ? h->dbg.pcg_split != 0 : h->dbg.pcg_split == 1;
}

// Reduced right-hand side term -> acc and, with the Schur-diagonal preconditioner, the diagonal blocks of
// W Vinv W^T -> sd in the same pass; all-reduce; block inverses.  (k_prep has written e into the records and, for
// the block-Jacobi-of-U form, Minv itself.)
int launch_rhs_and_preconditioner(sfmba_handle* h) {
    const int64_t C = h->C;
    if (h->dbg.precond == 0) {
        CHK(launch_cam_schur<1>(h, nullptr, nullptr, 0));
        return exchange(h, h->acc(), 6 * C, 0);
    }
    // single-chunk cameras on one rank, or sharded with the sums exchanged camera by camera inside the pass
    // (CamExchange): every workgroup holds its camera's complete sums and inverts its own preconditioner block (RhsPrecond)
    CamExchange cx{};
    if (cam_inline(h)) { cx = cam_exchange(h); ++h->p2p.calls; ++h->n_collectives; }
    if (xcd_cam(h)) {
        const CamMajor cmb{h->cam_chunks_b.as<int4>(), h->cm_pt.as<int>(), h->cm_uv.as<double>()};
        const int wgrid = h->n_chunks_b / kWaveChunkCams;
        const double* rr = h->use_rhsrec ? (const double*)h->rhsrec.as<double>() : (const double*)nullptr;
        if (h->f32 && !h->sweep_rc && !h->sweep_rc_g)
            hipLaunchKernelGGL((k_cam_rhs_diag_w<true>), dim3(wgrid), dim3(kCamThreads), 0, h->stream, cmb, (const double*)h->tab,
                               (const double*)h->rec, (const double*)vinv_ptr(h), h->K, h->cam_partial.as<double>(), rr);
        else
            hipLaunchKernelGGL((k_cam_rhs_diag_w<false>), dim3(wgrid), dim3(kCamThreads), 0, h->stream, cmb, (const double*)h->tab,
                               (const double*)h->rec, (const double*)vinv_ptr(h), h->K, h->cam_partial.as<double>(), rr);
        LAUNCHED(h);
        hipLaunchKernelGGL(k_cam_combine_w<27>, dim3((unsigned)((27 * C + 255) / 256)), dim3(256), 0, h->stream,
                           (const double*)h->cam_partial.as<double>(), (int)C, h->acc(), 1, (int)C, (const int*)nullptr,
                           (const double*)nullptr);
        LAUNCHED(h);
        CHK(exchange(h, h->acc(), 27 * C, 0));                  // acc | sd: one contiguous plane-major vector
        hipLaunchKernelGGL(k_cam_prep_schur, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, h->stream, (const double*)h->Ugc(),
                           (const double*)h->sd(), (int)C, h->Dc.as<double>(), h->Minv.as<double>());
        LAUNCHED(h);
        return 0;
    }
    const bool own_inverse = !h->cam_multi && (!multi_rank(h) || cam_inline(h));
    const RhsPrecond mp = own_inverse ? RhsPrecond{h->Ugc(), h->Dc.as<double>(), h->Minv.as<double>()} : RhsPrecond{nullptr, nullptr, nullptr};
    if (h->f32 && !h->sweep_rc && !h->sweep_rc_g)
        hipLaunchKernelGGL((k_cam_rhs_diag<true>), dim3(h->n_chunks), dim3(kRhsThreads), 0, h->stream, cam_major(h),
                           (const double*)h->tab, (const double*)h->rec, (const double*)vinv_ptr(h), h->K, (int)C,
                           h->acc(), h->cam_partial.as<double>(), mp, h->use_rhsrec ? (const double*)h->rhsrec.as<double>() : (const double*)nullptr, cx);
    else
        hipLaunchKernelGGL((k_cam_rhs_diag<false>), dim3(h->n_chunks), dim3(kRhsThreads), 0, h->stream, cam_major(h),
                           (const double*)h->tab, (const double*)h->rec, (const double*)vinv_ptr(h), h->K, (int)C,
                           h->acc(), h->cam_partial.as<double>(), mp, h->use_rhsrec ? (const double*)h->rhsrec.as<double>() : (const double*)nullptr, cx);
    LAUNCHED(h);
    if (own_inverse) return 0;
    CHK(launch_cam_combine(h, 27, h->acc(), 1, (int)C, nullptr, nullptr));
    CHK(exchange(h, h->acc(), 27 * C, 0));                  // acc | sd: one contiguous plane-major vector
    hipLaunchKernelGGL(k_cam_prep_schur, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, h->stream, (const double*)h->Ugc(),
                       (const double*)h->sd(), (int)C, h->Dc.as<double>(), h->Minv.as<double>());
    LAUNCHED(h);
    return 0;
}

// Few cameras: form the block pairs of W V^-1 W^T, then run the PCG on S = U + Dc - (...) inside one workgroup with
// S in LDS; the camera step and the control block are left where the implicit PCG leaves them (x of vector set 0).
// acc must hold the reduced right-hand-side term (pass B, MODE 1).
constexpr double kDenseTolFactor = 0.1;
// `rhs`: the workgroups of the diagonal pairs also leave the reduced right-hand-side term -sum W e in acc (e in the
// z half of the point records, k_prep); false: acc is the caller's (test entry).
int launch_dense_solve(sfmba_handle* h, double tol, int max_iters, bool rhs) {
    hipLaunchKernelGGL(k_schur_blocks, dim3(h->n_blk), dim3(kCamThreads), 0, h->stream, (const int*)h->cov_ptr.as<int>(),
                       (const int*)h->cov_pt.as<int>(), (const int2*)h->blk_ab.as<int2>(), (const double*)h->tab,
                       (const double*)h->rec, (const double*)vinv_ptr(h), h->K, (int)h->C,
                       h->Sblk.as<double>(), rhs ? h->acc() : (double*)nullptr);
    LAUNCHED(h);
    const int n = 6 * (int)h->C;
    const size_t lds = sizeof(double) * ((size_t)n * (size_t)(n | 1) + 2 * kDenseMaxN + ((21 * kDenseMaxN / 6 + 1) & ~1) +
                                         4 * (kDenseThreads / 64));      // matrix | r u | 6x6 inverses | reduction slots
    CHK(set_lds(h, k_dense_pcg, lds));
    hipLaunchKernelGGL(k_dense_pcg, dim3(1), dim3(kDenseThreads), lds, h->stream, (const double*)h->Sblk.as<double>(),
                       (const double*)h->Ugc(), (const double*)h->Dc.as<double>(),
                       (const double*)h->acc(), (int)h->C, tol, max_iters, h->vecs.as<double>(), h->ctrl.as<PcgCtrl>());
    LAUNCHED(h);
    h->pcg_L = 0;
    return 0;
}

// acc = (S - Dc) v for a plane-major vector v outside the PCG (test and timing entries): pass A, pass B
int schur_product_standalone(sfmba_handle* h, const double* v_planes) {
    const double* va = v_planes;
    if (!h->lds_vec && !h->sweep_rc && !h->sweep_rc_g) {    // pass A gathers v from a camera-major copy in L2
        hipLaunchKernelGGL(k_transpose, dim3((unsigned)((6 * h->C + 255) / 256)), dim3(256), 0, h->stream, v_planes,
                           6, (int)h->C, h->vcm.as<double>(), (const PcgCtrl*)nullptr, 0);
        LAUNCHED(h);
        va = h->vcm.as<double>();
    }
    CHK(launch_point_sweep(h, va, nullptr, 0));
    return launch_cam_schur<0>(h, v_planes, nullptr, 0);
}

// partials -> half B.  When k_update_scale's final sums are still pending they ride along (one extra workgroup).
int launch_jdot(sfmba_handle* h, int* nparts) {
    const int grid = grid_1d(h->N, kSweepThreads, h->n_cu);
    const double* sgc = h->sg.as<double>();
    const double* sgp = sgc + 6 * h->C;
    Piggyback pb{};
    if (h->pending_scale_sums) pb = slices_rider(h, 0, 4);
    const int launch_grid = grid + (pb.part != nullptr ? 1 : 0);
    const Recompute rc{h->tab, h->x + 6 * h->C, h->K};
    if (h->jfree) {
        const size_t lds = sizeof(double) * kCamRow * (size_t)h->C;
        auto kern = k_jdot<false, true>;
        CHK(set_lds(h, kern, lds));
        hipLaunchKernelGGL(kern, dim3(launch_grid), dim3(kSweepThreads), lds, h->stream, obs_arrays(h), sgc, sgp,
                           (int)h->N, (int)h->C, h->t1.as<double>(), h->partB(), pb, rc);
    } else if (h->lds_vec) {
        const size_t lds = sizeof(double) * 6 * h->C;
        auto kern = k_jdot<true>;
        CHK(set_lds(h, kern, lds));
        hipLaunchKernelGGL(kern, dim3(launch_grid), dim3(kSweepThreads), lds, h->stream, obs_arrays(h), sgc, sgp,
                           (int)h->N, (int)h->C, h->t1.as<double>(), h->partB(), pb, rc);
    } else {
        hipLaunchKernelGGL(k_jdot<false>, dim3(launch_grid), dim3(kSweepThreads), 0, h->stream, obs_arrays(h),
                           sgc, sgp, (int)h->N, (int)h->C, h->t1.as<double>(), h->partB(), pb, rc);
    }
    LAUNCHED(h);
    h->pending_scale_sums = false;
    *nparts = grid;
    return 0;
}

int launch_backsub(sfmba_handle* h, int* nparts) {
    const int grid = (h->n_ranges + kWavesPerSweepBlock - 1) / kWavesPerSweepBlock;
    double* dc = h->p.as<double>();
    double* dp = dc + 6 * h->C;
    const PcgCtrl* ctrl2 = h->ctrl.as<PcgCtrl>();
    const Recompute rc{h->tab, h->x + 6 * h->C, h->K};
    if (h->jfree) {
        hipLaunchKernelGGL(k_transpose, dim3((6 * h->C + 255) / 256), dim3(256), 0, h->stream,
                           (const double*)h->vecs.as<double>(), 6, (int)h->C, dc, ctrl2, h->pcg_L);
        LAUNCHED(h);
        const size_t lds = sizeof(double) * kCamRow * (size_t)h->C;
        auto kern = k_backsub<false, true>;
        CHK(set_lds(h, kern, lds));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kSweepThreads), lds, h->stream, h->ranges.as<int2>(), h->n_ranges, obs_arrays(h),
                           h->vecs.as<double>(), dc, vinv_ptr(h), h->gp.as<double>(), h->t1.as<double>(), dp, h->partB(),
                           (int)h->C, (const PcgCtrl*)nullptr, 0, h->g.as<double>(), h->si.as<double>(), h->sg.as<double>(), rc, FinalUpdate{});
    } else if (h->lds_vec) {
        const size_t lds = sizeof(double) * 6 * h->C;
        auto kern = k_backsub<true>;
        CHK(set_lds(h, kern, lds));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kSweepThreads), lds, h->stream, h->ranges.as<int2>(),
                           h->n_ranges, obs_arrays(h), h->vecs.as<double>(), dc, vinv_ptr(h),
                           h->gp.as<double>(), h->t1.as<double>(), dp, h->partB(), (int)h->C,
                           ctrl2, h->pcg_L, h->g.as<double>(), h->si.as<double>(), h->sg.as<double>(), rc,
                           h->pcg_a_owed ? FinalUpdate{h->pcg_part.as<double>(), h->vecs.as<double>(), h->ctrl.as<PcgCtrl>(), h->pcg_L - 1}
                                         : FinalUpdate{});
    } else {
        hipLaunchKernelGGL(k_transpose, dim3((6 * h->C + 255) / 256), dim3(256), 0, h->stream,
                           (const double*)h->vecs.as<double>(), 6, (int)h->C, dc, ctrl2, h->pcg_L);
        LAUNCHED(h);
        hipLaunchKernelGGL(k_backsub<false>, dim3(grid), dim3(kSweepThreads), 0, h->stream,
                           h->ranges.as<int2>(), h->n_ranges, obs_arrays(h), h->vecs.as<double>(), dc,
                           vinv_ptr(h), h->gp.as<double>(), h->t1.as<double>(), dp,
                           h->partB(), (int)h->C, (const PcgCtrl*)nullptr, 0, h->g.as<double>(),
                           h->si.as<double>(), h->sg.as<double>(), rc, FinalUpdate{});
    }
    LAUNCHED(h);
    *nparts = grid;
    return 0;
}

// column scale + gradient + q0..q4 of the new iterate (after the normal blocks are complete)
// `defer`: the final sums are left to ride with the next k_jdot launch (flush_scale_sums if none follows)
int launch_update_scale(sfmba_handle* h, int first, bool defer = false) {
    if (h->scale_pts == 2)
        hipLaunchKernelGGL(k_update_scale<2>, dim3(h->red_grid), dim3(256), 0, h->stream, h->Ugc(), h->V.as<double>(),
                           h->gp.as<double>(), h->x, (int)h->C, (int)h->P, first, h->red_bc, h->si.as<double>(),
                           h->g.as<double>(), h->sg.as<double>(), h->part.as<double>());
    else
        hipLaunchKernelGGL(k_update_scale<1>, dim3(h->red_grid), dim3(256), 0, h->stream, h->Ugc(), h->V.as<double>(),
                           h->gp.as<double>(), h->x, (int)h->C, (int)h->P, first, h->red_bc, h->si.as<double>(),
                           h->g.as<double>(), h->sg.as<double>(), h->part.as<double>());
    LAUNCHED(h);
    if (defer) { h->pending_scale_sums = true; return 0; }
    return launch_finish_slices(h, 0, 4);
}

// q0..q8 with the step vector p
// Final sums of k_backsub's partial rows (half B, kBacksubCols wide): G12, G22 -> slots 2, 3; q5..q8 of the
// point slice -> 4..7 (the run exchange_tail reduces over ranks); q5..q8 of the camera slice -> 21..24.
Piggyback backsub_rider(sfmba_handle* h, int nparts) {
    Piggyback pb{h->partB(), h->scal(), FinishJob{}, 1, kBacksubCols, 0};
    pb.job.row0[0] = 0; pb.job.nrows[0] = nparts;
    for (int k = 0; k < kFinishCols; ++k) { pb.job.slot[0][k] = -1; pb.job.slot[1][k] = -1; }
    for (int k = 0; k < 6; ++k) pb.job.slot[0][k] = 2 + k;
    for (int k = 0; k < 4; ++k) pb.job.slot[0][6 + k] = kCamSlot + 5 + k;
    return pb;
}
int launch_finish_backsub(sfmba_handle* h, int nparts) {
    const Piggyback pb = backsub_rider(h, nparts);
    hipLaunchKernelGGL(k_finish, dim3(1), dim3(64 * kBacksubCols), 0, h->stream, pb.part, pb.job, kBacksubCols, 0,
                       h->scal(), (const double*)nullptr, Mailbox{});
    LAUNCHED(h);
    return 0;
}

// all-reduce freshly written exchange scalars over ranks (no-op on one GPU); stays on the stream.
// Only slots written since their last reduction may be included.
int flush_scale_sums(sfmba_handle* h) {         // no k_jdot follows the last k_update_scale: finish on its own
    if (!h->pending_scale_sums) return 0;
    h->pending_scale_sums = false;
    return launch_finish_slices(h, 0, 4);
}
int exchange_linearise(sfmba_handle* h) {       // q1..q4 and max|g| of the point slice
    CHK(exchange(h, h->scal() + 8, 4, 0));
    CHK(exchange(h, h->scal() + kMaxSlot, 1, 1));
    return 0;
}
int exchange_tail(sfmba_handle* h) {            // G12, G22, q5..q8
    return exchange(h, h->scal() + 2, 6, 0);
}

// bring all 32 scalars to the host (h_scal) and wait: one small kernel posts them into the mailbox (no blit
// copy, no marker packet in the queue, no stream polling)
int fetch_scalars(sfmba_handle* h) {
    const Mailbox mb{h->mbox_dev, h->scal(), nullptr, ++h->mbox_seq, p2p_error_word(h)};
    hipLaunchKernelGGL(k_post, dim3(1), dim3(64), 0, h->stream, mb);
    LAUNCHED(h);
    CHK(wait_mailbox(h, h->mbox_seq));
    memcpy(h->h_scal, h->mbox, sizeof(double) * kScalSlots);
    return 0;
}

// q_k summed over the camera slice and the (rank-reduced) point slice
double qsum(const sfmba_handle* h, int q) { return h->h_scal[kPointSlot[q]] + h->h_scal[kCamSlot + q]; }

// x crosses PCIe through a pinned staging buffer (an async copy from pageable memory is staged by the
// runtime anyway, synchronously and in small pieces)
int ensure_h_x(sfmba_handle* h) {
    const size_t need = (size_t)std::max<int64_t>(h->n, 2 * h->N);      // also stages the residual vector
    if (h->h_x && h->h_x_doubles >= need) return 0;
    if (h->h_x) { (void)hipHostFree(h->h_x); h->h_x = nullptr; h->h_x_doubles = 0; }
    HIPCHK(h, hipHostMalloc((void**)&h->h_x, sizeof(double) * need, hipHostMallocDefault));
    h->h_x_doubles = need;
    return 0;
}

// pageable <-> pinned copy of a parameter vector: a few host threads from 1 MB on (one thread moves 2.4 MB in 85-115 us,
// which was 9 % of a solve at 306k parameters).  `chunk_done(k, offset, bytes)` (optional) is called on the calling
// thread, in chunk order, as soon as chunk k has arrived: the upload enqueues that chunk's DMA while the others are
// still being copied.
template <class Fn>
void staging_copy(sfmba_handle* h, void* dst, const void* src, size_t bytes, Fn chunk_done) {
    const int parts = (int)std::min<size_t>(4, bytes >> 19);
    if (parts <= 1) { memcpy(dst, src, bytes); chunk_done(0, (size_t)0, bytes); return; }
    const size_t per = ((bytes + parts - 1) / parts + 63) & ~(size_t)63;
    std::atomic<int> done[8];
    for (auto& d : done) d.store(0, std::memory_order_relaxed);
    h->pool.run(parts, [&](int t) {
        const size_t b = std::min(bytes, (size_t)t * per), e = std::min(bytes, b + per);
        if (e > b) memcpy(static_cast<char*>(dst) + b, static_cast<const char*>(src) + b, e - b);
        if (t != 0) { done[t].store(1, std::memory_order_release); return; }
        chunk_done(0, b, e - b);
        for (int k = 1; k < parts; ++k) {
            while (done[k].load(std::memory_order_acquire) == 0) { }
            const size_t bk = std::min(bytes, (size_t)k * per), ek = std::min(bytes, bk + per);
            chunk_done(k, bk, ek - bk);
        }
    });
}
void staging_copy(sfmba_handle* h, void* dst, const void* src, size_t bytes) {
    staging_copy(h, dst, src, bytes, [](int, size_t, size_t) {});
}

int upload_x(sfmba_handle* h, const double* x_host) {
    CHK(ensure_h_x(h));
    if (h->mixed) {                    // origin of the fp32 operands: mean of (a sample of) this vector's points
        const double* pts = x_host + 6 * h->C;
        const int64_t stride = std::max<int64_t>(1, h->P / 1024);
        double sx = 0.0, sy = 0.0, sz = 0.0;
        int64_t cnt = 0;
        for (int64_t q = 0; q < h->P; q += stride, ++cnt) { sx += pts[3 * q]; sy += pts[3 * q + 1]; sz += pts[3 * q + 2]; }
        h->origin = Origin{sx / (double)cnt, sy / (double)cnt, sz / (double)cnt};
        if (!std::isfinite(h->origin.x + h->origin.y + h->origin.z)) h->origin = Origin{0.0, 0.0, 0.0};
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));      // the staging buffer may still be in flight
    const double t0 = now_s();
    hipError_t err = hipSuccess;
    staging_copy(h, h->h_x, x_host, sizeof(double) * h->n, [&](int, size_t off, size_t bytes) {
        if (bytes == 0 || err != hipSuccess) return;
        err = hipMemcpyAsync(reinterpret_cast<char*>(h->x) + off, reinterpret_cast<const char*>(h->h_x) + off, bytes,
                             hipMemcpyHostToDevice, h->stream);
    });
    HIPCHK(h, err);
    if (h->dbg.trace_timing) fprintf(stderr, "sfmba: upload_x  staging copy + enqueue %.1f us\n", 1e6 * (now_s() - t0));
    return 0;
}

// residual vector of the current buffer set into the caller's array (2N doubles, caller's observation
// order), through the pinned staging buffer
int download_residuals(sfmba_handle* h, double* r_out) {
    CHK(ensure_h_x(h));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const size_t bytes = (h->f32 ? sizeof(float) : sizeof(double)) * 2 * (size_t)h->N;
    HIPCHK(h, hipMemcpyAsync(h->h_x, h->r.p, bytes, hipMemcpyDeviceToHost, h->stream));
    CHK(wait_stream(h));
    if (!h->f32 && !h->permuted) { memcpy(r_out, h->h_x, bytes); return 0; }
    const float* sf = reinterpret_cast<const float*>(h->h_x);
    for (int64_t k = 0; k < h->N; ++k) {
        const int64_t d = h->permuted ? h->order[k] : k;
        r_out[2 * d] = h->f32 ? (double)sf[2 * k] : h->h_x[2 * k];
        r_out[2 * d + 1] = h->f32 ? (double)sf[2 * k + 1] : h->h_x[2 * k + 1];
    }
    return 0;
}

int check_ready(sfmba_handle* h, const void* x) {
    if (!h->have_problem) return fail(h, -1, "sfmba_set_problem has not been called");
    if (h->transport_dropped)
        return fail(h, -1, "sfmba_set_problem removed this handle's multi-rank transport (direct link / RCCL communicator / "
                           "callback): set it up again for the new problem, or call sfmba_set_exchange(h, NULL, 0, NULL, NULL, 0) "
                           "to solve the shard on its own");
    if (!x) return fail(h, -1, "x is NULL");
    return 0;
}

int pcg_max_iters(const sfmba_handle* h, const sfmba_options& opt) {
    return opt.pcg_max_iter > 0 ? opt.pcg_max_iter : (int)std::max<int64_t>(20, 2 * 6 * h->C);
}

// x = 0, r = rhs, u = Minv r (acc holds the reduced right-hand side term of pass B, MODE 1)
int pcg_start(sfmba_handle* h, const sfmba_options& opt) {
    h->pcg_L = 0;
    h->pcg_b_owed = false;
    h->pcg_a_owed = false;
    // sharded over the direct link with single-chunk cameras: the local form itself, the all-reduce of the product inside
    // pass B (CamExchange); else the local form with its tail behind the reduction (pcg_split), else the general forms
    h->pcg_inline = p2p_inline_ok(h) && !h->cam_multi && !h->xcd_b && h->dbg.pcg_inline != 0 && h->dbg.pcg_local != 0 &&
                    h->dbg.precond != 0 && h->dbg.pcg_split != 1 && (h->pcg_fused || h->sweep_rc_g);
    h->pcg_split = !h->pcg_inline && pcg_split_mode(h);
    const bool own_cameras = (!multi_rank(h) && !h->cam_multi && !h->xcd_b) || h->pcg_split || h->pcg_inline;
    h->pcg_local = h->pcg_fused && h->dbg.pcg_local != 0 && own_cameras;
    h->pcg_local2 = !h->pcg_fused && h->sweep_rc_g && h->dbg.pcg_local != 0 && own_cameras;
    if (h->pcg_fused) {                   // launch 0 of the fused form initialises the solve itself
        h->pcg_tol = opt.pcg_tol;
        h->pcg_cap = pcg_max_iters(h, opt);
        return 0;
    }
    if (h->pcg_local2)                    // its update takes gamma from the iterate: no reduction at the start
        hipLaunchKernelGGL(k_pcg_init_local, dim3((unsigned)((h->C + 63) / 64)), dim3(64), 0, h->stream, (const double*)h->Ugc(),
                           (const double*)h->acc(), (const double*)h->Minv.as<double>(), (int)h->C, h->vecs.as<double>(),
                           (const double*)(h->scal() + kEtaSlot), pcg_max_iters(h, opt), h->ctrl.as<PcgCtrl>());
    else
        hipLaunchKernelGGL(k_pcg_init, dim3(1), dim3(1024), 0, h->stream, h->Ugc(), (const double*)h->acc(),
                           h->Minv.as<double>(), (int)h->C, h->vecs.as<double>(), (const double*)(h->scal() + kEtaSlot),
                           pcg_max_iters(h, opt), h->ctrl.as<PcgCtrl>());
    LAUNCHED(h);
    return 0;
}

// The product of pass B (camera chunks already combined) summed over the ranks, and the per-camera tail of the local form
// behind it: one launch on the direct path (k_p2p_pcg), else the collective of the transport and k_pcg_tail.
int launch_reduce_and_tail(sfmba_handle* h, const PcgCtrl* cd, int set) {
    const int C = (int)h->C;
    const PcgLocal pl{h->Dc.as<double>(), h->Minv.as<double>(), h->vecs.as<double>(), h->pcg_part.as<double>()};
    const int grid = std::max((C + kP2pPcgThreads - 1) / kP2pPcgThreads, std::min(kP2pMaxBlocks, (C + 63) / 64));
    if (h->p2p.ready && 6 * h->C <= h->p2p.stride && grid <= kP2pMaxBlocks) {
        P2pArgs a{};
        p2p_fill_args(h, a);
        hipLaunchKernelGGL(k_p2p_pcg, dim3(grid), dim3(kP2pPcgThreads), 0, h->stream, h->acc(), (const double*)h->vecs.as<double>(),
                           C, cd, set, pl, a);
        LAUNCHED(h);
        ++h->p2p.calls;
        ++h->n_collectives;
        return 0;
    }
    CHK(exchange(h, h->acc(), 6 * h->C, 0, &cd->done));
    hipLaunchKernelGGL(k_pcg_tail, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, h->stream, (const double*)h->acc(),
                       (const double*)h->vecs.as<double>(), C, cd, set, pl);
    LAUNCHED(h);
    return 0;
}

// enqueue `count` PCG iterations (pass A [+ update], pass B, all-reduce [, update]); iterations after
// convergence are device-side no-ops, so over-enqueueing is harmless and deterministic.
// `speculative` (fused form, one rank): the batch is the whole guess of an outer iteration and k_backsub follows it.  In
// the fused form k iterations need k + 1 launches of pass A -- the last one only applies the update of iteration k - 1 and
// finds the solve finished -- and the pass B behind that launch returns at its first instruction.  That last pair is not
// enqueued: in the local form k_backsub's prologue does the update itself (FinalUpdate: alpha, x += alpha p, the control
// block -- 8 us of launch less per outer iteration), otherwise pass A is launched and only its pass B is left out
// (4.5 us).  Either way the control block tells whether the guess sufficed; if it did not, the pair is OWED: the next
// call here starts with it (pass A / pass B of launch L need nothing but the vector sets, the partial sums and, for pass
// B, the z of pass A of launch L, all untouched since).
int pcg_pass_b(sfmba_handle* h, int L) {
    const PcgCtrl* cd = h->ctrl.as<PcgCtrl>() + ((L + 1) & 1);
    CHK(launch_cam_schur<0>(h, h->vecs.as<double>(), cd, L & 1, h->pcg_local && !h->pcg_split, h->pcg_inline));
    if (h->pcg_split) CHK(launch_reduce_and_tail(h, cd, L & 1));
    else if (!h->pcg_inline) CHK(exchange(h, h->acc(), 6 * h->C, 0, &cd->done));
    return 0;
}

int pcg_enqueue(sfmba_handle* h, int count, bool speculative = false) {
    PcgCtrl* ctrl2 = h->ctrl.as<PcgCtrl>();
    if (h->pcg_a_owed) {                                        // (k_backsub stood in for this launch and found work left)
        h->pcg_a_owed = false;
        h->pcg_b_owed = true;
        CHK(launch_pcg_fused(h, h->pcg_L - 1));
    }
    if (h->pcg_b_owed) {
        h->pcg_b_owed = false;
        CHK(pcg_pass_b(h, h->pcg_L - 1));
    }
    for (int k = 0; k < count; ++k) {
        const int L = h->pcg_L;
        if (h->pcg_fused) {
            // (several ranks: only where the product's exchange sits inside pass B -- every rank then takes the same
            // decision from the same record, and a launch that is not enqueued exchanges nothing on any of them)
            const bool last = speculative && k == count - 1 && (!multi_rank(h) || h->pcg_inline) && h->dbg.pcg_skip_last != 0;
            // ... and in the local form with the step vector in k_backsub's LDS not even that pass A: k_backsub's prologue
            // does its update (FinalUpdate)
            if (last && L > 0 && h->pcg_local && h->lds_vec && !h->jfree && h->dbg.pcg_skip_last != 2) {
                h->pcg_L = L + 1;
                h->pcg_a_owed = true;
                break;
            }
            CHK(launch_pcg_fused(h, L));
            // a launch that found the solve finished (or finished it) produced no z: its control block (written
            // to slot (L+1)&1) says so, and pass B and the collective behind it are void as well
            h->pcg_L = L + 1;
            if (last) { h->pcg_b_owed = true; break; }
            CHK(pcg_pass_b(h, L));
            continue;
        }
        const PcgCtrl* cd = ctrl2 + (L & 1);                     // current until k_pcg_update writes the other one
        CHK(launch_point_sweep(h, h->vecs.as<double>(), ctrl2, L));
        CHK(launch_cam_schur<0>(h, h->vecs.as<double>(), cd, -1, h->pcg_local2 && !h->pcg_split, h->pcg_inline));
        if (h->pcg_local2 && h->pcg_split) CHK(launch_reduce_and_tail(h, cd, -1));
        if (h->pcg_local2) {                                    // pass B / the tail did the bookkeeping: the light update
            hipLaunchKernelGGL(k_pcg_update_local, dim3((unsigned)((h->C + 1023) / 1024)), dim3(1024), 0, h->stream,
                               h->vecs.as<double>(), (const double*)h->pcg_part.as<double>(), ctrl2, L, (int)h->C);
            LAUNCHED(h);
            h->pcg_L = L + 1;
            continue;
        }
        CHK(exchange(h, h->acc(), 6 * h->C, 0, &cd->done));
        hipLaunchKernelGGL(k_pcg_update, dim3(kPcgUpdateBlocks), dim3(1024), 0, h->stream, (const double*)h->acc(),
                           h->Dc.as<double>(), h->Minv.as<double>(), (int)h->C, h->vecs.as<double>(), ctrl2, L);
        LAUNCHED(h);
        h->pcg_L = L + 1;
    }
    return 0;
}

int pcg_read(sfmba_handle* h, PcgCtrl* hc) {
    static_assert(sizeof(PcgCtrl) <= 24 * sizeof(double), "PcgCtrl fits the pinned tail");
    HIPCHK(h, hipMemcpyAsync(h->h_scal + 40, h->ctrl.as<PcgCtrl>() + (h->pcg_L & 1), sizeof *hc,
                             hipMemcpyDeviceToHost, h->stream));
    CHK(wait_stream(h));
    memcpy(hc, h->h_scal + 40, sizeof *hc);
    return 0;
}

// poll until the device reports the PCG finished
int pcg_finish_polling(sfmba_handle* h, const sfmba_options& opt, PcgCtrl* hc) {
    const int every = std::max(1, opt.pcg_check_every);
    const int cap = pcg_max_iters(h, opt) + every + 1;
    int launched = 0;
    for (;;) {
        CHK(pcg_enqueue(h, every));
        launched += every;
        CHK(pcg_read(h, hc));
        if (hc->done != 0 || launched > cap) return 0;
    }
}

void append_center(std::string& line, const char* s) {          // Python's format spec ^15
    const int w = 15, len = (int)strlen(s);
    const int left = (w - len) / 2 > 0 ? (w - len) / 2 : 0;
    const int right = w - len - left > 0 ? w - len - left : 0;
    line.append((size_t)left, ' ').append(s).append((size_t)right, ' ');
}
void emit_line(sfmba_handle* h, const std::string& line) {
    if (h->print_fn) { h->print_fn(h->print_ctx, line.c_str()); return; }
    fputs(line.c_str(), stdout);
    fputc('\n', stdout);
    fflush(stdout);
}

// scipy's iteration table (SCIPY common.py:545-563: print_header_nonlinear / print_iteration_nonlinear)
void print_header(sfmba_handle* h) {
    const char* cols[6] = {"Iteration", "Total nfev", "Cost", "Cost reduction", "Step norm", "Optimality"};
    std::string line;
    for (auto c : cols) append_center(line, c);
    emit_line(h, line);
}
void print_iter(sfmba_handle* h, int64_t it, int64_t nfev, double cost, bool have, double red, double step, double opt) {
    char b[64];
    std::string line;
    snprintf(b, sizeof b, "%lld", (long long)it); append_center(line, b);
    snprintf(b, sizeof b, "%lld", (long long)nfev); append_center(line, b);
    snprintf(b, sizeof b, "%.4e", cost); append_center(line, b);
    if (have) { snprintf(b, sizeof b, "%.2e", red); append_center(line, b); snprintf(b, sizeof b, "%.2e", step); append_center(line, b); }
    else { append_center(line, ""); append_center(line, ""); }
    snprintf(b, sizeof b, "%.2e", opt); append_center(line, b);
    emit_line(h, line);
}

}  // namespace

// ==================================================================================================
// C-ABI
// ==================================================================================================

extern "C" {

void sfmba_default_options(sfmba_options* o) {
    if (!o) return;
    memset(o, 0, sizeof *o);
    o->ftol = 1e-8; o->xtol = 1e-8; o->gtol = 1e-8;
    o->max_nfev = 0; o->verbose = 0; o->max_iter = 0;
    o->pcg_tol = 1e-2; o->pcg_max_iter = 0; o->pcg_check_every = 2;
    o->reg_min = 1e-6; o->profile = 0; o->reserved = 0; o->pcg_tol_max = 0.1;
}

int sfmba_create(sfmba_handle** out, int device_id) {
    if (!out) return -1;
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return -3;      // no HIP device: the product path fails loudly
    if (device_id < 0 || device_id >= ndev) return -1;
    auto* h = new sfmba_handle();
    h->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess) { delete h; return -3; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) h->n_cu = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { delete h; return -3; }
    h->own_stream = true;
    if (hipHostMalloc((void**)&h->h_scal, sizeof(double) * 64, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&h->mbox, sizeof(double) * 64, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void**)&h->mbox_dev, h->mbox, 0) != hipSuccess) {
        (void)hipStreamDestroy(h->stream); delete h; return -4;
    }
    memset(h->mbox, 0, sizeof(double) * 64);
    {   // lowest priority: where the copy runs as a kernel its waves yield to the evaluation it overlaps
        int prio_low = 0, prio_high = 0;
        if (hipDeviceGetStreamPriorityRange(&prio_low, &prio_high) != hipSuccess) prio_low = 0;
        if (hipStreamCreateWithPriority(&h->copy_stream, hipStreamNonBlocking, prio_low) != hipSuccess) h->copy_stream = nullptr;
    }
    if (h->copy_stream &&
        (hipEventCreateWithFlags(&h->ev_written, hipEventDisableTiming) != hipSuccess ||
         hipEventCreateWithFlags(&h->ev_copied[0], hipEventDisableTiming) != hipSuccess ||
         hipEventCreateWithFlags(&h->ev_copied[1], hipEventDisableTiming) != hipSuccess)) {
        (void)hipStreamDestroy(h->copy_stream);            // (the solve then returns x the plain way)
        h->copy_stream = nullptr;
    }
    *out = h;
    return 0;
}

void sfmba_destroy(sfmba_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->comm) { if (RcclApi* api = rccl_api()) (void)api->CommDestroy(h->comm); h->comm = nullptr; }
    p2p_release(h);
    if (h->copy_stream) { (void)hipStreamSynchronize(h->copy_stream); (void)hipStreamDestroy(h->copy_stream); }
    if (h->ev_written) (void)hipEventDestroy(h->ev_written);
    for (auto& ev : h->ev_copied) if (ev) (void)hipEventDestroy(ev);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    if (h->h_scal) (void)hipHostFree(h->h_scal);
    if (h->mbox) (void)hipHostFree(h->mbox);
    if (h->h_x) (void)hipHostFree(h->h_x);
    delete h;
}

const char* sfmba_last_error(const sfmba_handle* h) { return h ? h->err.c_str() : "null handle"; }

int sfmba_set_stream(sfmba_handle* h, void* hip_stream) {
    CHK(enter(h));
    if (h->stream) HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    h->stream = static_cast<hipStream_t>(hip_stream);
    h->own_stream = false;
    return 0;
}

int sfmba_set_print(sfmba_handle* h, sfmba_print_fn fn, void* ctx) {
    CHK(enter(h));
    h->print_fn = fn;
    h->print_ctx = ctx;
    return 0;
}

int sfmba_debug_option(sfmba_handle* h, const char* name, int64_t value) {
    CHK(enter(h));
    if (!name) return fail(h, -1, "option name is NULL");
    const std::string n(name);
    const int v = (int)value;
    if (n == "pcg_fused") h->dbg.pcg_fused = v;
    else if (n == "sweep_rc") h->dbg.sweep_rc = v;
    else if (n == "dense") h->dbg.dense = v;
    else if (n == "precond") h->dbg.precond = v;
    else if (n == "pcg_local") h->dbg.pcg_local = v;
    else if (n == "pcg_split") h->dbg.pcg_split = v;
    else if (n == "pcg_mixed") h->dbg.pcg_mixed = v;
    else if (n == "pcg_inline") h->dbg.pcg_inline = v;
    else if (n == "pcg_skip_last") h->dbg.pcg_skip_last = v;
    else if (n == "xcd_cam") h->dbg.xcd_cam = v;
    else if (n == "cm_device") h->dbg.cm_device = v;
    else if (n == "jfree") h->dbg.jfree = v;
    else if (n == "packed_upload") h->dbg.packed_upload = v;
    else if (n == "pcg_mixed_b") h->dbg.pcg_mixed_b = v;
    else if (n == "cost_rider") h->dbg.cost_rider = v;
    else if (n == "rhsrec") h->dbg.rhsrec = v;
    else if (n == "xcd_chunks") h->dbg.xcd_chunks = v;
    else if (n == "tab_lds") h->dbg.tab_lds = v;
    else if (n == "vec_lds") h->dbg.vec_lds = v;
    else if (n == "cam_chunk") h->dbg.cam_chunk = v;
    else if (n == "pcg_guess_bias") h->dbg.pcg_guess_bias = v;
    else if (n == "trace_pcg") h->dbg.trace_pcg = v;
    else if (n == "trace_stalls") h->dbg.trace_stalls = v;
    else if (n == "trace_timing") h->dbg.trace_timing = v;
    else if (n == "wait_deadline_s") h->dbg.wait_deadline_s = v;
    else if (n == "p2p_delay_ms") h->dbg.p2p_delay_ms = v;
    else if (n == "p2p_timeout_ms") h->dbg.p2p_timeout_ms = v;
    else return fail(h, -1, "unknown debug option '%s'", name);
    return 0;
}

int sfmba_set_precision(sfmba_handle* h, int32_t storage_bits) {
    CHK(enter(h));
    if (storage_bits != 64 && storage_bits != 32) return fail(h, -1, "storage_bits must be 64 or 32");
    h->f32_next = storage_bits == 32;
    return 0;
}

int sfmba_set_fixed_cameras(sfmba_handle* h, const int64_t* camera_indices, int64_t n_fixed) {
    CHK(enter(h));
    if (n_fixed < 0 || (n_fixed > 0 && !camera_indices)) return fail(h, -1, "bad fixed-camera list");
    h->fixed_next.assign(camera_indices, camera_indices + n_fixed);
    return 0;
}

int64_t sfmba_exchange_doubles(int64_t n_cameras) { return 54 * n_cameras + kScalSlots; }   // kScalSlots = 32

int sfmba_set_exchange(sfmba_handle* h, void* arena, int64_t arena_doubles, sfmba_allreduce_fn fn,
                       void* ctx, int64_t n_obs_total) {
    CHK(enter(h));
    if (!h->have_problem) return fail(h, -1, "call sfmba_set_problem before sfmba_set_exchange");
    h->transport_dropped = false;
    if (!fn) {
        h->ar_fn = nullptr; h->ar_ctx = nullptr;
        h->arena = h->arena_own.as<double>();
        h->N_total = h->N;
        return 0;
    }
    if (!arena || arena_doubles < sfmba_exchange_doubles(h->C))
        return fail(h, -1, "exchange arena too small: need %lld doubles", (long long)sfmba_exchange_doubles(h->C));
    if (n_obs_total < h->N) return fail(h, -1, "n_obs_total is smaller than the local shard");
    h->arena = static_cast<double*>(arena);
    h->arena_doubles = arena_doubles;
    h->ar_fn = fn; h->ar_ctx = ctx;
    h->N_total = n_obs_total;
    return 0;
}

int sfmba_comm_get_unique_id(void* id128_out) {
    RcclApi* api = rccl_api();
    if (!api || !id128_out) return -5;
    ncclUniqueId id;
    if (api->GetUniqueId(&id) != ncclSuccess) return -5;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    memcpy(id128_out, &id, sizeof id);
    return 0;
}

int sfmba_comm_init(sfmba_handle* h, const void* id128, int32_t rank, int32_t world, int64_t n_obs_total) {
    CHK(enter(h));
    if (!h->have_problem) return fail(h, -1, "call sfmba_set_problem before sfmba_comm_init");
    if (!id128 || world < 1 || rank < 0 || rank >= world) return fail(h, -1, "bad communicator arguments");
    if (n_obs_total < h->N) return fail(h, -1, "n_obs_total is smaller than the local shard");
    RcclApi* api = rccl_api();
    if (!api) return fail(h, -5, "librccl.so.1 could not be loaded: %s", dlerror());
    if (h->comm) { (void)api->CommDestroy(h->comm); h->comm = nullptr; }
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    const ncclResult_t rc = api->CommInitRank(&h->comm, world, id, rank);
    if (rc != ncclSuccess) { h->comm = nullptr; return fail(h, -5, "ncclCommInitRank failed: %s", api->GetErrorString(rc)); }
    h->ar_fn = nullptr; h->ar_ctx = nullptr;
    h->arena = h->arena_own.as<double>();
    h->N_total = n_obs_total;
    h->transport_dropped = false;
    return 0;
}

int sfmba_comm_destroy(sfmba_handle* h) {
    CHK(enter(h));
    h->transport_dropped = false;
    if (h->comm) {
        if (h->stream) HIPCHK(h, hipStreamSynchronize(h->stream));
        if (RcclApi* api = rccl_api()) (void)api->CommDestroy(h->comm);
        h->comm = nullptr;
    }
    h->N_total = h->N;
    return 0;
}

// ---- direct all-reduce over peer-mapped memory ---------------------------------------------------------
namespace {
// unmap the peers; this rank's own buffer stays allocated (peers may still be storing into it)
void p2p_close_peers(sfmba_handle* h) {
    auto& p = h->p2p;
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (int q = 0; q < kP2pMaxRanks; ++q) {
        if (p.opened[q]) (void)hipIpcCloseMemHandle(p.opened[q]);
        p.opened[q] = nullptr; p.data[q] = nullptr; p.flags[q] = nullptr; p.camdata[q] = nullptr;
    }
    p.ready = false;
}
void p2p_release(sfmba_handle* h) {
    auto& p = h->p2p;
    p2p_close_peers(h);
    if (p.own) (void)hipFree(p.own);
    if (p.words) (void)hipFree(p.words);
    p.own = nullptr; p.words = nullptr; p.ready = false; p.world = 0; p.stride = 0;
    p.shared_device = 1; p.any_multi = false;
}
}  // namespace

int sfmba_p2p_export(sfmba_handle* h, int32_t world, void* handle64_out) {
    CHK(enter(h));
    if (!h->have_problem) return fail(h, -1, "call sfmba_set_problem before sfmba_p2p_export");
    if (!handle64_out || world < 1 || world > kP2pMaxRanks) return fail(h, -1, "bad arguments (1 <= world <= %d)", kP2pMaxRanks);
    p2p_release(h);
    auto& p = h->p2p;
    p.world = world;
    p.stride = (27 * h->C + 15) / 16 * 16;                // the largest vector exchanged: [U | g_c]
    const size_t bytes = kP2pFlagBytes + sizeof(double) * 2 * (size_t)world * (size_t)p.stride +
                         sizeof(double) * cam_slots_total(world, (int)h->C);          // + the per-camera slots (CamExchange)
    // uncached device memory: remote stores land in HBM and local loads do not see stale cache lines
    if (hipExtMallocWithFlags(&p.own, bytes, hipDeviceMallocUncached) != hipSuccess) {
        p.own = nullptr; (void)hipGetLastError();
        return fail(h, -5, "hipExtMallocWithFlags(uncached, %zu bytes) failed", bytes);
    }
    HIPCHK(h, hipMalloc((void**)&p.words, 4 * sizeof(unsigned)));
    HIPCHK(h, hipMemset(p.own, 0, bytes));
    {   // the per-camera slots start EMPTY (CamExchange)
        const size_t cam_off = kP2pFlagBytes + sizeof(double) * 2 * (size_t)world * (size_t)p.stride;
        HIPCHK(h, hipMemsetD32((hipDeviceptr_t)(static_cast<char*>(p.own) + cam_off), (int)0xFFF8A5A5u, (bytes - cam_off) / 4));
    }
    HIPCHK(h, hipMemset(p.words, 0, 4 * sizeof(unsigned)));
    HIPCHK(h, hipDeviceSynchronize());
    hipIpcMemHandle_t mh;
    if (hipIpcGetMemHandle(&mh, p.own) != hipSuccess) {
        (void)hipGetLastError(); p2p_release(h);
        return fail(h, -5, "hipIpcGetMemHandle failed");
    }
    static_assert(sizeof mh == 64, "hipIpcMemHandle_t is 64 bytes");
    memcpy(handle64_out, &mh, sizeof mh);
    return 0;
}

int sfmba_p2p_attach(sfmba_handle* h, const void* handles, int32_t rank, int32_t world) {
    CHK(enter(h));
    auto& p = h->p2p;
    if (!p.own || world != p.world) return fail(h, -1, "sfmba_p2p_export(world) must precede sfmba_p2p_attach");
    if (!handles || rank < 0 || rank >= world) return fail(h, -1, "bad arguments");
    p.rank = rank;
    for (int q = 0; q < world; ++q) {
        void* base = p.own;
        if (q != rank) {
            hipIpcMemHandle_t mh;
            memcpy(&mh, static_cast<const char*>(handles) + 64 * (size_t)q, sizeof mh);
            if (hipIpcOpenMemHandle(&base, mh, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
                (void)hipGetLastError(); p2p_close_peers(h);
                return fail(h, -5, "hipIpcOpenMemHandle failed for rank %d", q);
            }
            p.opened[q] = base;
        }
        p.flags[q] = static_cast<unsigned long long*>(base);
        p.data[q] = reinterpret_cast<double*>(static_cast<char*>(base) + kP2pFlagBytes);
        p.camdata[q] = p.data[q] + 2 * (size_t)world * (size_t)p.stride;
    }
    // Self-test: 24 rounds over the three message sizes the solver uses (a few scalars, 6 C, 27 C), each a
    // chain of one to three back-to-back collectives on the same vector (parity and sequence protocol without
    // the host in between).  Rank r contributes (r + 1 + round)(i mod 97 + 1): every sum is exact in fp64.
    const int64_t sizes[3] = {7, 6 * h->C, std::min<int64_t>(27 * h->C, p.stride)};
    std::vector<double> v((size_t)sizes[2]);
    double* dev = h->arena;                                    // 45 C + 32 doubles: room for 27 C
    bool ok = true;
    for (int round = 0; round < 24 && ok; ++round) {
        const int nt = (int)sizes[round % 3], chain = 1 + round % 3;
        for (int i = 0; i < nt; ++i) v[i] = (double)(rank + 1 + round) * (double)(i % 97 + 1);
        HIPCHK(h, hipMemcpyAsync(dev, v.data(), sizeof(double) * nt, hipMemcpyHostToDevice, h->stream));
        for (int k = 0; k < chain; ++k) CHK(p2p_allreduce(h, dev, nt, 0, nullptr));
        HIPCHK(h, hipMemcpyAsync(v.data(), dev, sizeof(double) * nt, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        double wsum = 0.5 * world * (world + 1) + (double)round * world;
        for (int k = 1; k < chain; ++k) wsum *= world;
        for (int i = 0; i < nt; ++i) ok = ok && v[i] == wsum * (double)(i % 97 + 1);
    }
    // What the ranks have to agree on before the per-camera exchanges may run inside the producing kernels (cam_inline):
    // slot q = the identity of rank q's GPU (ranks sharing one: a rehearsal), slot `world` = how many ranks hold a shard
    // whose cameras need combine launches.  One more sum over the link; exact (40-bit integers).
    {
        char bus[64] = {0};
        (void)hipDeviceGetPCIBusId(bus, (int)sizeof bus - 1, h->device);
        unsigned long long id = 1469598103934665603ull;
        for (const char* c = bus; *c; ++c) id = (id ^ (unsigned char)*c) * 1099511628211ull;
        std::fill(v.begin(), v.begin() + world + 1, 0.0);
        v[(size_t)rank] = (double)(1 + (id >> 24));
        v[(size_t)world] = (h->cam_multi || h->xcd_b) ? 1.0 : 0.0;
        HIPCHK(h, hipMemcpyAsync(dev, v.data(), sizeof(double) * (world + 1), hipMemcpyHostToDevice, h->stream));
        CHK(p2p_allreduce(h, dev, world + 1, 0, nullptr));
        HIPCHK(h, hipMemcpyAsync(v.data(), dev, sizeof(double) * (world + 1), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        p.shared_device = 0;
        for (int q = 0; q < world; ++q) p.shared_device += v[(size_t)q] == v[(size_t)rank] ? 1 : 0;
        p.any_multi = v[(size_t)world] != 0.0;
    }
    const int nt = (int)sizes[2];
    unsigned words[2] = {0, 0};
    HIPCHK(h, hipMemcpy(words, p.words, sizeof words, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemsetAsync(dev, 0, sizeof(double) * nt, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (!ok || words[1] != 0) {
        p2p_close_peers(h);                                // own buffer is freed by sfmba_p2p_detach, once all ranks agree
        return fail(h, -5, "direct all-reduce self-test failed (%s)", words[1] ? "timeout waiting for a peer" : "wrong sum");
    }
    p.ready = true;
    h->transport_dropped = false;
    return 0;
}

int sfmba_p2p_detach(sfmba_handle* h) {
    CHK(enter(h));
    h->transport_dropped = false;
    p2p_release(h);
    return 0;
}

int64_t sfmba_p2p_calls(const sfmba_handle* h) { return h ? h->p2p.calls : 0; }

int sfmba_problem_reuse(const sfmba_handle* h, int64_t* obs_reused, int64_t* obs_uploaded) {
    if (!h) return -1;
    if (obs_reused) *obs_reused = h->obs_reused;
    if (obs_uploaded) *obs_uploaded = h->obs_uploaded;
    return 0;
}

int32_t sfmba_get_pcg_history(const sfmba_handle* h, int32_t* out, int32_t cap) {
    if (!h) return -1;
    const int32_t n = (int32_t)h->pcg_hist.size();
    for (int32_t k = 0; k < n && k < cap && out; ++k) out[k] = h->pcg_hist[(size_t)k];
    return n;
}

int sfmba_get_counters(const sfmba_handle* h, int64_t* kernel_launches, int64_t* collectives) {
    if (!h) return -1;
    if (kernel_launches) *kernel_launches = h->n_launches;
    if (collectives) *collectives = h->n_collectives;
    return 0;
}

static int set_problem_impl(sfmba_handle* h, int64_t C, int64_t P, int64_t N, const int64_t* cam, const int64_t* pt,
                            const double* uv, const int64_t* uv_i64, const double* K);

int sfmba_set_problem(sfmba_handle* h, int64_t C, int64_t P, int64_t N, const int64_t* cam, const int64_t* pt,
                      const double* uv, const double* K) {
    return set_problem_impl(h, C, P, N, cam, pt, uv, nullptr, K);
}

int sfmba_set_problem_i64(sfmba_handle* h, int64_t C, int64_t P, int64_t N, const int64_t* cam, const int64_t* pt,
                          const int64_t* uv, const double* K) {
    return set_problem_impl(h, C, P, N, cam, pt, nullptr, uv, K);
}

static int set_problem_impl(sfmba_handle* h, int64_t C, int64_t P, int64_t N, const int64_t* cam, const int64_t* pt,
                            const double* uv, const int64_t* uv_i64, const double* K) {
    CHK(enter(h));
    const bool timing = h->dbg.trace_timing != 0;
    const double tp0 = now_s();
    double tp1 = tp0, tp2 = tp0, tp3 = tp0, ts1 = tp0, ts2 = tp0, ts3 = tp0;
    h->have_problem = false;
    h->solved = false;
    if (C <= 0 || P <= 0 || N <= 0) return fail(h, -1, "n_cameras, n_points, n_obs must be positive");
    if (!cam || !pt || (!uv && !uv_i64) || !K) return fail(h, -1, "NULL array argument");
    if (N >= (int64_t)1 << 30 || 6 * C + 3 * P >= (int64_t)1 << 31)
        return fail(h, -1, "problem too large for 32-bit observation indices");
    for (int k = 0; k < 9; ++k)
        if (!std::isfinite(K[k])) return fail(h, -1, "K is not finite");
    // Cameras held still (create_sparsity_matrix(..., fixed_camera_indices), bundle_adjustment.py:6,13-14: their six
    // Jacobian columns are structurally zero, so scipy's finite differences leave them zero, the gradient and the
    // step of those parameters are zero and x_scale='jac' gives them scale 1).  Here: their observations are left out
    // of the CAMERA-MAJOR lists (and of the block-pair lists of the few-camera path), so every per-camera sum --
    // U_c, g_c, the reduced right-hand side, the Schur-diagonal block, the product of pass B -- is empty for them,
    // exactly as for a camera nobody observes; the point-major sweeps keep the observations (residual, cost, V_p,
    // g_p) and only ever multiply the camera part of their Jacobian with a camera vector that stays zero.
    std::vector<char> fixed((size_t)C, 0);
    for (int64_t c : h->fixed_next) {
        if (c < 0 || c >= C) return fail(h, -1, "fixed camera index %lld out of range [0,%lld)", (long long)c, (long long)C);
        fixed[(size_t)c] = 1;
    }
    h->n_fixed = 0;
    for (char f : fixed) h->n_fixed += f;
    // A new problem returns the handle to single-process operation: every transport (direct link, RCCL
    // communicator, callback) is torn down and has to be set up again after this call (include/sfmba.h).  The
    // staging buffer of the direct link stays allocated until sfmba_p2p_detach / _export / _destroy, because
    // peers may still have it mapped.
    // The drop is not silent: the next compute call on this handle fails (-1) until a transport is attached again
    // or single-rank operation is acknowledged (check_ready).
    const bool had_transport = multi_rank(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));                  // collectives in flight; a previous upload may still read the staging
    if (h->p2p.ready) p2p_close_peers(h);
    if (h->comm) { if (RcclApi* api = rccl_api()) (void)api->CommDestroy(h->comm); h->comm = nullptr; }
    if (had_transport) h->transport_dropped = true;

    // ---- incremental re-use ---------------------------------------------------------
    // The reference calls BA once per fused edge on a growing reconstruction (/root/reference/sfm_lite/sfm.py:59-71):
    // once every camera is registered, a new edge only appends points and their observations, so the argument
    // arrays of one call start with those of the previous call.  The converted arrays of the last problem stay in
    // pinned host memory and in HBM; this call compares as it converts, finds the first observation that differs
    // and uploads from there on only.  The structure tables are always rebuilt from the (complete) host arrays, by
    // the same code whatever was re-used: results are bitwise those of a fresh handle.
    auto& prev = h->prev;
    const bool f32 = h->f32_next;
    const int64_t ld = (N + 255) / 256 * 256;
    const size_t ldz = (size_t)ld;
    const int64_t n_cmp = (prev.valid && prev.f32 == f32) ? std::min(prev.N, N) : 0;
    prev.valid = false;                                          // until this call has completed
    auto& sg = h->stage;
    const size_t keep_obs = (size_t)n_cmp;
    HIPCHK(h, sg.uv.ensure(sizeof(double) * 2 * ldz, sizeof(double) * 2 * keep_obs));
    HIPCHK(h, sg.ci.ensure(sizeof(int) * ldz, sizeof(int) * keep_obs));
    HIPCHK(h, sg.pi.ensure(sizeof(int) * ldz, sizeof(int) * keep_obs));
    HIPCHK(h, sg.perm.ensure(sizeof(int) * ldz, 0));
    HIPCHK(h, sg.ptr.ensure(sizeof(int) * ((size_t)P + 1), 0));
    if (f32) HIPCHK(h, sg.uvf.ensure(sizeof(float) * 2 * ldz, sizeof(float) * 2 * keep_obs));
    const bool try_pack = (h->dbg.packed_upload == 1 || (h->dbg.packed_upload != 0 && N >= 65536)) && C <= 65535;   // (small: two launches > the bytes)
    const bool packed_prefix_ok = try_pack && prev.packed && n_cmp > 0;      // the staged prefix holds valid packed entries
    if (try_pack) {
        HIPCHK(h, sg.ci16.ensure(sizeof(unsigned short) * ldz, sizeof(unsigned short) * keep_obs));
        HIPCHK(h, sg.uv16.ensure(sizeof(short) * 2 * ldz, sizeof(short) * 2 * keep_obs));
    }
    unsigned short* ci16 = try_pack ? sg.ci16.as<unsigned short>() : nullptr;
    short* uv16 = try_pack ? sg.uv16.as<short>() : nullptr;
    std::vector<char> not16((size_t)h->pool.parts_for(N), 0);       // a pixel that is no int16 integer: no packing
    double* uvs = sg.uv.as<double>();
    float* uvf = f32 ? sg.uvf.as<float>() : nullptr;
    int* ci = sg.ci.as<int>();
    int* pi = sg.pi.as<int>();
    int* perm = sg.perm.as<int>();
    int* ptr = sg.ptr.as<int>();

    // ---- pass 1 (parallel): range check, order check, conversion with comparison, camera histogram -------------
    const int parts = h->pool.parts_for(N);
    std::vector<int> hist((size_t)parts * (size_t)C, 0);
    std::vector<int64_t> bad((size_t)parts, -1), first_diff((size_t)parts, N);
    std::vector<char> unsorted((size_t)parts, 0);
    const int64_t per = (N + parts - 1) / parts;
    auto convert = [&](const int64_t* ord, int64_t n_compare) {
        h->pool.run(parts, [&](int t) {
            const int64_t b = std::min<int64_t>(N, t * per), e = std::min<int64_t>(N, (t + 1) * per);
            int* hc = hist.data() + (size_t)t * (size_t)C;
            for (int64_t c = 0; c < C; ++c) hc[c] = 0;
            int64_t fd = N;
            for (int64_t k = b; k < e; ++k) {
                const int64_t s = ord ? ord[k] : k;
                const int64_t cv = cam[s], pv = pt[s];
                if (cv < 0 || cv >= C || pv < 0 || pv >= P) { if (bad[t] < 0) bad[t] = s; continue; }
                if (!ord && k > 0 && pv < pt[k - 1]) unsorted[t] = 1;
                double u0, u1;
                if (uv) { u0 = uv[2 * s]; u1 = uv[2 * s + 1]; }
                else { u0 = (double)uv_i64[2 * s]; u1 = (double)uv_i64[2 * s + 1]; }          // as numpy promotes
                if (!fixed[(size_t)cv]) ++hc[cv];
                const bool same = k < n_compare && ci[k] == (int)cv && pi[k] == (int)pv && uvs[2 * k] == u0 && uvs[2 * k + 1] == u1;
                if (ci16 && !(same && packed_prefix_ok)) {     // (an unchanged entry is packed already, if the previous call packed)
                    const bool in = std::fabs(u0) < 32768.0 && std::fabs(u1) < 32768.0;
                    const short q0 = in ? (short)u0 : (short)0, q1 = in ? (short)u1 : (short)0;
                    if (!in || (double)q0 != u0 || (double)q1 != u1) not16[t] = 1;
                    ci16[k] = (unsigned short)cv; uv16[2 * k] = q0; uv16[2 * k + 1] = q1;
                }
                if (same) continue;
                if (fd == N) fd = k;
                ci[k] = (int)cv; pi[k] = (int)pv; uvs[2 * k] = u0; uvs[2 * k + 1] = u1;
                if (uvf) { uvf[2 * k] = (float)u0; uvf[2 * k + 1] = (float)u1; }    // integer pixels up to 2^24 are exact
            }
            first_diff[t] = fd;
        });
    };
    convert(nullptr, n_cmp);
    auto report_bad = [&]() -> int {
        for (int t = 0; t < parts; ++t) {
            const int64_t i = bad[t];
            if (i < 0) continue;
            if (cam[i] < 0 || cam[i] >= C)
                return fail(h, -1, "camera_indices[%lld]=%lld out of range [0,%lld)", (long long)i, (long long)cam[i], (long long)C);
            return fail(h, -1, "point_indices[%lld]=%lld out of range [0,%lld)", (long long)i, (long long)pt[i], (long long)P);
        }
        return 0;
    };
    CHK(report_bad());                  // (numpy's fancy indexing would raise IndexError in the reference)
    bool sorted = true;
    for (int t = 0; t < parts; ++t) sorted = sorted && !unsorted[t];
    h->permuted = !sorted;
    h->order.clear();
    if (!sorted) {                       // any order is accepted; the kernels want point-major.  No re-use on this path.
        h->order.resize(N);
        std::iota(h->order.begin(), h->order.end(), (int64_t)0);
        std::stable_sort(h->order.begin(), h->order.end(), [&](int64_t a, int64_t b) { return pt[a] < pt[b]; });
        convert(h->order.data(), 0);
    }
    int64_t fdiff = N;
    for (int t = 0; t < parts; ++t) fdiff = std::min(fdiff, first_diff[t]);
    fdiff = std::min(fdiff, n_cmp);      // nothing beyond the compared prefix is on the device
    for (size_t k = (size_t)N; k < ldz; ++k) {                       // padding up to the next multiple of 256
        ci[k] = 0; pi[k] = 0; uvs[2 * k] = 0.0; uvs[2 * k + 1] = 0.0;
        if (uvf) { uvf[2 * k] = 0.f; uvf[2 * k + 1] = 0.f; }
        if (ci16) { ci16[k] = 0; uv16[2 * k] = 0; uv16[2 * k + 1] = 0; }
    }
    bool packed = try_pack;
    for (char f : not16) packed = packed && !f;
    tp1 = now_s();

    // ---- camera-major order: stable counting sort of the positions by camera (per-part histograms, so that the
    // parts scatter independently and the order inside a camera stays the point-major one) -----------------------
    std::vector<int> cam_ptr((size_t)C + 1);
    {
        int run = 0;
        for (int64_t c = 0; c < C; ++c) {
            cam_ptr[c] = run;
            for (int t = 0; t < parts; ++t) { int& v = hist[(size_t)t * (size_t)C + c]; const int n = v; v = run; run += n; }
        }
        cam_ptr[C] = run;
    }
    // ---- pass 2 (parallel): run offsets of the points, camera-major permutation ---------------------------------
    // ptr[p] = first position whose point index is >= p (pi is non-decreasing): every run start k writes the
    // entries (pi[k-1], pi[k]], so the parts touch disjoint pieces of ptr.
    // The camera-major order itself is sorted on the DEVICE (k_cam_hist / k_cam_offsets / k_cam_scatter: the same stable
    // order) unless the camera counters do not fit the LDS; the host then only needs the run offsets.
    // (from 64k observations on: below that the three launches cost more than the host's sort)
    const bool cm_device = (h->dbg.cm_device == 1 || (h->dbg.cm_device != 0 && N >= 65536)) && sizeof(int) * (size_t)C <= kLdsDynMax;
    h->pool.run(parts, [&](int t) {
        const int64_t b = std::min<int64_t>(N, t * per), e = std::min<int64_t>(N, (t + 1) * per);
        int* off = hist.data() + (size_t)t * (size_t)C;
        for (int64_t k = b; k < e; ++k) {
            const int lo = k == 0 ? -1 : pi[k - 1];
            for (int q = lo + 1; q <= pi[k]; ++q) ptr[q] = (int)k;
            if (!cm_device && !fixed[(size_t)ci[k]]) perm[off[ci[k]]++] = (int)k;
        }
    });
    for (int64_t q = (int64_t)pi[N - 1] + 1; q <= P; ++q) ptr[q] = (int)N;
    ts1 = now_s();
    if (!cm_device)
        for (size_t k = (size_t)cam_ptr[C]; k < ldz; ++k) perm[k] = 0;  // (shorter than N when cameras are held still)

    for (int k = 0; k < 9; ++k) h->K.k[k] = K[k];
    h->f32 = f32;
    h->C = C; h->P = P; h->N = N; h->n = 6 * C + 3 * P;
    h->N_total = N;
    h->ld = ld;
    // wave ranges: cut at point boundaries, >= T observations each
    const int64_t total_waves = (int64_t)h->n_cu * kWavesPerSweepBlock;
    const int64_t T = std::max<int64_t>(64, (N + total_waves - 1) / total_waves);
    std::vector<int2>& ranges = h->host_ranges;
    ranges.clear();
    {
        int64_t start = 0;
        for (int64_t p = 0; p < P; ++p) {
            const int64_t endp = ptr[p + 1];
            if (endp - start >= T || (p == P - 1 && endp > start)) {
                ranges.push_back(make_int2((int)start, (int)endp));
                start = endp;
            }
        }
    }
    h->n_ranges = (int)ranges.size();
    ts2 = now_s();
    // step table of the sweeps: per wave range, batches of <= 64 observations that end on a point
    // boundary; a point with more than 64 observations is one step of its own
    std::vector<int2>& wsteps = h->host_wsteps;
    std::vector<int2>& steps = h->host_steps;
    wsteps.resize(ranges.size());
    steps.clear();
    {   // the ranges are independent: every worker walks a slice of them into its own list (the walk is a chain of
        // dependent reads of pi / ptr -- 1.2-1.4 ms of a 3.6 ms call at 1M observations when one thread did all of it),
        // the lists are concatenated in range order afterwards
        const int sparts = (int)std::max<size_t>(1, std::min<size_t>((size_t)parts, ranges.size() / 64));
        std::vector<std::vector<int2>> local((size_t)sparts);
        const size_t rper = (ranges.size() + (size_t)sparts - 1) / (size_t)sparts;
        h->pool.run(sparts, [&](int t) {
            std::vector<int2>& out = local[(size_t)t];
            const size_t w0 = std::min(ranges.size(), (size_t)t * rper), w1 = std::min(ranges.size(), w0 + rper);
            out.reserve((w1 - w0) * 6);
            for (size_t w = w0; w < w1; ++w) {
                const int first = (int)out.size();                       // (relative to the slice; rebased below)
                int64_t pos = ranges[w].x;
                const int64_t end = ranges[w].y;
                while (pos < end) {
                    const int64_t pfirst = pi[pos];
                    if (ptr[pfirst + 1] - pos > 64) {                    // long run (pos is always a run start)
                        out.push_back(make_int2((int)pos, (int)(ptr[pfirst + 1] - pos)));
                        pos = ptr[pfirst + 1];
                        continue;
                    }
                    // largest run boundary <= pos + 64
                    int64_t lim = std::min<int64_t>(pos + 64, end), cut;
                    if (lim == end) cut = end;
                    else { const int64_t pl = pi[lim]; cut = (ptr[pl] == lim) ? lim : ptr[pl]; }   // lim inside a run -> its start
                    out.push_back(make_int2((int)pos, (int)(cut - pos)));
                    pos = cut;
                }
                wsteps[w] = make_int2(first, (int)out.size() - first);
            }
        });
        size_t total = 0;
        for (auto& v : local) total += v.size();
        steps.reserve(total);
        for (int t = 0; t < sparts; ++t) {
            const int base = (int)steps.size();
            const size_t w0 = std::min(ranges.size(), (size_t)t * rper), w1 = std::min(ranges.size(), w0 + rper);
            for (size_t w = w0; w < w1; ++w) wsteps[w].x += base;
            steps.insert(steps.end(), local[(size_t)t].begin(), local[(size_t)t].end());
        }
    }
    h->n_steps = (int)steps.size();
    ts3 = now_s();
    // chunk table of the camera-major kernels: every camera gets at least one chunk (an empty one writes its
    // zeros), runs longer than chunk_len are cut; one 256-thread workgroup per chunk
    std::vector<int4>& chunks = h->host_chunks;
    std::vector<int>& chunk_ptr = h->host_chunk_ptr;
    chunks.clear();
    chunk_ptr.resize((size_t)C + 1);
    {
        // a camera of up to 4096 observations is one workgroup (16 per lane) and needs no combine launch
        int64_t chunk_len = std::max<int64_t>(4096, (N + 2 * h->n_cu - 1) / (2 * h->n_cu));
        if (h->dbg.cam_chunk > 0) chunk_len = h->dbg.cam_chunk;
        h->cam_multi = false;
        for (int64_t c = 0; c < C; ++c) {
            chunk_ptr[c] = (int)chunks.size();
            const int b = cam_ptr[c], e = cam_ptr[c + 1];
            const int nch = std::max<int>(1, (int)((e - b + chunk_len - 1) / chunk_len));
            if (nch > 1) h->cam_multi = true;
            for (int j = 0; j < nch; ++j)
                chunks.push_back(make_int4((int)c, (int)std::min<int64_t>(e, b + j * chunk_len),
                                           (int)std::min<int64_t>(e, b + (j + 1) * chunk_len), nch));
        }
        chunk_ptr[C] = (int)chunks.size();
        h->n_chunks = (int)chunks.size();
        // XCD-aware chunks for pass B of the Schur product (many points).  Every camera-major pass gathers one record
        // per observation from a table of P x 48 bytes; workgroup i runs on XCD i mod 8 and each XCD has its own 4 MiB
        // L2: with one chunk per camera every L2 sees the WHOLE table (48 MB at a million points).  A camera's list is
        // ascending in the point index, so it is cut at the eight point-range boundaries P k / 8: chunk 8 c + k runs on
        // XCD k and touches points of range k only, each L2 serves an eighth of the table (pass B at 5000 / 1M / 10M:
        // 201 -> 134 us); the eight partial rows of a camera are added by k_cam_combine and the PCG tail runs behind
        // it (k_pcg_tail).  Pass B only: the passes with 27 sums per workgroup (K3, rhs + preconditioner) lose more to
        // eight times as many block reductions than they gain (208 -> 320 us, 188 -> 257 us).
        std::vector<int4>& chunks_b = h->host_chunks_b;
        std::vector<int>& chunk_ptr_b = h->host_chunk_ptr_b;
        chunks_b.clear();
        chunk_ptr_b.clear();
        h->xcd_b = h->dbg.xcd_chunks == 1 || (h->dbg.xcd_chunks != 0 && P >= 250000);
        if (h->xcd_b) {
            // row (8 g + k) 4 + j = camera 4 g + j, range k: the four waves of workgroup 8 g + k (XCD k) take the pieces of
            // four cameras over the same point range (k_cam_schur_w); cameras behind the last one are padding (camera -1)
            constexpr int kX = kWaveChunkRanges, kG = kWaveChunkCams;
            static_assert(kX == 8, "one range per XCD");
            const int64_t groups = (C + kG - 1) / kG;
            chunks_b.assign((size_t)(groups * kX * kG), make_int4(-1, 0, 0, kX));
            for (int64_t c = 0; c < C && !cm_device; ++c) {            // (device-side sort: k_xcd_chunks fills the table)
                const int b = cam_ptr[c], e = cam_ptr[c + 1];
                int prev = b;
                for (int k = 0; k < kX; ++k) {
                    int bound = e;
                    if (k + 1 < kX) {
                        const int64_t p_hi = P * (int64_t)(k + 1) / kX;          // first point of the next range
                        int lo = prev, hi = e;                                   // lower bound over pi[perm[.]] (ascending)
                        while (lo < hi) { const int mid = (lo + hi) >> 1; if (pi[perm[mid]] < p_hi) lo = mid + 1; else hi = mid; }
                        bound = lo;
                    }
                    chunks_b[(size_t)(((c / kG) * kX + k) * kG + c % kG)] = make_int4((int)c, prev, bound, kX);
                    prev = bound;
                }
            }
        }
        h->n_chunks_b = (int)chunks_b.size();
    }
    // few cameras: for every block pair (a <= b) the points seen by both cameras, with multiplicity (a point seen
    // m_a, m_b times contributes m_a m_b times); two counting passes over the runs.  The diagonal pairs' workgroups
    // also sum the reduced right-hand side of their camera, over the entries that pair an observation with itself.
    h->dense = 6 * C <= kDenseMaxN && h->dbg.dense != 0;
    std::vector<int> cov_ptr, cov_pt;
    std::vector<int2> blk_ab;
    if (h->dense) {
        const int nblk = (int)(C * (C + 1) / 2);
        cov_ptr.assign((size_t)nblk + 1, 0);
        // (unordered pairs i <= j: a pair of different cameras is one entry of its block, two observations of the same
        // camera by the same point are two, an observation with itself one -- what the ordered double loop counted)
        // Two passes over the points, count and fill; from 32k points on split over a few threads by contiguous point
        // ranges: per-part counts per block, then part t's entries of a block follow part t - 1's -- ascending point order
        // inside a block, whatever the number of parts.  (Below that size waking the pool costs more than the passes:
        // measured at the SceauxCastle scale, 0.10 ms single-threaded against 0.13 ms on four threads.)
        const int pp = P >= 32768 ? 4 : 1;
        std::vector<int> cnt((size_t)pp * (size_t)nblk, 0);
        const int64_t pper = (P + pp - 1) / pp;
        auto each_pair = [&](int t, auto&& visit) {
            const int64_t p0 = std::min<int64_t>(P, t * pper), p1 = std::min<int64_t>(P, p0 + pper);
            for (int64_t p = p0; p < p1; ++p)
                for (int i = ptr[p]; i < ptr[p + 1]; ++i)
                    for (int j = i; j < ptr[p + 1]; ++j) {
                        const int a = std::min(ci[i], ci[j]), b = std::max(ci[i], ci[j]);
                        if (fixed[(size_t)a] || fixed[(size_t)b]) continue;
                        visit((int)p, dense_block_index(a, b, (int)C), i == j, a == b);
                    }
        };
        h->pool.run(pp, [&](int t) {
            int* ct = cnt.data() + (size_t)t * (size_t)nblk;
            each_pair(t, [&](int, int blk, bool self, bool same_cam) { ct[blk] += (self || !same_cam) ? 1 : 2; });
        });
        int64_t total = 0;
        for (int blk = 0; blk < nblk; ++blk) {
            cov_ptr[(size_t)blk] = (int)std::min<int64_t>(total, INT32_MAX);
            for (int t = 0; t < pp; ++t) { int& v = cnt[(size_t)t * (size_t)nblk + blk]; const int n_e = v; v = (int)std::min<int64_t>(total, INT32_MAX); total += n_e; }
        }
        cov_ptr[(size_t)nblk] = (int)std::min<int64_t>(total, INT32_MAX);
        if (total > ((int64_t)1 << 26)) h->dense = false;      // very long tracks: the pair lists would not pay
        else {
            cov_pt.resize((size_t)std::max<int64_t>(1, total));
            h->pool.run(pp, [&](int t) {
                int* fill = cnt.data() + (size_t)t * (size_t)nblk;
                each_pair(t, [&](int p, int blk, bool self, bool same_cam) {
                    int& f = fill[blk];
                    cov_pt[(size_t)f++] = self ? ~p : p;                       // (~p: the term of the right-hand side)
                    if (!self && same_cam) cov_pt[(size_t)f++] = p;
                });
            });
            blk_ab.resize((size_t)nblk);
            for (int a = 0; a < (int)C; ++a)
                for (int b = a; b < (int)C; ++b) blk_ab[(size_t)dense_block_index(a, b, (int)C)] = make_int2(a, b);
            h->n_blk = nblk;
        }
    }
    tp2 = now_s();
    h->lds_tab = (size_t)C * kCamRow * sizeof(double) <= kLdsDynMax;
    h->lds_vec = (size_t)C * 6 * sizeof(double) <= kLdsDynMax;
    if (h->dbg.tab_lds == 0) h->lds_tab = false;               // test hooks (sfmba_debug_option): force the L2 placements
    if (h->dbg.vec_lds == 0) h->lds_vec = false;
    h->sweep_rc = C <= kRcMaxCams && (size_t)C * kRcRow * sizeof(double) <= kLdsDynMax && h->dbg.sweep_rc != 0 &&
                  h->dbg.sweep_rc != 2;                        // (2: test hook -- the table in global memory whatever the size)
    h->sweep_rc_g = !h->sweep_rc && h->dbg.sweep_rc != 0;      // too many cameras for the LDS: the table lives in L2
    h->pcg_fused = (h->lds_vec || h->sweep_rc) && C <= kSweepThreads;
    if (h->dbg.pcg_fused == 0) h->pcg_fused = false;
    // fp32 operands in the implicit Schur product: with fp32 storage (BASELINE config 5 names it), or on request
    h->mixed = (h->sweep_rc || h->sweep_rc_g) && (h->dbg.pcg_mixed == 1 || (h->dbg.pcg_mixed != 0 && f32));
    h->mixed_b = h->mixed && h->dbg.pcg_mixed_b != 0;
    h->jfree = h->dbg.jfree == 1 && h->lds_tab;

    // ---- device arrays: grow-only; the index / pixel arrays keep their re-used prefix when they grow -------------
    const size_t esz = f32 ? sizeof(float) : sizeof(double);     // element size of the per-observation streams
    const size_t keep = (size_t)fdiff;
    HIPCHK(h, h->cam_idx.ensure_keep(sizeof(int) * ldz, sizeof(int) * keep));
    HIPCHK(h, h->pt_idx.ensure_keep(sizeof(int) * ldz, sizeof(int) * keep));
    HIPCHK(h, h->uv.ensure_keep(esz * 2 * ldz, esz * 2 * keep));
    // run offsets of the points before the first changed observation's point are unchanged
    // (entries up to the point of the last unchanged observation are determined by unchanged positions alone)
    const int64_t p_keep = keep == 0 ? 0 : std::min<int64_t>(std::min<int64_t>(prev.P, P), (int64_t)pi[fdiff - 1] + 1);
    HIPCHK(h, h->pt_ptr.ensure_keep(sizeof(int) * ((size_t)P + 1), sizeof(int) * (size_t)p_keep));
    // the structure tables: one device buffer, one pinned staging buffer, one copy
    struct Piece { const void* src; size_t bytes; DevView* view; size_t off; };
    Piece pieces[11] = {
        {cam_ptr.data(), sizeof(int) * cam_ptr.size(), &h->cam_ptr_dev, 0},
        {ranges.data(), sizeof(int2) * ranges.size(), &h->ranges, 0}, {wsteps.data(), sizeof(int2) * wsteps.size(), &h->wsteps, 0},
        {steps.data(), sizeof(int2) * steps.size(), &h->steps, 0}, {chunks.data(), sizeof(int4) * chunks.size(), &h->cam_chunks, 0},
        {chunk_ptr.data(), sizeof(int) * chunk_ptr.size(), &h->cam_chunk_ptr, 0},
        {cov_ptr.data(), h->dense ? sizeof(int) * cov_ptr.size() : 0, &h->cov_ptr, 0},
        {cov_pt.data(), h->dense ? sizeof(int) * cov_pt.size() : 0, &h->cov_pt, 0},
        {blk_ab.data(), h->dense ? sizeof(int2) * blk_ab.size() : 0, &h->blk_ab, 0},
        {h->host_chunks_b.data(), sizeof(int4) * h->host_chunks_b.size(), &h->cam_chunks_b, 0},
        {h->host_chunk_ptr_b.data(), sizeof(int) * h->host_chunk_ptr_b.size(), &h->cam_chunk_ptr_b, 0}};
    size_t tables_bytes = 0;
    for (auto& pc : pieces) { pc.off = tables_bytes; tables_bytes += (pc.bytes + 255) / 256 * 256; }
    HIPCHK(h, h->tables.ensure(tables_bytes + 256));
    HIPCHK(h, sg.tables.ensure(tables_bytes + 256, 0));
    for (auto& pc : pieces) {
        if (pc.bytes) memcpy(sg.tables.as<char>() + pc.off, pc.src, pc.bytes);
        pc.view->p = h->tables.as<char>() + pc.off;
    }
    HIPCHK(h, h->xa.ensure(sizeof(double) * h->n));
    HIPCHK(h, h->xb.ensure(sizeof(double) * h->n));
    HIPCHK(h, h->tabA.ensure(sizeof(double) * cam_table_doubles((int)C)));
    HIPCHK(h, h->tabB.ensure(sizeof(double) * cam_table_doubles((int)C)));
    HIPCHK(h, h->r.ensure(esz * 2 * ldz));
    HIPCHK(h, h->J.ensure(esz * 12 * ldz));
    HIPCHK(h, h->cm_perm.ensure(sizeof(int) * ldz));
    HIPCHK(h, h->cm_pt.ensure(sizeof(int) * ldz));
    HIPCHK(h, h->cm_uv.ensure(esz * 2 * ldz));
    HIPCHK(h, h->cam_partial.ensure(sizeof(double) * std::max<size_t>(27 * chunks.size(), 27 * h->host_chunks_b.size())));
    HIPCHK(h, h->recA.ensure(sizeof(double) * kRec * P));
    HIPCHK(h, h->recB.ensure(sizeof(double) * kRec * P));
    // One 128-byte gather record per point for k_cam_rhs_diag pays once the point tables no longer sit in the L2s
    // (1M points: 432 -> 188 us); at 100k points the two 4.8 MB tables it replaces are L2-resident and k_prep's 11 MB of
    // extra writes cost what the pass gains
    h->use_rhsrec = h->dbg.rhsrec == 1 || (h->dbg.rhsrec != 0 && P >= 250000);
    if (h->use_rhsrec) HIPCHK(h, h->rhsrec.ensure(sizeof(double) * kRhsRec * P));
    if (h->dense) {
        HIPCHK(h, h->Sblk.ensure(sizeof(double) * 36 * blk_ab.size()));
    }
    HIPCHK(h, h->t1.ensure(esz * 2 * ldz));
    HIPCHK(h, h->V.ensure(sizeof(double) * 6 * P));
    HIPCHK(h, h->Vinv.ensure(sizeof(double) * (kVinvInRec < 0 ? kVinvRow * P : 8)));
    HIPCHK(h, h->gp.ensure(sizeof(double) * 3 * P + 16));      // (+16: zeroed in 16-byte units)
    HIPCHK(h, h->edge.ensure(sizeof(double) * 2 * kEdgeRow * (size_t)((N + 63) / 64)));
    HIPCHK(h, h->e.ensure(sizeof(double) * 3 * P));
    HIPCHK(h, h->g.ensure(sizeof(double) * h->n));
    HIPCHK(h, h->si.ensure(sizeof(double) * h->n));
    HIPCHK(h, h->sg.ensure(sizeof(double) * h->n));
    HIPCHK(h, h->p.ensure(sizeof(double) * h->n + 16));
    HIPCHK(h, h->Dc.ensure(sizeof(double) * 6 * C));
    HIPCHK(h, h->Minv.ensure(sizeof(double) * 21 * C));
    HIPCHK(h, h->vecs.ensure(sizeof(double) * 2 * kPcgVecs * 6 * C));
    HIPCHK(h, h->vtmp.ensure(sizeof(double) * 6 * C));
    HIPCHK(h, h->vcm.ensure(sizeof(double) * 6 * C));
    if (h->sweep_rc_g) HIPCHK(h, h->rctab.ensure(sizeof(double) * kRcRow * (size_t)C));
    if (h->mixed) {
        HIPCHK(h, h->rt32.ensure(sizeof(float) * kRt32 * (size_t)C));
        HIPCHK(h, h->rtd.ensure(sizeof(double) * kRt32 * (size_t)C));
        HIPCHK(h, h->rec32.ensure(sizeof(float) * kRec32 * (size_t)P));
        if (h->sweep_rc_g) HIPCHK(h, h->rctab32.ensure(sizeof(float) * kRc32Row * (size_t)C));
    }
    // k_update_scale: cameras one element per thread, points kScalePts points per thread (all loads of a thread in flight
    // together); at most 1024 partial rows, summed by k_jdot's rider workgroup
    h->scale_pts = P >= 65536 ? 2 : 1;
    h->red_bc = grid_1d(6 * C, 256, 32);
    h->red_grid = h->red_bc + grid_1d(P, 256 * h->scale_pts, 992);
    HIPCHK(h, h->part.ensure(sizeof(double) * (size_t)(2 * kPartRows * kNQ)));
    HIPCHK(h, h->ctrl.ensure(2 * sizeof(PcgCtrl)));
    HIPCHK(h, h->pcg_part.ensure(sizeof(double) * 4 * C));
    HIPCHK(h, h->arena_own.ensure(sizeof(double) * (size_t)sfmba_exchange_doubles(C)));
    h->arena = h->arena_own.as<double>();
    h->ar_fn = nullptr; h->ar_ctx = nullptr;
    h->x = h->xa.as<double>(); h->x_new = h->xb.as<double>();
    h->tab = h->tabA.as<double>(); h->tab_new = h->tabB.as<double>();
    h->rec = h->recA.as<double>(); h->rec_new = h->recB.as<double>();

    // ---- uploads: observations from the first changed one on, run offsets from its point on, structure tables ----
    auto up = [&](void* dst, const void* src, size_t elem, size_t from, size_t to) -> hipError_t {
        if (to <= from) return hipSuccess;
        return hipMemcpyAsync(static_cast<char*>(dst) + elem * from, static_cast<const char*>(src) + elem * from,
                              elem * (to - from), hipMemcpyHostToDevice, h->stream);
    };
    if (packed) {
        HIPCHK(h, h->ci16_dev.ensure(sizeof(unsigned short) * ldz));
        HIPCHK(h, h->uv16_dev.ensure(sizeof(short) * 2 * ldz));
        HIPCHK(h, up(h->ci16_dev.p, ci16, sizeof(unsigned short), keep, ldz));
        HIPCHK(h, up(h->uv16_dev.p, uv16, 2 * sizeof(short), keep, ldz));
    } else {
        HIPCHK(h, up(h->cam_idx.p, ci, sizeof(int), keep, ldz));
        HIPCHK(h, up(h->pt_idx.p, pi, sizeof(int), keep, ldz));
        HIPCHK(h, up(h->uv.p, f32 ? (const void*)uvf : (const void*)uvs, 2 * esz, keep, ldz));
    }
    HIPCHK(h, up(h->pt_ptr.p, ptr, sizeof(int), (size_t)p_keep, (size_t)P + 1));
    if (packed) {
        if (keep < ldz) {
            hipLaunchKernelGGL(k_unpack_obs, dim3((unsigned)((ldz - keep + 255) / 256)), dim3(256), 0, h->stream,
                               (const unsigned short*)h->ci16_dev.as<unsigned short>(), (const short2*)h->uv16_dev.as<short2>(),
                               (int)keep, (int)ldz, f32 ? 1 : 0, h->cam_idx.as<int>(), h->uv.as<double>());
            LAUNCHED(h);
            hipLaunchKernelGGL(k_expand_pt_idx, dim3((unsigned)((P + 1 + 255) / 256)), dim3(256), 0, h->stream,
                               (const int*)h->pt_ptr.as<int>(), (int)P, (int)N, (int)ld, h->pt_idx.as<int>());
            LAUNCHED(h);
        }
    }
    h->obs_reused = fdiff;
    h->obs_uploaded = ld - fdiff;
    if (!(fdiff == N && n_cmp == N && prev.N == N && prev.P == P && h->C == C)) ++h->problem_gen;
    HIPCHK(h, hipMemcpyAsync(h->tables.p, sg.tables.p, tables_bytes, hipMemcpyHostToDevice, h->stream));
    {   // every zero-initialised array in ONE launch: the exchange arena; V, g_p (points without observations are never
        // written by the normal-block kernels: their blocks must be 0) and p (... and their step is 0); r; the point records
        ZeroJob z{};
        auto put = [&](int k, void* ptr_, size_t bytes) { z.p[k] = ptr_; z.n16[k] = (int64_t)(bytes / 16); };
        put(0, h->arena, sizeof(double) * (size_t)sfmba_exchange_doubles(C));
        put(1, h->V.p, sizeof(double) * 6 * P);
        put(2, h->gp.p, (sizeof(double) * 3 * P + 15) / 16 * 16);
        put(3, h->p.p, (sizeof(double) * h->n + 15) / 16 * 16);
        put(4, h->r.p, esz * 2 * ldz);
        put(5, h->recA.p, sizeof(double) * kRec * P);
        put(6, h->recB.p, sizeof(double) * kRec * P);
        int64_t most = 0;
        for (int k = 0; k < 7; ++k) most = std::max(most, z.n16[k]);
        hipLaunchKernelGGL(k_zero_many, dim3((unsigned)grid_1d(most, 256 * 4, 2048), 7), dim3(256), 0, h->stream, z);
        LAUNCHED(h);
    }
    if (cm_device) {
        const int B = (int)std::min<int64_t>(kSortSlices, (N + 63) / 64);
        const int per_slice = (int)(((N + B - 1) / B + 63) / 64 * 64);
        int key_bits = 0;
        while (((int64_t)1 << key_bits) < C) ++key_bits;
        HIPCHK(h, h->sort_hist.ensure(sizeof(int) * 2 * (size_t)B * (size_t)C));          // counts | offsets
        int* sort_off = h->sort_hist.as<int>() + (size_t)B * (size_t)C;
        const unsigned char* fixed_dev = nullptr;
        if (h->n_fixed > 0) {
            HIPCHK(h, h->fixed_dev.ensure((size_t)C));
            HIPCHK(h, hipMemcpyAsync(h->fixed_dev.p, fixed.data(), (size_t)C, hipMemcpyHostToDevice, h->stream));
            fixed_dev = h->fixed_dev.as<unsigned char>();
        }
        const size_t lds = sizeof(int) * (size_t)C;
        CHK(set_lds(h, k_cam_hist, lds));
        CHK(set_lds(h, k_cam_scatter, lds));
        hipLaunchKernelGGL(k_cam_hist, dim3(B), dim3(64), lds, h->stream, (const int*)h->cam_idx.as<int>(), fixed_dev, (int)N, (int)C,
                           per_slice, h->sort_hist.as<int>());
        LAUNCHED(h);
        hipLaunchKernelGGL(k_cam_offsets, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, h->stream, (const int*)h->sort_hist.as<int>(),
                           (const int*)h->cam_ptr_dev.as<int>(), B, (int)C, sort_off);
        LAUNCHED(h);
        hipLaunchKernelGGL(k_cam_scatter, dim3(B), dim3(64), lds, h->stream, (const int*)h->cam_idx.as<int>(), fixed_dev,
                           (const int*)h->pt_idx.as<int>(), (const double*)h->uv.as<double>(), f32 ? 1 : 0, (int)N, (int)C, per_slice,
                           key_bits, (const int*)sort_off, h->cm_pt.as<int>(), h->cm_uv.as<double>());
        LAUNCHED(h);
        if (h->xcd_b) {
            hipLaunchKernelGGL(k_xcd_chunks, dim3((unsigned)((h->n_chunks_b + 255) / 256)), dim3(256), 0, h->stream,
                               (const int*)h->cam_ptr_dev.as<int>(), (const int*)h->cm_pt.as<int>(), (int)C, (int)P,
                               h->cam_chunks_b.as<int4>());
            LAUNCHED(h);
        }
    } else {
        HIPCHK(h, hipMemcpyAsync(h->cm_perm.p, perm, sizeof(int) * ldz, hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(k_build_cam_major, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, h->stream, h->cm_perm.as<int>(),
                           h->pt_idx.as<int>(), h->uv.as<double>(), f32 ? 1 : 0, (int)N, h->cm_pt.as<int>(),
                           h->cm_uv.as<double>());
        LAUNCHED(h);
    }
    tp3 = now_s();
    HIPCHK(h, hipStreamSynchronize(h->stream));     // the host tables are rebuilt by the next call
    if (timing)
        fprintf(stderr, "sfmba: set_problem  structure = run offsets + camera-major permutation %.2f ms, wave ranges %.2f ms, step table %.2f ms, "
                        "chunks + pair lists %.2f ms\n", 1e3 * (ts1 - tp1), 1e3 * (ts2 - ts1), 1e3 * (ts3 - ts2), 1e3 * (tp2 - ts3));
    if (timing)
        fprintf(stderr, "sfmba: set_problem  convert+compare %.2f ms  structure %.2f ms  allocate+enqueue %.2f ms  upload wait %.2f ms"
                        "  (%lld of %lld observations re-used)\n",
                1e3 * (tp1 - tp0), 1e3 * (tp2 - tp1), 1e3 * (tp3 - tp2), 1e3 * (now_s() - tp3), (long long)fdiff, (long long)N);
    if (sorted) { prev.valid = true; prev.f32 = f32; prev.N = N; prev.P = P; }
    prev.packed = sorted && packed;
    h->have_problem = true;
    return 0;
}

int sfmba_residuals(sfmba_handle* h, const double* x, double* r_out) {
    CHK(enter(h));
    CHK(check_ready(h, x));
    if (!r_out) return fail(h, -1, "r_out is NULL");
    CHK(upload_x(h, x));
    CHK(launch_cam_table(h, h->x, h->tab, h->rec));
    int np = 0;
    CHK((launch_resjac<false, true>(h, h->x, h->tab, &np)));
    return download_residuals(h, r_out);
}

int sfmba_residual_jacobian(sfmba_handle* h, const double* x, double* r_out, double* Jc_out, double* Jp_out) {
    CHK(enter(h));
    CHK(check_ready(h, x));
    if (!r_out || !Jc_out || !Jp_out) return fail(h, -1, "NULL output");
    CHK(upload_x(h, x));
    CHK(launch_cam_table(h, h->x, h->tab, h->rec));
    int np = 0;
    CHK((launch_resjac<true, true>(h, h->x, h->tab, &np, nullptr, nullptr, /*blocks=*/false)));   // (blocks = false: always stored)
    DevBuf jc_rm, jp_rm;
    HIPCHK(h, jc_rm.ensure(sizeof(double) * 12 * h->N));
    HIPCHK(h, jp_rm.ensure(sizeof(double) * 6 * h->N));
    hipLaunchKernelGGL(k_unpack_jac, dim3((h->N + 255) / 256), dim3(256), 0, h->stream, h->J.as<double>(),
                       (int)h->N, h->ld, h->f32 ? 1 : 0, jc_rm.as<double>(), jp_rm.as<double>());
    LAUNCHED(h);
    std::vector<double> tc(12 * h->N), tp(6 * h->N);
    CHK(download_residuals(h, r_out));
    HIPCHK(h, hipMemcpyAsync(tc.data(), jc_rm.p, sizeof(double) * 12 * h->N, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(tp.data(), jp_rm.p, sizeof(double) * 6 * h->N, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int64_t k = 0; k < h->N; ++k) {
        const int64_t d = h->permuted ? h->order[k] : k;
        memcpy(Jc_out + 12 * d, tc.data() + 12 * k, sizeof(double) * 12);
        memcpy(Jp_out + 6 * d, tp.data() + 6 * k, sizeof(double) * 6);
    }
    return 0;
}

int sfmba_normal_blocks(sfmba_handle* h, const double* x, double* U, double* V, double* gc, double* gp) {
    CHK(enter(h));
    CHK(check_ready(h, x));
    CHK(upload_x(h, x));
    CHK(launch_cam_table(h, h->x, h->tab, h->rec));
    int np = 0;
    CHK((launch_resjac<true, true>(h, h->x, h->tab, &np)));
    CHK(launch_normal_blocks(h, h->x, h->tab, h->rec));
    std::vector<double> ugc(27 * h->C);
    HIPCHK(h, hipMemcpyAsync(ugc.data(), h->Ugc(), sizeof(double) * 27 * h->C, hipMemcpyDeviceToHost, h->stream));
    if (V) HIPCHK(h, hipMemcpyAsync(V, h->V.p, sizeof(double) * 6 * h->P, hipMemcpyDeviceToHost, h->stream));
    if (gp) HIPCHK(h, hipMemcpyAsync(gp, h->gp.p, sizeof(double) * 3 * h->P, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int64_t c = 0; c < h->C; ++c) {
        if (U) memcpy(U + 21 * c, ugc.data() + 27 * c, sizeof(double) * 21);
        if (gc) memcpy(gc + 6 * c, ugc.data() + 27 * c + 21, sizeof(double) * 6);
    }
    return 0;
}

int sfmba_schur_matvec(sfmba_handle* h, const double* x, const double* dc, const double* dp, const double* v, double* y) {
    CHK(enter(h));
    CHK(check_ready(h, x));
    if (!dc || !dp || !v || !y) return fail(h, -1, "NULL argument");
    CHK(upload_x(h, x));
    CHK(launch_cam_table(h, h->x, h->tab, h->rec));
    int np = 0;
    CHK((launch_resjac<true, true>(h, h->x, h->tab, &np)));
    CHK(launch_normal_blocks(h, h->x, h->tab, h->rec));
    // stage dp in e (as explicit diagonal), v in pk -- camera vectors are plane-major on the device
    const int64_t C = h->C;
    std::vector<double> vp(6 * C);
    for (int64_t c = 0; c < C; ++c)
        for (int k = 0; k < 6; ++k) vp[k * C + c] = v[6 * c + k];
    HIPCHK(h, hipMemcpyAsync(h->e.p, dp, sizeof(double) * 3 * h->P, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->vtmp.p, vp.data(), sizeof(double) * 6 * C, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_point_prep, dim3((h->P + 255) / 256), dim3(256), 0, h->stream, h->V.as<double>(),
                       h->gp.as<double>(), (const double*)nullptr, h->e.as<double>(), (int)h->P, 0.0,
                       vinv_ptr(h), (double*)nullptr, (const double*)nullptr, (double*)nullptr);
    LAUNCHED(h);
    CHK(launch_mixed_prep(h, h->x, h->tab));
    CHK(schur_product_standalone(h, h->vtmp.as<double>()));
    CHK(exchange(h, h->acc(), 6 * C, 0));
    std::vector<double> a(6 * C);
    HIPCHK(h, hipMemcpyAsync(a.data(), h->acc(), sizeof(double) * 6 * C, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int64_t c = 0; c < C; ++c)
        for (int k = 0; k < 6; ++k) y[6 * c + k] = a[k * C + c] + dc[6 * c + k] * v[6 * c + k];
    HIPCHK(h, hipMemsetAsync(h->acc(), 0, sizeof(double) * 6 * C, h->stream));
    return 0;
}

int sfmba_dense_schur(sfmba_handle* h, const double* x, const double* dc, const double* dp, const double* rhs,
                      double* S_out, double* sol_out) {
    CHK(enter(h));
    CHK(check_ready(h, x));
    if (!dc || !dp || !rhs || !sol_out) return fail(h, -1, "NULL argument");
    if (!h->dense) return fail(h, -1, "the dense reduced-camera path needs 6 * n_cameras <= %d", kDenseMaxN);
    const int64_t C = h->C;
    CHK(upload_x(h, x));
    CHK(launch_cam_table(h, h->x, h->tab, h->rec));
    int np = 0;
    CHK((launch_resjac<true, true>(h, h->x, h->tab, &np)));
    CHK(launch_normal_blocks(h, h->x, h->tab, h->rec));
    std::vector<double> ugc(27 * C), planes(6 * C), accp(6 * C);
    HIPCHK(h, hipMemcpyAsync(ugc.data(), h->Ugc(), sizeof(double) * 27 * C, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int64_t c = 0; c < C; ++c)
        for (int k = 0; k < 6; ++k) {
            planes[k * C + c] = dc[6 * c + k];
            accp[k * C + c] = -ugc[27 * c + 21 + k] - rhs[6 * c + k];      // the kernel solves S y = -g_c - acc
        }
    HIPCHK(h, hipMemcpyAsync(h->e.p, dp, sizeof(double) * 3 * h->P, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->Dc.p, planes.data(), sizeof(double) * 6 * C, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->acc(), accp.data(), sizeof(double) * 6 * C, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_point_prep, dim3((h->P + 255) / 256), dim3(256), 0, h->stream, h->V.as<double>(),
                       h->gp.as<double>(), (const double*)nullptr, h->e.as<double>(), (int)h->P, 0.0,
                       vinv_ptr(h), (double*)nullptr, (const double*)nullptr, (double*)nullptr);
    LAUNCHED(h);
    CHK(launch_dense_solve(h, 1e-14, 40 * 6 * (int)C, /*rhs=*/false));   // (test entry: to the end, to be compared with a direct solve)
    std::vector<double> blk(36 * (size_t)h->n_blk), sol(6 * C);
    HIPCHK(h, hipMemcpyAsync(blk.data(), h->Sblk.p, sizeof(double) * blk.size(), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(sol.data(), h->vecs.p, sizeof(double) * 6 * C, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int64_t c = 0; c < C; ++c)
        for (int k = 0; k < 6; ++k) sol_out[6 * c + k] = sol[k * C + c];
    if (S_out) {
        const int64_t n = 6 * C;
        for (int64_t a = 0; a < C; ++a)
            for (int64_t b = a; b < C; ++b) {
                const double* B = blk.data() + 36 * (size_t)dense_block_index((int)a, (int)b, (int)C);
                for (int u = 0; u < 6; ++u)
                    for (int v = 0; v < 6; ++v) {
                        double val = -B[6 * u + v];
                        if (a == b) {
                            const int lo = std::min(u, v), hi = std::max(u, v);
                            val += ugc[27 * a + (lo * 6 - lo * (lo - 1) / 2 + (hi - lo))];
                            if (u == v) val += dc[6 * a + u];
                        }
                        S_out[(6 * a + u) * n + 6 * b + v] = val;
                        if (a != b) S_out[(6 * b + v) * n + 6 * a + u] = val;
                    }
            }
    }
    return 0;
}

int sfmba_time_kernel(sfmba_handle* h, const double* x, int32_t which, int32_t reps, double* avg_us) {
    CHK(enter(h));
    CHK(check_ready(h, x));
    if (!avg_us || reps <= 0) return fail(h, -1, "bad reps / avg_us");
    CHK(upload_x(h, x));
    CHK(launch_cam_table(h, h->x, h->tab, h->rec));
    int np = 0;
    CHK((launch_resjac<true, true>(h, h->x, h->tab, &np)));
    if (which >= 2) {
        CHK(launch_normal_blocks(h, h->x, h->tab, h->rec));
        CHK(launch_update_scale(h, 1));
        hipLaunchKernelGGL(k_point_prep, dim3((h->P + 255) / 256), dim3(256), 0, h->stream, h->V.as<double>(),
                           h->gp.as<double>(), h->si.as<double>() + 6 * h->C, (const double*)nullptr, (int)h->P,
                           1e-6, vinv_ptr(h), h->rec + 3, (const double*)(h->x + 6 * h->C),
                           h->use_rhsrec ? h->rhsrec.as<double>() : (double*)nullptr);
        LAUNCHED(h);
        CHK(launch_mixed_prep(h, h->x, h->tab));
        // v = the camera slice of the gradient, as plane-major planes (and camera-major when v is not staged in LDS)
        hipLaunchKernelGGL(k_transpose, dim3((unsigned)((6 * h->C + 255) / 256)), dim3(256), 0, h->stream,
                           (const double*)h->g.as<double>(), (int)h->C, 6, h->vtmp.as<double>(), (const PcgCtrl*)nullptr, 0);
        HIPCHK(h, hipMemcpyAsync(h->vcm.p, h->g.p, sizeof(double) * 6 * h->C, hipMemcpyDeviceToDevice, h->stream));
        CHK(schur_product_standalone(h, h->vtmp.as<double>()));          // leaves a valid z for case 5
    }
    struct EventPair {                    // destroyed on every exit path
        hipEvent_t a = nullptr, b = nullptr;
        ~EventPair() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    } ev;
    HIPCHK(h, hipEventCreate(&ev.a));
    HIPCHK(h, hipEventCreate(&ev.b));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipEventRecord(ev.a, h->stream));
    for (int k = 0; k < reps; ++k) {
        switch (which) {
            case 0: CHK((launch_resjac<true, true>(h, h->x, h->tab, &np))); break;
            case 1: CHK((launch_resjac<false, false>(h, h->x, h->tab, &np))); break;
            case 2: CHK(launch_normal_blocks(h, h->x, h->tab, h->rec)); break;
            case 3: CHK(schur_product_standalone(h, h->vtmp.as<double>())); break;
            case 4: CHK(launch_point_sweep(h, (h->lds_vec || h->sweep_rc || h->sweep_rc_g) ? h->vtmp.as<double>() : h->vcm.as<double>(), nullptr, 0)); break;
            case 5: CHK(launch_cam_schur<0>(h, h->vtmp.as<double>(), nullptr, 0)); break;
            case 6: CHK(launch_cam_schur<1>(h, nullptr, nullptr, 0)); break;
            case 7: CHK((launch_resjac<true, true>(h, h->x, h->tab, &np, nullptr, nullptr, /*blocks=*/false))); break;
            case 8:    // reduced right-hand side + Schur-diagonal blocks (without the 6x6 inverses)
                if (xcd_cam(h)) { CHK(launch_rhs_and_preconditioner(h)); break; }      // (wave-per-chunk form: with combine + inverses)
                hipLaunchKernelGGL((k_cam_rhs_diag<false>), dim3(h->n_chunks), dim3(kRhsThreads), 0, h->stream, cam_major(h),
                                   (const double*)h->tab, (const double*)h->rec, (const double*)vinv_ptr(h), h->K, (int)h->C,
                                   h->acc(), h->cam_partial.as<double>(), RhsPrecond{nullptr, nullptr, nullptr},
                                   h->use_rhsrec ? (const double*)h->rhsrec.as<double>() : (const double*)nullptr, CamExchange{});
                LAUNCHED(h);
                break;
            case 10:   // streaming-store ceiling: fill the Jacobian planes, 16 B per lane, one stream
                hipLaunchKernelGGL(k_fill16, dim3(h->n_cu * 2), dim3(1024), 0, h->stream, h->J.as<double>(),
                                   (int64_t)((h->f32 ? 3 : 6) * h->ld), 1.0);
                break;
            default: return fail(h, -1, "unknown kernel id %d", which);
        }
    }
    HIPCHK(h, hipEventRecord(ev.b, h->stream));
    HIPCHK(h, hipEventSynchronize(ev.b));
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, ev.a, ev.b));
    *avg_us = 1e3 * (double)ms / reps;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

static int solve_impl(sfmba_handle* h, const double* x_start, double* x_inout, const sfmba_options* opt_in, sfmba_result* out);

int sfmba_solve(sfmba_handle* h, double* x_inout, const sfmba_options* opt_in, sfmba_result* out) {
    return sfmba_solve_from(h, x_inout, x_inout, opt_in, out);
}

int sfmba_solve_from(sfmba_handle* h, const double* x0, double* x_out, const sfmba_options* opt_in, sfmba_result* out) {
    const int rc = solve_impl(h, x0, x_out, opt_in, out);
    if (rc != 0 && h) {
        // Leave the handle reusable: drain what was enqueued (speculative launches may still be in flight) and,
        // after a collective failure, unmap the peers -- sequence numbers no longer agree across ranks, so the
        // direct link is dead; the transport registered before it (RCCL / callback) serves later solves.
        const std::string msg = h->err;
        h->skip = nullptr; h->post = Mailbox{};
        if (h->stream) (void)hipStreamSynchronize(h->stream);
        if (h->copy_stream) (void)hipStreamSynchronize(h->copy_stream);
        if (rc == -5 && h->p2p.ready) {
            p2p_close_peers(h);
            if (h->p2p.words) (void)hipMemset(h->p2p.words, 0, 4 * sizeof(unsigned));
        }
        h->err = msg;
    }
    return rc;
}

static int solve_impl(sfmba_handle* h, const double* x_start, double* x_inout, const sfmba_options* opt_in, sfmba_result* out) {
    CHK(enter(h));
    CHK(check_ready(h, x_start));
    if (!x_inout) return fail(h, -1, "x_out is NULL");
    if (!out) return fail(h, -1, "result is NULL");
    sfmba_options opt;
    if (opt_in) opt = *opt_in; else sfmba_default_options(&opt);
    memset(out, 0, sizeof *out);
    h->solved = false;
    h->skip = nullptr; h->post = Mailbox{};
    h->pending_scale_sums = false;
    h->p2p.first_in_solve = true;
    if (h->dbg.p2p_delay_ms > 0) std::this_thread::sleep_for(std::chrono::milliseconds(h->dbg.p2p_delay_ms));
    const double t_begin = now_s();
    const int64_t C = h->C, P = h->P, n = h->n;
    const int64_t max_nfev = opt.max_nfev > 0 ? opt.max_nfev : 100 * (6 * C + 3 * P);
    const double m_total = 2.0 * (double)h->N_total;
    double* sc = h->scal();

    h->x = h->xa.as<double>(); h->x_new = h->xb.as<double>();
    h->tab = h->tabA.as<double>(); h->tab_new = h->tabB.as<double>();
    h->rec = h->recA.as<double>(); h->rec_new = h->recB.as<double>();
    CHK(upload_x(h, x_start));
    h->mirror_on = h->copy_stream != nullptr && n >= 2000000;           // (see the note at copy_stream)
    h->x_tag = 0; h->mirror_tag[0] = h->mirror_tag[1] = 0;
    if (h->mirror_on) {
        HIPCHK(h, h->mirror[0].ensure(sizeof(double) * n, 0));
        HIPCHK(h, h->mirror[1].ensure(sizeof(double) * n, 0));
    }
    HIPCHK(h, hipMemsetAsync(h->scal() + kGhPrevSlot, 0, 2 * sizeof(double), h->stream));   // forcing-term memory of k_prep
    const double t_dev0 = now_s();
    report_stall(h, "upload_x", t_dev0 - t_begin);

    // Host/device hand-offs per outer iteration: ONE read-back after the whole linear phase
    // (Cauchy product, Schur PCG, back-substitution, Gram/dot reductions are enqueued without the
    // host seeing intermediate values: the regularisation term is computed by k_prep on the device
    // and the PCG stops itself through its device-side control block) and ONE per trial step.
    struct EventList : std::vector<std::pair<hipEvent_t, hipEvent_t>> {     // destroyed on every exit path
        ~EventList() { for (auto& pr : *this) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); } }
    } evs;
    int np_cost = 0;                                                  // partial rows of the last K1 launch
    auto eval_jac = [&](const double* x, double* tab, bool table_ready, bool finish = true) -> int {   // K0 + K1, sum r^2 -> scalar 0
        if (!table_ready) CHK(launch_cam_table(h, x, tab, h->rec));
        int& np = np_cost;
        if (opt.profile) {
            hipEvent_t a, b;
            // timing-only events (hip_runtime_api.h: hipEventDisableSystemFence): no system-scope cache write-back and
            // invalidation when they are recorded, which otherwise lands inside the measured interval (+1.3 us)
            // and delays the launches around the kernel
            HIPCHK(h, hipEventCreateWithFlags(&a, hipEventDisableSystemFence));
            if (hipEventCreateWithFlags(&b, hipEventDisableSystemFence) != hipSuccess) {
                (void)hipEventDestroy(a);
                return fail(h, -3, "hipEventCreateWithFlags failed");
            }
            evs.emplace_back(a, b);
            CHK((launch_resjac<true, true>(h, x, tab, &np, a, b)));
        } else {
            CHK((launch_resjac<true, true>(h, x, tab, &np)));
        }
        if (finish) CHK(launch_finish(h, h->part.as<double>(), np, 1, 0));
        return 0;
    };
    auto linearise = [&](int first) -> int {      // normal blocks, scale, gradient, q0..q4 at h->x
        CHK(launch_normal_blocks(h, h->x, h->tab, h->rec));
        CHK(launch_update_scale(h, first));
        CHK(exchange_linearise(h));
        return 0;
    };

    // f0, J0 (least_squares.py:838, 903-912)
    CHK(eval_jac(h->x, h->tab, false));
    CHK(exchange(h, sc, 1, 0));                      // sum r^2
    CHK(linearise(1));
    CHK(fetch_scalars(h));
    double cost = 0.5 * h->h_scal[0];
    if (!std::isfinite(cost)) return fail(h, -2, "Residuals are not finite in the initial point.");
    out->cost0 = cost;
    out->rmse0 = std::sqrt(2.0 * cost / m_total);
    int64_t nfev = 1, njev = 1, iteration = 0, pcg_total = 0;
    double Delta = std::sqrt(qsum(h, 2));                       // |x0 * scale_inv|, trf.py:428
    if (Delta == 0.0) Delta = 1.0;
    int status = -1;
    double step_norm = 0.0, actual_reduction = 0.0, g_norm = 0.0, reg_term = 0.0;
    bool have_red = false;
    int pcg_guess = h->pcg_hint;                                // iterations to enqueue without reading back
    std::vector<int> pcg_hist_new;
    bool guess_exact = false;                                   // pcg_guess comes from the record: no spare launch
    // The record is trusted count for count only on the problem it was recorded on; on a neighbouring problem (the
    // reference's growing reconstruction: a camera or a few hundred observations more) it is a guess with a spare
    // launch; on an unrelated problem it is dropped together with the running hint.
    // (content, not only sizes: the problem arrays' generation and a checksum of the start vector -- a different start on
    // the same problem, or another problem of the same sizes, is a "neighbouring" one and gets the spare launch)
    unsigned long long sig = h->problem_gen * 0x9E3779B97F4A7C15ull;
    {
        const int64_t stride = std::max<int64_t>(1, n / 509);
        for (int64_t k = 0; k < n; k += stride) {
            unsigned long long bits;
            memcpy(&bits, x_start + k, sizeof bits);
            sig = (sig ^ bits) * 0x100000001B3ull;
        }
    }
    const bool hist_same = h->hist_C == C && h->hist_P == P && h->hist_N == h->N && h->hist_sig == sig;
    const bool hist_near = hist_same || (h->hist_N > 0 && std::llabs(h->hist_C - C) <= 1 && 2 * h->N >= h->hist_N && h->N <= 2 * h->hist_N);
    if (!hist_near) { h->pcg_hist.clear(); h->pcg_hint = 0; pcg_guess = 0; }
    if (!h->pcg_hist.empty() && h->pcg_hist[0] > 0) { pcg_guess = h->pcg_hist[0]; guess_exact = hist_same; }
    int64_t pcg_breakdowns = 0;
    bool nb_valid = true;                                       // V, g_p, [U|g_c] (and, single-buffered, J and r) belong to h->x
    // Single-buffered Jacobian: the trial point is evaluated into the SAME J / r buffers the sweeps of
    // this iteration have just read, so K1's stores land on lines that are still resident in the
    // Infinity Cache instead of cold ones.  A rejected step leaves J / r describing the rejected point;
    // nothing uses them before the next trial overwrites them, except the rare exits handled below.
    const bool pcg_debug = h->dbg.trace_pcg != 0;
    const int pcg_bias = h->dbg.pcg_guess_bias;                 // test hook (sfmba_debug_option)
    if (opt.verbose >= 2) print_header(h);

    for (;;) {                                                  // trf.py:450
        if (!nb_valid) {                                        // a rejected trial overwrote the blocks and no
            CHK(eval_jac(h->x, h->tab, true));         // step was accepted afterwards
            CHK(launch_normal_blocks(h, h->x, h->tab, h->rec));                       // (nfev limit)
            nb_valid = true;
        }
        // ---- enqueue the whole linear phase ---------------------------------------------------
        // Final sums of per-workgroup partials are not launches of their own where a neighbour can carry
        // them: k_update_scale's ride with k_jdot; on a single rank k_jdot's are repeated by every workgroup
        // of k_prep and k_backsub's are done by k_tr_step (with several
        // ranks the all-reduce has to sit between producer and consumer, so those two stay launches).
        const bool one_rank = !multi_rank(h);
        int np = 0;
        const bool scale_sums_rode = h->pending_scale_sums;
        CHK(launch_jdot(h, &np));                               // t1 = J D^2 g, G11 = |t1|^2
        if (!one_rank && h->p2p.ready) {
            // direct path: the collective's own workgroup first sums k_jdot's partials into slot 1, then reduces
            // slots 1..12 over the ranks when an accepted step left fresh q1..q4 (8..11, sums) and max|g| (12, a
            // maximum by the mask), G11 alone otherwise.  Slots 2..7 (G12, G22, q5..q8 of the previous iteration)
            // are summed once more along the way; nothing reads them before k_backsub's sums rewrite them.
            Piggyback pb{h->partB(), sc, FinishJob{}, 1, 1, 0};
            pb.job.row0[0] = 0; pb.job.nrows[0] = np;
            for (int k = 0; k < kNQ; ++k) { pb.job.slot[0][k] = 1 + k; pb.job.slot[1][k] = -1; }
            static_assert(kMaxSlot == 12, "mask below assumes the maximum sits at the end of the run");
            CHK(p2p_allreduce(h, sc + 1, scale_sums_rode ? kMaxSlot : 1, 0, nullptr,
                              scale_sums_rode ? 1ull << (kMaxSlot - 1) : 0ull, &pb));
            ++h->n_collectives;
        } else if (!one_rank) {
            CHK(launch_finish(h, h->partB(), np, 1, 1));
            if (scale_sums_rode) CHK(exchange_linearise(h));        // q1..q4, max|g| of the accepted point
            CHK(exchange(h, sc + 1, 1, 0));
        }
        {   // regularisation (trf.py:471-475), Vinv/e per point, Dc/Minv per camera, acc0 = 0: one launch
            const int bc = (int)((C + 63) / 64), bp = (int)((P + 63) / 64);
            hipLaunchKernelGGL(k_prep, dim3(bc + bp), dim3(64), 0, h->stream, sc, Delta, opt.reg_min, h->Ugc(),
                               h->V.as<double>(), h->gp.as<double>(), h->si.as<double>(), (int)C, (int)P, bc,
                               h->Dc.as<double>(), (h->dbg.precond == 0 ? h->Minv.as<double>() : (double*)nullptr),
                               vinv_ptr(h), h->rec + 3,
                               one_rank ? (const double*)h->partB() : (const double*)nullptr, np, opt.pcg_tol,
                               std::max(opt.pcg_tol, opt.pcg_tol_max), (const double*)(h->x + 6 * C),
                               (h->use_rhsrec && !(h->dense && one_rank)) ? h->rhsrec.as<double>() : (double*)nullptr,   // (read by k_cam_rhs_diag only)
                               (h->dense && one_rank) ? MixedPrep{nullptr, nullptr, nullptr, nullptr, Origin{0.0, 0.0, 0.0}} : mixed_prep(h, h->tab));
            LAUNCHED(h);
        }
        const bool dense = h->dense && one_rank;               // (sharded: the block pairs would need their own all-reduce)
        if (!dense) CHK(launch_rhs_and_preconditioner(h));      // reduced rhs term -> acc, preconditioner blocks
                                                                // (few cameras: both inside launch_dense_solve)
        if (!dense) CHK(pcg_start(h, opt));                     // replaces lsmr, trf.py:477-480
        PcgCtrl hc{};
        if (dense) {
            // the same PCG inside one workgroup; an iteration there costs a twentieth of the two launches of the
            // implicit product, so the forcing term is a decade tighter (cfg2: 5 outer iterations instead of 7)
            CHK(launch_dense_solve(h, kDenseTolFactor * opt.pcg_tol, pcg_max_iters(h, opt), /*rhs=*/true));
        } else if (pcg_guess > 0) {
            // speculative: no read-back; surplus launches are no-ops.  Fused launches apply the update of
            // iteration k in launch k + 1, so k iterations need k + 1 launches; one spare either way.
            CHK(pcg_enqueue(h, pcg_guess + (h->pcg_fused ? 1 : 0) + (guess_exact ? 0 : 1), true));
        } else {
            CHK(pcg_finish_polling(h, opt, &hc));
        }
        // back-substitution + model products (all in k_backsub); `for_device_step`: k_tr_step follows and
        // (single rank) sums k_backsub's partial rows itself
        int np_tail = 0;
        auto tail = [&](bool for_device_step) -> int {
            CHK(launch_backsub(h, &np));
            np_tail = np;
            if (!one_rank && h->p2p.ready) {                    // the sums run inside the collective's workgroup
                const Piggyback pb = backsub_rider(h, np);
                CHK(p2p_allreduce(h, sc + 2, 6, 0, nullptr, 0, &pb));
                ++h->n_collectives;
                return 0;
            }
            if (!(for_device_step && one_rank)) CHK(launch_finish_backsub(h, np));
            CHK(exchange_tail(h));
            return 0;
        };
        CHK(tail(true));

        Mailbox trial_post{};
        bool trial_post_pending = false;
        // ---- the first trial step is decided ON THE DEVICE (k_tr_step) and evaluated right away -------
        // so that an outer iteration hands control to the host ONCE, after the trial cost is known.
        auto enqueue_trial = [&](const double* coef_dev, double c1, double c2) -> int {
            const int bc = (int)((C + 255) / 256);
            hipLaunchKernelGGL(k_step_table, dim3(bc + grid_1d(3 * P, 256, 2048)), dim3(256), 0, h->stream, h->x,
                               h->sg.as<double>(), h->p.as<double>(), c1, c2, coef_dev, (int)C, n, bc, h->x_new,
                               h->tab_new, h->rec_new, h->skip);
            LAUNCHED(h);
            if (h->mirror_on) {                                 // x_new to its host mirror, beside the evaluation below
                const int w = h->x_new == h->xa.as<double>() ? 0 : 1;
                HIPCHK(h, hipEventRecord(h->ev_written, h->stream));
                HIPCHK(h, hipStreamWaitEvent(h->copy_stream, h->ev_written, 0));
                HIPCHK(h, hipMemcpyAsync(h->mirror[w].p, h->x_new, sizeof(double) * n, hipMemcpyDeviceToHost, h->copy_stream));
                HIPCHK(h, hipEventRecord(h->ev_copied[w], h->copy_stream));
                h->mirror_tag[w] = ++h->copy_count;
            }
            // the trial point is evaluated WITH its Jacobian, into the same buffers: when
            // the step is accepted (the common case) nothing has to be recomputed
            // The cost reduction that ends the evaluation also posts the hand-off (scalars + PCG control
            // block) into the host mailbox; with several ranks the post follows the all-reduce of the cost.
            const Mailbox mb{h->mbox_dev, sc, h->ctrl.as<PcgCtrl>() + (h->pcg_L & 1), ++h->mbox_seq, p2p_error_word(h)};
            const bool ranks = multi_rank(h);
            if (ranks && h->p2p.ready) {
                // direct path: ONE single-workgroup launch sums K1's cost partials, reduces the cost over the
                // ranks and posts the hand-off (or only posts, when k_tr_step cancelled the trial)
                CHK(eval_jac(h->x_new, h->tab_new, true, /*finish=*/false));
                Piggyback pb{h->part.as<double>(), sc, FinishJob{}, 1, 1, 0};
                pb.job.row0[0] = 0; pb.job.nrows[0] = np_cost;
                for (int k = 0; k < kNQ; ++k) { pb.job.slot[0][k] = k; pb.job.slot[1][k] = -1; }
                CHK(p2p_allreduce(h, sc, 1, 0, nullptr, 0, &pb, &mb, h->skip));
                ++h->n_collectives;
                return 0;
            }
            if (!ranks && h->dbg.cost_rider == 0) {
                h->post = mb;
                const int rc = eval_jac(h->x_new, h->tab_new, true);
                h->post = Mailbox{};
                return rc;
            }
            if (!ranks) {
                // single rank: the cost sum and the post ride with the launch that builds the trial point's blocks
                // (handoff below), one launch less per iteration
                CHK(eval_jac(h->x_new, h->tab_new, true, /*finish=*/false));
                trial_post = mb;
                trial_post_pending = true;
                return 0;
            }
            CHK(eval_jac(h->x_new, h->tab_new, true));
            CHK(exchange(h, sc, 1, 0));
            hipLaunchKernelGGL(k_post, dim3(1), dim3(64), 0, h->stream, mb);
            LAUNCHED(h);
            return 0;
        };
        auto handoff = [&](bool with_ctrl) -> int {
            // While the host waits, the GPU already builds the normal-equation blocks of the trial point
            // (speculating on acceptance, the common case).  They overwrite V / g_p / [U|g_c], which a
            // rejected step does not need: a retry only re-solves the 2-D model (host scalars) and
            // re-applies k_step_table to x, D^2 g and p, all untouched.
            if (trial_post_pending) CHK(launch_normal_blocks(h, h->x_new, h->tab_new, h->rec_new, np_cost, trial_post));
            else CHK(launch_normal_blocks(h, h->x_new, h->tab_new, h->rec_new));
            trial_post_pending = false;
            nb_valid = false;
            CHK(wait_mailbox(h, h->mbox_seq));
            memcpy(h->h_scal, h->mbox, sizeof(double) * kScalSlots);
            if (with_ctrl) memcpy(&hc, h->mbox + kMboxCtrl, sizeof hc);
            return 0;
        };
        const bool speculated = pcg_guess > 0 || dense;         // dense: the control block always says "finished"
        const int pcg_enqueued = pcg_guess > 0 ? pcg_guess + (h->pcg_fused ? 1 : 0) + (guess_exact ? 0 : 1) : 0;
        bool missed = false;
        for (bool speculative = speculated;;) {
            hipLaunchKernelGGL(k_tr_step, dim3(1), dim3(one_rank ? 1024 : 64), 0, h->stream, sc, Delta,
                               speculative ? (const PcgCtrl*)(h->ctrl.as<PcgCtrl>() + (h->pcg_L & 1)) : (const PcgCtrl*)nullptr,
                               one_rank ? backsub_rider(h, np_tail) : Piggyback{});
            LAUNCHED(h);
            h->skip = sc + 30;                                  // k_tr_step's verdict gates every launch below
            int rc_trial = enqueue_trial(sc + 25, 0.0, 0.0);
            if (rc_trial == 0) rc_trial = handoff(speculative); // THE hand-off of this iteration
            h->skip = nullptr;
            CHK(rc_trial);
            if (!(speculative && hc.done == 0)) break;
            // The PCG needed more iterations than were enqueued.  k_tr_step saw that on the device and
            // cancelled the trial launches, so J, r and the normal blocks still describe x: finish the
            // PCG (host-polled), redo the tail and decide the step on the device again -- the arithmetic of
            // an iteration does not depend on whether its guess sufficed.
            missed = true;
            nb_valid = true;
            if (opt.profile && !evs.empty()) {                  // the cancelled launch is not a K1 timing sample
                (void)hipEventDestroy(evs.back().first);
                (void)hipEventDestroy(evs.back().second);
                evs.pop_back();
            }
            CHK(pcg_finish_polling(h, opt, &hc));
            CHK(tail(true));
            speculative = false;
        }
        bool first_trial_ready = true;
        // hc.done == 3: the CG recurrences lost positive definiteness (rounding, typically on a converged
        // system whose right-hand side is noise).  The iterate of the last good step is kept -- it is
        // zero when the very first step failed, in which case the 2-D model below degenerates to the
        // steepest-descent line, exactly scipy's fallback when gn_h adds nothing to span(g_h).
        if (hc.done == 3) ++pcg_breakdowns;

        // ---- loop head of trf.py:450-459, evaluated now that the scalars are on the host -----------
        g_norm = std::max(h->h_scal[kMaxSlot], h->h_scal[kCamSlot + 0]);
        if (g_norm < opt.gtol) status = 1;
        if (opt.verbose >= 2) print_iter(h, iteration, nfev, cost, have_red, actual_reduction, step_norm, g_norm);
        if (status != -1 || nfev >= max_nfev || (opt.max_iter > 0 && iteration >= opt.max_iter)) break;
        pcg_total += hc.iters;
        if (pcg_debug && dense)
            fprintf(stderr, "sfmba: iteration %lld: PCG in LDS, %d iterations, outcome %d, reg %.3e\n", (long long)iteration,
                    hc.iters, hc.done, h->h_scal[kRegSlot]);
        else if (pcg_debug)
            fprintf(stderr, "sfmba: iteration %lld: PCG enqueued %d, needed %d%s\n", (long long)iteration,
                    pcg_enqueued, hc.iters, missed ? " (miss)" : "");
        // Next guess: the largest recent count, forgotten by one iteration per outer iteration.  A surplus
        // iteration costs two empty launches (~10 us); a miss costs a hand-off per polled batch.
        h->pcg_hint = std::max(hc.iters, h->pcg_hint - 1);
        pcg_hist_new.push_back(hc.iters);
        {
            const int rec = (iteration + 1 < (int64_t)h->pcg_hist.size()) ? h->pcg_hist[iteration + 1] : 0;
            guess_exact = rec > 0 && pcg_bias == 0 && hist_same;
            pcg_guess = std::max(1, (guess_exact ? rec : h->pcg_hint) + pcg_bias);
        }
        reg_term = h->h_scal[kRegSlot];

        const double x_norm = std::sqrt(qsum(h, 3));
        const TrModel model = tr_build_model(h->h_scal[1], h->h_scal[2], h->h_scal[3], qsum(h, 1), qsum(h, 5),
                                             qsum(h, 6), qsum(h, 4), qsum(h, 7), qsum(h, 8));

        actual_reduction = -1.0;
        double cost_new = cost;
        while (actual_reduction <= 0.0 && nfev < max_nfev) {    // trf.py:488
            TrStep st;
            if (first_trial_ready) {                            // decided and evaluated on the device above
                st.c1 = h->h_scal[25]; st.c2 = h->h_scal[26]; st.predicted = h->h_scal[27];
                st.step_h_norm = h->h_scal[28]; st.step_norm = h->h_scal[29];
                first_trial_ready = false;
            } else {                                            // retry after a rejected step (or fallback)
                st = tr_solve_step(model, Delta);
                CHK(enqueue_trial(nullptr, st.c1, st.c2));
                CHK(handoff(false));
            }
            ++nfev;
            cost_new = 0.5 * h->h_scal[0];
            if (!std::isfinite(cost_new)) {                     // trf.py:504-506
                Delta = 0.25 * st.step_h_norm;
                continue;
            }
            actual_reduction = cost - cost_new;
            double ratio;
            const double Delta_new = update_tr_radius(Delta, actual_reduction, st.predicted, st.step_h_norm,
                                                      st.step_h_norm > 0.95 * Delta, &ratio);
            step_norm = st.step_norm;
            const int term = check_termination(actual_reduction, cost, step_norm, x_norm, ratio, opt.ftol, opt.xtol);
            if (term != 0) { status = term; break; }
            Delta = Delta_new;
        }
        have_red = true;
        if (actual_reduction > 0.0) {                           // trf.py:528
            std::swap(h->x, h->x_new);
            std::swap(h->tab, h->tab_new);
            std::swap(h->rec, h->rec_new);
            h->x_tag = h->mirror_tag[h->x == h->xa.as<double>() ? 0 : 1];      // (0 when no mirror copy was made)
            nb_valid = true;                                    // J, f and the normal blocks of the accepted
                                                                // point are already there / in flight
            cost = cost_new;
            ++njev;
            CHK(launch_update_scale(h, 0, /*defer=*/true));     // its final sums ride with the next k_jdot and
                                                                // are read with the next hand-off
        } else {
            step_norm = 0.0;
            actual_reduction = 0.0;
        }
        ++iteration;
        if (status != -1) {                                     // terminated inside the step loop
            if (h->pending_scale_sums) {                        // no k_jdot follows
                CHK(flush_scale_sums(h));
                CHK(exchange_linearise(h));
            }
            CHK(fetch_scalars(h));
            g_norm = std::max(h->h_scal[kMaxSlot], h->h_scal[kCamSlot + 0]);
            if (opt.verbose >= 2) print_iter(h, iteration, nfev, cost, have_red, actual_reduction, step_norm, g_norm);
            break;
        }
    }
    if (status == -1) status = 0;
    if (!nb_valid) {                                            // last trial was not accepted: result.fun is f(x)
        int np = 0;
        CHK((launch_resjac<false, true>(h, h->x, h->tab, &np)));
    }

    CHK(ensure_h_x(h));
    const double t_dl0 = now_s();
    const int xw = h->x == h->xa.as<double>() ? 0 : 1;
    const bool mirrored = h->mirror_on && h->x_tag != 0 && h->mirror_tag[xw] == h->x_tag;     // x is on the host already
    const size_t xbytes = sizeof(double) * (size_t)n;
    if (!mirrored) HIPCHK(h, hipMemcpyAsync(h->h_x, h->x, xbytes, hipMemcpyDeviceToHost, h->stream));
    if (h->p2p.ready)                                           // did a direct all-reduce give up waiting for a peer?
        HIPCHK(h, hipMemcpyAsync(h->h_scal + 62, h->p2p.words + 1, sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
    CHK(wait_stream(h));
    if (mirrored) HIPCHK(h, hipEventSynchronize(h->ev_copied[xw]));
    const double t_dl1 = now_s();
    if (h->p2p.ready) {
        unsigned err = 0;
        memcpy(&err, h->h_scal + 62, sizeof err);
        if (err != 0) return p2p_timed_out(h, err);
    }
    staging_copy(h, x_inout, mirrored ? h->mirror[xw].p : (const void*)h->h_x, xbytes);
    if (h->mirror_on) HIPCHK(h, hipStreamSynchronize(h->copy_stream));     // (a copy of a rejected last trial may still run)
    const double t_end = now_s();
    if (h->dbg.trace_timing)
        fprintf(stderr, "sfmba: solve  %.3f ms total, upload %.1f us, result: copy + wait %.1f us, staging copy %.1f us\n", 1e3 * (t_end - t_begin),
                1e6 * (t_dev0 - t_begin), 1e6 * (t_dl1 - t_dl0), 1e6 * (t_end - t_dl1));
    if (!evs.empty()) {
        double tot = 0.0;
        for (auto& pr : evs) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) tot += ms;
        }
        out->resjac_avg_us = 1e3 * tot / (double)evs.size();
        out->resjac_launches = (int64_t)evs.size();
    }
    h->pcg_hist.swap(pcg_hist_new);
    h->hist_C = C; h->hist_P = P; h->hist_N = h->N; h->hist_sig = sig;
    out->cost = cost;
    out->optimality = g_norm;
    out->rmse = std::sqrt(2.0 * cost / m_total);
    out->nfev = nfev; out->njev = njev; out->iterations = iteration; out->pcg_iterations = pcg_total;
    out->status = status;
    out->seconds_total = t_end - t_begin;
    out->seconds_device = t_end - t_dev0;
    out->last_step_norm = step_norm;
    out->last_reg = reg_term;
    out->reserved = (int32_t)pcg_breakdowns;
    h->solved = true;
    return 0;
}

int sfmba_get_fun_grad(sfmba_handle* h, double* fun_out, double* grad_out) {
    CHK(enter(h));
    if (!h->have_problem || !h->solved) return fail(h, -1, "no completed sfmba_solve on this handle");
    if (fun_out) CHK(download_residuals(h, fun_out));
    if (grad_out) {
        CHK(ensure_h_x(h));
        HIPCHK(h, hipMemcpyAsync(h->h_x, h->g.p, sizeof(double) * h->n, hipMemcpyDeviceToHost, h->stream));
        CHK(wait_stream(h));
        memcpy(grad_out, h->h_x, sizeof(double) * h->n);
    }
    return 0;
}

// Host-only helper exported for the CPU test-suite (no GPU needed): the 2-D trust-region solve.
int sfmba_tr2d_solve(const double* B3, const double* g2, double Delta, double* p2) {
    return solve_trust_region_2d(B3, g2, Delta, p2) ? 1 : 0;
}

}  // extern "C"
