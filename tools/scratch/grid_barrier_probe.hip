// Scratch probe (not part of the library): what does a device-wide barrier of resident workgroups cost on MI355X?
//
// Round 2 measured ONE form -- a central counter polled by every workgroup with two __threadfence() -- at 22-25 us for
// 256 x 1024 threads and concluded against a persistent PCG kernel.  /opt/skills/guides/MI355X_MICROARCH.md ("Persistent
// kernels: synchronisation and hand-off price list") prices that form at 7.4-9.3 us and an XCD-hierarchical one at
// 4.1 us (256 workgroups).  This probe measures both, with a visibility check, over
//     G in {8, 32, 64, 256} workgroups x {256, 1024} threads, with and without 32 KB dirtied per workgroup and round.
//
//   central : release fence (agent) -> atomicAdd on one counter -> relaxed poll -> acquire fence (agent), by lane 0
//   xcd     : per-XCC arrival counter (the workgroups of one XCD share an L2: their stores are already there once
//             vmcnt == 0, so the arrival needs no L2 write-back) -> the LAST arriver of an XCC is its leader: release
//             fence (agent; one L2 write-back per XCD instead of one per workgroup) -> top counter -> poll -> acquire
//             fence -> per-XCC generation word; the other workgroups poll that word and take an acquire fence.
// The XCC of a workgroup is read from the hardware (s_getreg XCC_ID), not assumed from blockIdx.
// Every wait is bounded (spin cap -> error word), so the grid always drains.
//
//   hipcc --offload-arch=gfx950 -O3 -o grid_barrier_probe grid_barrier_probe.hip && ./grid_barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int kLine = 32;                       // unsigned words per 128-byte line
struct Bar {
    unsigned central[kLine];
    unsigned top[kLine];
    unsigned xcc_cnt[8][kLine];
    unsigned xcc_gen[8][kLine];
    unsigned members[8][kLine];
    unsigned err[kLine];
};

__device__ __forceinline__ unsigned ld_relaxed(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool spin_until(const unsigned* p, unsigned target, unsigned* err) {
    unsigned spins = 0;
    while ((int)(ld_relaxed(p) - target) < 0) {
        if (++spins > (1u << 22)) { atomicExch(err, 1u); return false; }
        __builtin_amdgcn_s_sleep(1);
    }
    return true;
}

__device__ __forceinline__ bool barrier_central(Bar* b, unsigned round, unsigned G) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(b->central, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = spin_until(b->central, round * G, b->err);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return ok;
}

// n_mine = number of workgroups on this workgroup's XCC, n_xcc = number of XCCs that have workgroups
__device__ __forceinline__ bool barrier_xcd(Bar* b, unsigned round, int xcc, unsigned n_mine, unsigned n_xcc) {
    __syncthreads();                            // (every wave's stores have been issued)
    bool ok = true;
    if (threadIdx.x < 64) {                     // wave 0
        if (threadIdx.x == 0) {
            // the workgroup's stores are in the XCD's L2 once they have completed: a workgroup-scope release is
            // s_waitcnt vmcnt(0) without an L2 write-back
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0);
            const unsigned prev = __hip_atomic_fetch_add(b->xcc_cnt[xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev + 1 == round * n_mine) {   // last of this XCC: the leader of this round
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");             // L2 write-back, once per XCD
                __hip_atomic_fetch_add(b->top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = spin_until(b->top, round * n_xcc, b->err);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __hip_atomic_store(b->xcc_gen[xcc], round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                ok = spin_until(b->xcc_gen[xcc], round, b->err);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
        }
    }
    __syncthreads();
    return ok;
}

template <int MODE>                             // 0 central, 1 xcd
__global__ void k_probe(Bar* bar, double* buf, unsigned* seen, size_t per_wg, int rounds, int touch) {
    __shared__ int s_xcc;
    __shared__ unsigned s_n_mine, s_n_xcc;
    const unsigned G = gridDim.x;
    if (threadIdx.x == 0) {
        const int xcc = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7);      // hwreg(HW_REG_XCC_ID, 0, 4)
        s_xcc = xcc;
        __hip_atomic_fetch_add(bar->members[xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // membership is final after one central barrier (round 1 of the central counter in both modes)
    if (!barrier_central(bar, 1u, G)) return;
    if (threadIdx.x == 0) {
        unsigned nx = 0;
        for (int x = 0; x < 8; ++x) nx += ld_relaxed(bar->members[x]) != 0u;
        s_n_mine = ld_relaxed(bar->members[s_xcc]);
        s_n_xcc = nx;
    }
    __syncthreads();
    const int xcc = s_xcc;
    const unsigned n_mine = s_n_mine, n_xcc = s_n_xcc;
    double* mine = buf + (size_t)blockIdx.x * per_wg;
    unsigned bad = 0;
    for (int r = 1; r <= rounds; ++r) {
        if (touch) for (size_t i = threadIdx.x; i < per_wg; i += blockDim.x) mine[i] = (double)r + (double)i;
        if (threadIdx.x == 0) mine[0] = (double)r;            // the word the others check
        const bool ok = MODE == 0 ? barrier_central(bar, (unsigned)r + 1u, G) : barrier_xcd(bar, (unsigned)r, xcc, n_mine, n_xcc);
        if (!ok) return;
        // visibility: every workgroup reads the word of a workgroup half the grid away (another XCD when G >= 2); its
        // owner may already have written the next round's value, an older one is a stale read
        if (threadIdx.x == 0) {
            const unsigned other = (blockIdx.x + G / 2 + 1) % G;
            const double v = __hip_atomic_load(buf + (size_t)other * per_wg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (v < (double)r) ++bad;
        }
    }
    if (threadIdx.x == 0 && bad) atomicAdd(seen, bad);
}


// ---- one XCD only -------------------------------------------------------------------------------------------------
// Workgroups are dealt to the XCDs round-robin (blockIdx % 8), so the workgroups with blockIdx % 8 == 0 of an 8 G-wide
// launch share ONE L2.  Among them a barrier needs no L2 write-back and no L2 invalidation: stores are in the L2 once
// vmcnt == 0 (the vector L1 is write-through), and the reader only has to drop its own L1 (buffer_inv sc0).
// MODE 2: that barrier.  MODE 3: the same workgroups with the agent-scope central barrier (what the data pay for
// an L2 write-back + invalidate).  After the barrier every workgroup reads the WHOLE 32 KB another one wrote, with
// plain loads, and checks every value.
__device__ __forceinline__ bool barrier_one_xcd(Bar* b, unsigned round, unsigned G) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(b->central, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = spin_until(b->central, round * G, b->err);
    }
    __syncthreads();
    asm volatile("buffer_inv sc0" ::: "memory");
    return ok;
}

template <int MODE>                             // 2 one-XCD barrier, 3 agent-scope central barrier on the same workgroups
__global__ void k_probe_one_xcd(Bar* bar, double* buf, unsigned* seen, size_t per_wg, int rounds, unsigned* xcc_seen) {
    if ((blockIdx.x & 7) != 0) return;
    const unsigned G = gridDim.x / 8, me = blockIdx.x / 8;
    if (threadIdx.x == 0) xcc_seen[me] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;
    double* mine = buf + (size_t)me * per_wg;
    unsigned bad = 0;
    for (int r = 1; r <= rounds; ++r) {
        for (size_t i = threadIdx.x; i < per_wg; i += blockDim.x) mine[i] = (double)r + (double)i;
        const bool ok = MODE == 2 ? barrier_one_xcd(bar, 2u * r - 1u, G) : barrier_central(bar, 2u * r - 1u, G);
        if (!ok) return;
        const double* other = buf + (size_t)((me + G / 2 + 1) % G) * per_wg;
        for (size_t i = threadIdx.x; i < per_wg; i += blockDim.x) bad += other[i] != (double)r + (double)i;
        // nobody may overwrite what a neighbour is still reading
        if (!(MODE == 2 ? barrier_one_xcd(bar, 2u * r, G) : barrier_central(bar, 2u * r, G))) return;
    }
    if (bad) atomicAdd(seen, bad);
}

int main(int argc, char**) {
    hipSetDevice(0);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("device: %s, %d CUs\n", p.name, p.multiProcessorCount);
    Bar* bar; hipMalloc(&bar, sizeof(Bar));
    unsigned* seen; hipMalloc(&seen, 4);
    const size_t per_wg = 4096;                 // 32 KiB per workgroup and round
    double* buf; hipMalloc(&buf, sizeof(double) * per_wg * 256); hipMemset(buf, 0, sizeof(double) * per_wg * 256);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    printf("%-8s %5s %8s %6s %12s %8s %6s\n", "barrier", "G", "threads", "touch", "us/barrier", "stale", "err");
    for (int mode = argc > 1 ? 2 : 0; mode < 2; ++mode)      // any argument: the one-XCD part only
        for (int G : {8, 32, 64, 256})
            for (int threads : {256, 1024})
                for (int touch = 0; touch < 2; ++touch) {
                    float t[2] = {0, 0};
                    unsigned stale = 0, err = 0;
                    const int rr[2] = {1, 401};
                    for (int k = 0; k < 2; ++k) {
                        hipMemset(bar, 0, sizeof(Bar)); hipMemset(seen, 0, 4);
                        hipDeviceSynchronize();
                        hipEventRecord(a);
                        if (mode == 0) hipLaunchKernelGGL(k_probe<0>, dim3(G), dim3(threads), 0, 0, bar, buf, seen, per_wg, rr[k], touch);
                        else hipLaunchKernelGGL(k_probe<1>, dim3(G), dim3(threads), 0, 0, bar, buf, seen, per_wg, rr[k], touch);
                        hipEventRecord(b); hipEventSynchronize(b);
                        hipEventElapsedTime(&t[k], a, b);
                        Bar hb; hipMemcpy(&hb, bar, sizeof hb, hipMemcpyDeviceToHost);
                        unsigned s; hipMemcpy(&s, seen, 4, hipMemcpyDeviceToHost);
                        stale += s; err += hb.err[0];
                    }
                    const double per_round = 1e3 * (t[1] - t[0]) / (rr[1] - rr[0]);
                    printf("%-8s %5d %8d %6d %12.2f %8u %6u\n", mode == 0 ? "central" : "xcd", G, threads, touch,
                           per_round, stale, err);
                    fflush(stdout);
                }
    unsigned* xcc_seen; hipMalloc(&xcc_seen, 4 * 64);
    printf("%-8s %5s %8s %14s %8s %6s  %s\n", "barrier", "G", "threads", "us/round(2 bar)", "bad", "err", "XCC ids of the participants");
    for (int mode = 2; mode < 4; ++mode)
        for (int G : {8, 16, 32})
            for (int threads : {256, 512, 1024}) {
                float t[2] = {0, 0};
                unsigned stale = 0, err = 0;
                const int rr[2] = {1, 201};
                for (int k = 0; k < 2; ++k) {
                    hipMemset(bar, 0, sizeof(Bar)); hipMemset(seen, 0, 4); hipMemset(xcc_seen, 0xff, 4 * 64);
                    hipDeviceSynchronize();
                    hipEventRecord(a);
                    if (mode == 2) hipLaunchKernelGGL(k_probe_one_xcd<2>, dim3(8 * G), dim3(threads), 0, 0, bar, buf, seen, per_wg, rr[k], xcc_seen);
                    else hipLaunchKernelGGL(k_probe_one_xcd<3>, dim3(8 * G), dim3(threads), 0, 0, bar, buf, seen, per_wg, rr[k], xcc_seen);
                    hipEventRecord(b); hipEventSynchronize(b);
                    hipEventElapsedTime(&t[k], a, b);
                    Bar hb; hipMemcpy(&hb, bar, sizeof hb, hipMemcpyDeviceToHost);
                    unsigned s; hipMemcpy(&s, seen, 4, hipMemcpyDeviceToHost);
                    stale += s; err += hb.err[0];
                }
                unsigned xs[64]; hipMemcpy(xs, xcc_seen, 4 * 64, hipMemcpyDeviceToHost);
                unsigned mask = 0;
                for (int i = 0; i < G; ++i) mask |= 1u << (xs[i] & 7);
                printf("%-8s %5d %8d %14.2f %8u %6u  mask 0x%02x\n", mode == 2 ? "one-xcd" : "agent", G, threads,
                       1e3 * (t[1] - t[0]) / (rr[1] - rr[0]), stale, err, mask);
                fflush(stdout);
            }
    return 0;
}
