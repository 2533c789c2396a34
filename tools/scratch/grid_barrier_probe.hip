// Scratch probe (not part of the library): what does a device-wide barrier of resident workgroups cost on MI355X?
// Measured (round 2): 22.2 us per barrier with relaxed polling, 24.5 us with acquire loads in the poll, with or
// without 8 MiB of dirty lines per round -- against ~4.5 us for a kernel boundary.
// 256 workgroups x 1024 threads with 144 KiB of LDS each (the shape of pass A), every workgroup dirties some global
// lines between barriers, as pass A / pass B of a PCG iteration would.  hipcc --offload-arch=gfx950 -O3 -o probe probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ bool grid_barrier(unsigned* bar, unsigned target, unsigned* err) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __threadfence();
        atomicAdd(bar, 1u);
        unsigned spins = 0;
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > (1u << 22)) { atomicExch(err, 1u); ok = false; break; }
            __builtin_amdgcn_s_sleep(0);
        }
        __threadfence();
    }
    __syncthreads();
    return ok;
}
__global__ __launch_bounds__(1024) void k_probe(unsigned* bar, unsigned* err, double* buf, size_t per_wg, int rounds, int touch) {
    extern __shared__ double smem[];
    smem[threadIdx.x] = 1.0;
    double* mine = buf + (size_t)blockIdx.x * per_wg;
    for (int r = 0; r < rounds; ++r) {
        if (touch) for (size_t i = threadIdx.x; i < per_wg; i += blockDim.x) mine[i] += smem[threadIdx.x & 7];
        if (!grid_barrier(bar, (unsigned)(r + 1) * gridDim.x, err)) return;
        if (*(volatile unsigned*)err) return;
    }
}
int main() {
    int dev = 0; hipSetDevice(dev);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, dev);
    const int G = p.multiProcessorCount;
    unsigned* bar; hipMalloc(&bar, 8); 
    const size_t per_wg = 4096;            // 32 KiB dirtied per workgroup and round (8 MiB per round over the grid)
    double* buf; hipMalloc(&buf, sizeof(double) * per_wg * G); hipMemset(buf, 0, sizeof(double) * per_wg * G);
    const size_t lds = 144 * 1024;
    hipFuncSetAttribute((const void*)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int touch = 0; touch < 2; ++touch)
        for (int rounds : {1, 101, 1001}) {
            hipMemset(bar, 0, 8);
            unsigned* err = bar + 1;
            void* args[] = {&bar, &err, &buf, (void*)&per_wg, &rounds, &touch};
            hipEventRecord(a);
            hipError_t e = hipLaunchCooperativeKernel((const void*)k_probe, dim3(G), dim3(1024), args, lds, 0);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms = 0; hipEventElapsedTime(&ms, a, b);
            unsigned h[2]; hipMemcpy(h, bar, 8, hipMemcpyDeviceToHost);
            printf("G=%d touch=%d rounds=%d: %s, %.1f us total, err=%u\n", G, touch, rounds, hipGetErrorString(e), 1e3 * ms, h[1]);
        }
    return 0;
}
