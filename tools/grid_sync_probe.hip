// How much does a grid-wide barrier cost on this part, compared with the ~4 us of a dependent kernel launch?
// Cooperative launch (the runtime guarantees co-residency or refuses), cooperative_groups grid.sync(), plus a hand-made
// barrier on a device-scope atomic counter.  hipcc --offload-arch=gfx950 -O3 tools/grid_sync_probe.hip -o tools/grid_sync_probe
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <vector>
#include <algorithm>
namespace cg = cooperative_groups;

__global__ void k_cg(unsigned long long* buf, int iters) {
    cg::grid_group grid = cg::this_grid();
    unsigned long long x = threadIdx.x;
    for (int i = 0; i < iters; i++) {
        buf[blockIdx.x * blockDim.x + threadIdx.x] = x + i;  // something every phase writes and the next one reads
        grid.sync();
        x += buf[((blockIdx.x + 1) % gridDim.x) * blockDim.x + threadIdx.x];
    }
    if (x == 0x123456789ull) buf[0] = x;
}
// sense-reversing barrier on one counter (all blocks co-resident: launched cooperatively)
__device__ void atomic_barrier(unsigned int* counter, unsigned int nblocks, unsigned int phase) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        atomicAdd(counter, 1u);
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < nblocks * (phase + 1)) __builtin_amdgcn_s_sleep(1);
        __threadfence();
    }
    __syncthreads();
}
__global__ void k_atomic(unsigned long long* buf, unsigned int* counter, int iters) {
    unsigned long long x = threadIdx.x;
    for (int i = 0; i < iters; i++) {
        buf[blockIdx.x * blockDim.x + threadIdx.x] = x + i;
        atomic_barrier(counter, gridDim.x, i);
        x += __builtin_nontemporal_load(&buf[((blockIdx.x + 1) % gridDim.x) * blockDim.x + threadIdx.x]);
    }
    if (x == 0x123456789ull) buf[0] = x;
}
// the cheaper protocol of MI355X_MICROARCH.md (inter-workgroup visibility): lane 0 releases once, arrives with a relaxed
// atomic, polls with RELAXED agent-scope loads (an acquire load per poll is 2-3x slower per hop) and acquires once
__device__ void relaxed_barrier(unsigned int* counter, unsigned int nblocks, unsigned int phase) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nblocks * (phase + 1)) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}
__global__ void k_relaxed(unsigned long long* buf, unsigned int* counter, int iters) {
    unsigned long long x = threadIdx.x;
    for (int i = 0; i < iters; i++) {
        buf[blockIdx.x * blockDim.x + threadIdx.x] = x + i;
        relaxed_barrier(counter, gridDim.x, i);
        x += buf[((blockIdx.x + 1) % gridDim.x) * blockDim.x + threadIdx.x];
    }
    if (x == 0x123456789ull) buf[0] = x;
}
__global__ void k_plain(unsigned long long* buf, int i) {
    buf[blockIdx.x * blockDim.x + threadIdx.x] += buf[((blockIdx.x + 1) % gridDim.x) * blockDim.x + threadIdx.x] + i;
}

int main() {
    const int iters = 200;
    for (int nblocks : {4, 16, 64, 256, 512, 1024}) {
        unsigned long long* buf;
        unsigned int* counter;
        hipMalloc(&buf, (size_t)nblocks * 256 * 8);
        hipMalloc(&counter, 4);
        hipMemset(buf, 0, (size_t)nblocks * 256 * 8);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        float ms;
        int it = iters;
        void* args1[] = {&buf, &it};
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            hipError_t e = hipLaunchCooperativeKernel((void*)k_cg, dim3(nblocks), dim3(256), args1, 0, 0);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("blocks %4d  cg grid.sync      : %7.2f us per phase (%s)\n", nblocks, ms * 1e3 / iters, hipGetErrorString(e));
        }
        void* args2[] = {&buf, &counter, &it};
        for (int rep = 0; rep < 2; rep++) {
            hipMemset(counter, 0, 4);
            hipEventRecord(e0);
            hipError_t e = hipLaunchCooperativeKernel((void*)k_atomic, dim3(nblocks), dim3(256), args2, 0, 0);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("blocks %4d  threadfence + acquire-poll barrier: %7.2f us per phase (%s)\n", nblocks, ms * 1e3 / iters, hipGetErrorString(e));
        }
        for (int rep = 0; rep < 2; rep++) {
            hipMemset(counter, 0, 4);
            hipEventRecord(e0);
            hipError_t e = hipLaunchCooperativeKernel((void*)k_relaxed, dim3(nblocks), dim3(256), args2, 0, 0);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("blocks %4d  release/relaxed-poll/acquire barrier: %7.2f us per phase (%s)\n", nblocks, ms * 1e3 / iters, hipGetErrorString(e));
        }
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            for (int i = 0; i < iters; i++) hipLaunchKernelGGL(k_plain, dim3(nblocks), dim3(256), 0, 0, buf, i);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("blocks %4d  dependent launches: %7.2f us per phase\n", nblocks, ms * 1e3 / iters);
        }
        hipFree(buf);
        hipFree(counter);
    }
    return 0;
}
