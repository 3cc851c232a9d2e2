// Barrier among workgroups of ONE XCD: workgroups are dealt to the 8 XCDs round-robin by id, so the blocks with id % 8 == 0 of a launch all
// run on one XCD and share its L2.  A barrier between them needs no cross-XCD coherence: stores are write-through to that L2, the arrive /
// poll atomics execute in it (workgroup-scope atomics carry no sc bits), and the acquire side only has to drop the CU's own L1 lines
// (buffer_inv sc1) -- no L2 write-back (buffer_wbl2), which is what makes the device-wide barriers of grid_sync_probe.hip cost 1.3-60 us.
// Prints us per phase for n = 4 .. 64 working blocks, against dependent launches, and checks the data every phase exchanges.
// hipcc --offload-arch=gfx950 -O3 tools/xcd_sync_probe.hip -o tools/xcd_sync_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void xcd_barrier(unsigned int* counter, unsigned int nblocks, unsigned int phase) {
    __syncthreads();
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this block's stores have reached the L2
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        unsigned int spins = 0;  // bounded: a probe must not hang the box if the blocks turn out not to share an L2
        while (__hip_atomic_fetch_add(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < nblocks * (phase + 1) && ++spins < 3000u) __builtin_amdgcn_s_sleep(1);
        if (spins >= 3000u) counter[16] = 1u;  // timed out
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // buffer_inv sc1: later loads miss the L1
}
// each phase: block b writes f(phase, b, tid), then reads block (b + 1) % n's value of this phase and folds it into x
__global__ void k_xcd(unsigned long long* buf, unsigned int* counter, unsigned long long* out, int iters, int payload) {
    if (threadIdx.x == 0 && blockIdx.x < 64) counter[40 + blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20);  // HW_REG_XCC_ID
    if (blockIdx.x & 7u) return;
    const unsigned int b = blockIdx.x >> 3, n = gridDim.x >> 3;
    unsigned long long x = threadIdx.x;
    for (int i = 0; i < iters; i++) {
        for (int q = 0; q < payload; q++) buf[((size_t)b * payload + q) * blockDim.x + threadIdx.x] = x * 3 + i + q;
        xcd_barrier(counter, n, i);
        unsigned long long y = 0;
        for (int q = 0; q < payload; q++) y += buf[((size_t)((b + 1) % n) * payload + q) * blockDim.x + threadIdx.x];
        // everyone has to finish reading before the next phase overwrites: second barrier (a real pipeline double-buffers instead)
        xcd_barrier(counter + 32, n, i);
        x = x * 5 + y;
    }
    out[(size_t)b * blockDim.x + threadIdx.x] = x;
}
__global__ void k_plain(unsigned long long* buf, int i) { buf[blockIdx.x * blockDim.x + threadIdx.x] += buf[((blockIdx.x + 1) % gridDim.x) * blockDim.x + threadIdx.x] + i; }

int main() {
    const int iters = 20; setvbuf(stdout, nullptr, _IONBF, 0);
    for (int payload : {1, 8}) {
        for (int n : {4, 8, 16, 32, 64}) {
            unsigned long long *buf, *out;
            unsigned int* counter;
            hipMalloc(&buf, (size_t)n * payload * 256 * 8);
            hipMalloc(&out, (size_t)n * 256 * 8);
            hipMalloc(&counter, 1024);
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                hipMemset(counter, 0, 1024);
                hipEventRecord(e0);
                hipLaunchKernelGGL(k_xcd, dim3(8 * n), dim3(256), 0, 0, buf, counter, out, iters, payload);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            std::vector<unsigned long long> h((size_t)n * 256), x((size_t)n * 256), nx((size_t)n * 256);
            hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
            for (int b = 0; b < n; b++) for (int t = 0; t < 256; t++) x[b * 256 + t] = t;
            for (int i = 0; i < iters; i++) {
                for (int b = 0; b < n; b++) for (int t = 0; t < 256; t++) {
                    unsigned long long y = 0, xo = x[((b + 1) % n) * 256 + t];
                    for (int q = 0; q < payload; q++) y += xo * 3 + i + q;
                    nx[b * 256 + t] = x[b * 256 + t] * 5 + y;
                }
                x = nx;
            }
            size_t bad = 0;
            for (size_t k = 0; k < h.size(); k++) bad += h[k] != x[k];
            unsigned int hc[256];
            hipMemcpy(hc, counter, 1024, hipMemcpyDeviceToHost);
            printf("payload %d x 2 KiB  blocks %3d on one XCD: %6.2f us per phase (two barriers + the exchange), data %s%s  xcc of blocks 0..15:", payload, n, ms * 1e3 / iters, bad ? "WRONG" : "ok", hc[16] ? " BARRIER TIMED OUT" : "");
            for (int k = 0; k < 16; k++) printf(" %u", hc[40 + k]);
            printf("\n");
            hipFree(buf);
            hipFree(out);
            hipFree(counter);
        }
    }
    for (int n : {4, 16, 64}) {
        unsigned long long* buf;
        hipMalloc(&buf, (size_t)n * 256 * 8);
        hipMemset(buf, 0, (size_t)n * 256 * 8);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            for (int i = 0; i < iters; i++) hipLaunchKernelGGL(k_plain, dim3(n), dim3(256), 0, 0, buf, i);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        printf("blocks %3d dependent launches: %6.2f us per phase\n", n, ms * 1e3 / iters);
    }
    return 0;
}
