// Is a launch of thousands of short workgroups bound by the rate at which workgroups (waves) can be LAUNCHED?  (fold_mac_kernel<2> at four lanes:
// 4096 workgroups of ~7 us each in 51 us = 80 workgroups/us, 55 % of the register-limited residency.)  The same J jobs -- each reads 16 KiB, does a
// little arithmetic, writes 16 KiB; 256 threads -- run as (a) J workgroups, (b) G < J workgroups that each loop over J / G jobs (grid-stride), for several
// G; plus the same with 64-thread workgroups (one wave per job).  hipcc --offload-arch=gfx950 -O3 tools/dispatch_probe.hip -o tools/dispatch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned long long u64;
template <int T>
__global__ __launch_bounds__(T) void k(const u64* in, u64* out, unsigned njobs, int spin) {
    __shared__ u64 sh[2048];
    for (unsigned j = blockIdx.x; j < njobs; j += gridDim.x) {
        const u64* src = in + (size_t)j * 2048;
        u64* dst = out + (size_t)j * 2048;
        u64 x[2048 / T];
        for (int r = 0; r < 2048 / T; r++) x[r] = src[threadIdx.x + T * r];
        for (int i = 0; i < spin; i++)
            for (int r = 0; r < 2048 / T; r++) x[r] = x[r] * 6364136223846793005ull + 1442695040888963407ull;
        for (int r = 0; r < 2048 / T; r++) sh[threadIdx.x + T * r] = x[r];
        __syncthreads();
        for (int r = 0; r < 2048 / T; r++) dst[threadIdx.x + T * r] = sh[(threadIdx.x + T * r) ^ 1];
        __syncthreads();
    }
}
int main() {
    const unsigned J = 12288;
    u64 *a, *b;
    OK(hipMalloc(&a, (size_t)J * 2048 * 8));
    OK(hipMalloc(&b, (size_t)J * 2048 * 8));
    OK(hipMemset(a, 1, (size_t)J * 2048 * 8));
    hipEvent_t e0, e1;
    OK(hipEventCreate(&e0));
    OK(hipEventCreate(&e1));
    for (int spin : {0, 8, 32}) {
        for (unsigned G : {12288u, 6144u, 4096u, 2048u, 1024u, 512u}) {
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                OK(hipEventRecord(e0));
                for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k<256>, dim3(G), dim3(256), 0, 0, a, b, J, spin);
                OK(hipEventRecord(e1));
                OK(hipEventSynchronize(e1));
                OK(hipEventElapsedTime(&ms, e0, e1));
            }
            printf("spin %2d: %5u jobs of 16 KiB in + 16 KiB out as %5u workgroups of 256 threads: %7.2f us per launch (%6.1f jobs/us)\n", spin, J, G, ms * 50, J / (ms * 50));
        }
    }
    return 0;
}
