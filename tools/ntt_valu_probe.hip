// The register-resident arithmetic of one forward transform (4 passes of the real ntt_device.h code, mid and final range
// reductions) in a loop with no LDS, no barriers and no memory traffic: what the VALU part alone costs per transform at
// 8 / 4 / 2 / 1 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -I spiral_amd/csrc tools/ntt_valu_probe.hip -o tools/ntt_valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include "ntt_device.h"
using namespace spiral;
#define ITERS 256
template <int MODE>
__global__ __launch_bounds__(256) void k(const uint4* tw, uint32_t* out) {
    const uint32_t tid = threadIdx.x;
    uint32_t lo[8], hi[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        lo[r] = tid * 977u + r;
        hi[r] = tid * 131u + r;
    }
    const Tw7 wb = tw_load8(tw, 8 + (tid >> 5), 16 + 2 * (tid >> 5), 32 + 4 * (tid >> 5));
    const Tw7 wc = tw_load8(tw, 64 + (tid >> 2), 128 + 2 * (tid >> 2), 256 + 4 * (tid >> 2));
    const Tw7 wd = tw_load4x2(tw, 512 + 2 * tid, 1024 + 4 * tid);
    for (int it = 0; it < ITERS; it++) {
        if (MODE == 0) {  // the transform's four passes + reductions
            ct_radix8(lo, hi, tw, 1, 2, 4);
            ct_radix8_pre(lo, hi, wb);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                lo[q] = lazy_reduce(lo[q], kP);
                hi[q] = lazy_reduce(hi[q], kB);
            }
            ct_radix8_pre(lo, hi, wc);
            ct_radix4x2_pre(lo, hi, wd);
#pragma unroll
            for (int q = 0; q < 8; q++) {
                lo[q] = csub_min(lazy_reduce(lo[q], kP), kP);
                hi[q] = csub_min(lazy_reduce(hi[q], kB), kB);
            }
        } else {  // butterflies only (no reductions): values wrap, timing only
            ct_radix8(lo, hi, tw, 1, 2, 4);
            ct_radix8_pre(lo, hi, wb);
            ct_radix8_pre(lo, hi, wc);
            ct_radix4x2_pre(lo, hi, wd);
        }
    }
    uint32_t x = 0;
#pragma unroll
    for (int r = 0; r < 8; r++) x ^= lo[r] ^ hi[r];
    if (x == 0x1234567u) out[tid] = x;
}
template <int MODE>
void run(const char* name, const uint4* tw, uint32_t* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-34s", name);
    for (int wps : {8, 4, 2, 1}) {
        const int blocks = 256 * wps;
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, tw, out);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, tw, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double per_wg_iter_ns = ms * 1e6 / ITERS / wps;  // ns per transform-equivalent per CU
        printf("  %dw/SIMD %6.0f ns/CU = %5.2f ns chip-wide (%5.0f cyc)", wps, per_wg_iter_ns, per_wg_iter_ns / 256, per_wg_iter_ns * 2.4);
    }
    printf("\n");
}
int main() {
    uint4* tw; uint32_t* out;
    hipMalloc(&tw, 2048 * 16); hipMemset(tw, 0x11, 2048 * 16); hipMalloc(&out, 4096);
    run<0>("4 passes + range reductions", tw, out);
    run<1>("4 passes, butterflies only", tw, out);
    return 0;
}
