// HBM / cache bandwidth of pure-write, pure-read and copy streams at the buffer sizes the digit-transform kernels produce
// (16 MiB .. 1 GiB): hipcc --offload-arch=gfx950 -O3 tools/mem_bw_probe.hip -o tools/mem_bw_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// each 256-thread block moves 16 KiB (one polynomial): 4 x 16 B per thread, as pk_store8 / pk_load8
__global__ __launch_bounds__(256) void wr(u32x4* dst, uint32_t v) {
    u32x4* p = dst + (size_t)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int q = 0; q < 4; q++) p[q * 256] = u32x4{v, v + 1, v + 2, v + (uint32_t)q};
}
__global__ __launch_bounds__(256) void wr_nt(u32x4* dst, uint32_t v) {
    u32x4* p = dst + (size_t)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int q = 0; q < 4; q++) __builtin_nontemporal_store(u32x4{v, v + 1, v + 2, v + (uint32_t)q}, p + q * 256);
}
__global__ __launch_bounds__(256) void rd(const u32x4* src, uint32_t* out) {
    const u32x4* p = src + (size_t)blockIdx.x * 1024 + threadIdx.x;
    u32x4 a = p[0] ^ p[256] ^ p[512] ^ p[768];
    if ((a.x ^ a.y ^ a.z ^ a.w) == 0x12345u) out[0] = 1;
}
__global__ __launch_bounds__(256) void cp(const u32x4* src, u32x4* dst) {
    const size_t o = (size_t)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int q = 0; q < 4; q++) dst[o + q * 256] = src[o + q * 256];
}
int main() {
    const size_t maxb = (size_t)1 << 30;
    u32x4 *a, *b;
    uint32_t* o;
    hipMalloc(&a, maxb); hipMalloc(&b, maxb); hipMalloc(&o, 64);
    hipMemset(a, 1, maxb); hipMemset(b, 2, maxb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (size_t mb : {16, 32, 64, 128, 256, 1024}) {
        const size_t bytes = mb << 20; const uint32_t blocks = (uint32_t)(bytes / 16384);
        float t[4];
        for (int k = 0; k < 4; k++) {
            for (int rep = 0; rep < 2; rep++) {  // second pass timed (steady state: the buffer may sit in the caches)
                hipEventRecord(e0);
                for (int i = 0; i < 10; i++) {
                    if (k == 0) hipLaunchKernelGGL(wr, dim3(blocks), dim3(256), 0, 0, a, (uint32_t)i);
                    if (k == 1) hipLaunchKernelGGL(wr_nt, dim3(blocks), dim3(256), 0, 0, a, (uint32_t)i);
                    if (k == 2) hipLaunchKernelGGL(rd, dim3(blocks), dim3(256), 0, 0, a, o);
                    if (k == 3) hipLaunchKernelGGL(cp, dim3(blocks), dim3(256), 0, 0, a, b);
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&t[k], e0, e1);
            }
        }
        printf("%5zu MiB: write %6.0f GB/s  write nt %6.0f GB/s  read %6.0f GB/s  copy %6.0f GB/s (read + write bytes)   [us per launch: %.1f %.1f %.1f %.1f]\n", mb,
               bytes / (t[0] * 1e-4) / 1e9, bytes / (t[1] * 1e-4) / 1e9, bytes / (t[2] * 1e-4) / 1e9, 2.0 * bytes / (t[3] * 1e-4) / 1e9, t[0] * 100, t[1] * 100, t[2] * 100, t[3] * 100);
    }
    return 0;
}
