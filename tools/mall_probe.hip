// Does a buffer that one launch wrote come back out of the 256 MiB Infinity Cache in the next launch?  (the batch's digit-transform -> product pairs)
// build: hipcc -O3 --offload-arch=gfx950 tools/mall_probe.hip -o tools/mall_probe;  run: tools/mall_probe
// For S = 32 MiB ... 1 GiB: write S bytes (16 B per lane, streaming), then read them back in a second launch (sum); also read-after-read (clean lines) and a read with a cold cache
// (a 1 GiB memset of another buffer in between).  Prints us and TB/s of the read launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void write_kernel(u32x4* dst, size_t n16, uint32_t v) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = u32x4{v, v + 1, v + 2, (uint32_t)i};
}
__global__ __launch_bounds__(256) void read_kernel(const u32x4* src, size_t n16, uint32_t* out) {
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const u32x4 x = src[i];
        acc += x.x ^ x.y ^ x.z ^ x.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main() {
    const size_t maxb = 1ull << 30;
    u32x4 *buf, *other;
    uint32_t* out;
    CK(hipMalloc(&buf, maxb));
    CK(hipMalloc(&other, maxb));
    CK(hipMalloc(&out, 64));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int grid = 256 * 8;
    printf("%8s %28s %28s %28s %28s\n", "MiB", "read after write", "read after read", "read, cache flushed", "write");
    for (size_t mib : {32, 64, 96, 128, 160, 192, 256, 384, 512, 1024}) {
        const size_t n16 = mib * (1ull << 20) / 16;
        float best[4] = {1e9f, 1e9f, 1e9f, 1e9f};
        for (int rep = 0; rep < 6; rep++) {
            float ms;
            // write, then read
            CK(hipEventRecord(e0, s));
            write_kernel<<<grid, 256, 0, s>>>(buf, n16, rep);
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best[3]) best[3] = ms;
            CK(hipEventRecord(e0, s));
            read_kernel<<<grid, 256, 0, s>>>(buf, n16, out);
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best[0]) best[0] = ms;
            // read again
            CK(hipEventRecord(e0, s));
            read_kernel<<<grid, 256, 0, s>>>(buf, n16, out);
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best[1]) best[1] = ms;
            // flush: stream 1 GiB of something else through the caches
            write_kernel<<<grid, 256, 0, s>>>(other, maxb / 16, rep);
            read_kernel<<<grid, 256, 0, s>>>(other, maxb / 16, out);
            CK(hipEventRecord(e0, s));
            read_kernel<<<grid, 256, 0, s>>>(buf, n16, out);
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best[2]) best[2] = ms;
        }
        const double b = (double)mib * (1 << 20);
        printf("%8zu", mib);
        for (int k = 0; k < 4; k++) printf("   %9.1f us %6.2f TB/s     ", best[k] * 1e3, b / (best[k] * 1e-3) / 1e12);
        printf("\n");
    }
    return 0;
}
