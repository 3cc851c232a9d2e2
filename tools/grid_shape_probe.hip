// Does a launch's grid SHAPE change what a small dependent launch costs?  The query-lane dimension of the batched answer path (kernels.h Lanes)
// was first put on gridDim.z, and the latency-bound launches of a 4-lane batch then took 10-19 us where the same number of workgroups takes ~5 us.
// A chain of dependent launches of one small kernel (each block reads 16 KiB written by the previous launch, spins a little, writes 16 KiB),
// replayed as a hipGraph, for several grid shapes with the same number of workgroups; and the same with the lanes' data far apart in memory.
// hipcc --offload-arch=gfx950 -O3 tools/grid_shape_probe.hip -o tools/grid_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k(const unsigned long long* in, unsigned long long* out, size_t lane_stride, unsigned work_per_lane, int spin) {
    // linear workgroup index -> (lane, block within the lane), whatever the grid shape
    const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned lane = lin / work_per_lane, b = lin - lane * work_per_lane;
    const unsigned long long* src = in + lane * lane_stride + (size_t)b * 2048;
    unsigned long long* dst = out + lane * lane_stride + (size_t)b * 2048;
    unsigned long long x[8];
    for (int r = 0; r < 8; r++) x[r] = src[threadIdx.x + 256 * r];
    for (int i = 0; i < spin; i++)
        for (int r = 0; r < 8; r++) x[r] = x[r] * 6364136223846793005ull + 1442695040888963407ull;
    for (int r = 0; r < 8; r++) dst[threadIdx.x + 256 * r] = x[r];
}

int main() {
    const size_t stride = (size_t)60 << 20;  // words between lanes: 480 MB, as the servers' arenas
    unsigned long long *a, *b;
    OK(hipMalloc(&a, 4 * stride * 8));
    OK(hipMalloc(&b, 4 * stride * 8));
    OK(hipMemset(a, 1, 4 * stride * 8));
    OK(hipMemset(b, 1, 4 * stride * 8));
    hipStream_t s;
    OK(hipStreamCreate(&s));
    struct Shape { const char* name; dim3 g; unsigned wpl; size_t ls; };
    const int chain = 20;
    for (unsigned wpl : {6u, 24u, 96u, 384u}) {
        std::vector<Shape> shapes = {
            {"(4w,1,1) lanes far apart  ", dim3(4 * wpl, 1, 1), wpl, stride},
            {"(w,1,4)  lanes far apart  ", dim3(wpl, 1, 4), wpl, stride},
            {"(w,4,1)  lanes far apart  ", dim3(wpl, 4, 1), wpl, stride},
            {"(4w,1,1) lanes adjacent   ", dim3(4 * wpl, 1, 1), wpl, (size_t)wpl * 2048},
            {"(w,1,4)  lanes adjacent   ", dim3(wpl, 1, 4), wpl, (size_t)wpl * 2048},
            {"(w,1,1)  one lane         ", dim3(wpl, 1, 1), wpl, stride},
        };
        for (auto& sh : shapes) {
            hipGraph_t g;
            hipGraphExec_t ge;
            OK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
            for (int i = 0; i < chain; i++) hipLaunchKernelGGL(k, sh.g, dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, sh.ls, sh.wpl, 40);
            OK(hipStreamEndCapture(s, &g));
            OK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            hipEvent_t e0, e1;
            OK(hipEventCreate(&e0));
            OK(hipEventCreate(&e1));
            for (int w = 0; w < 5; w++) OK(hipGraphLaunch(ge, s));
            OK(hipEventRecord(e0, s));
            const int reps = 20;
            for (int w = 0; w < reps; w++) OK(hipGraphLaunch(ge, s));
            OK(hipEventRecord(e1, s));
            OK(hipEventSynchronize(e1));
            float ms;
            OK(hipEventElapsedTime(&ms, e0, e1));
            printf("w = %3u workgroups per lane  %s %6.2f us per launch\n", wpl, sh.name, ms * 1e3 / (reps * chain));
            OK(hipGraphExecDestroy(ge));
            OK(hipGraphDestroy(g));
        }
    }
    return 0;
}
