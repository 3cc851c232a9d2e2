// Tuning harness for the first-dimension sweep: variants of sweep_kernel on a synthetic 2 GiB database, timed
// interleaved with HIP events; every variant is checked against variant 0.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I spiral_amd/csrc tools/sweep_tune.hip -o tools/sweep_tune
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "common.h"
using namespace spiral;

typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void mac6(uint64_t (&a)[6], uint32_t p0, uint32_t p1, uint32_t p2, uint32_t b0, uint32_t b1, uint32_t b2, uint64_t w) {
    const uint32_t bl = lo32(w), bh = hi32(w);
    a[0] += (uint64_t)p0 * bl; a[1] += (uint64_t)p1 * bl; a[2] += (uint64_t)p2 * bl;
    a[3] += (uint64_t)b0 * bh; a[4] += (uint64_t)b1 * bh; a[5] += (uint64_t)b2 * bh;
}
__device__ __forceinline__ void mac_j(uint64_t (&a)[6], const uint4* q, uint64_t w0, uint64_t w1) {
    const uint4 qa = q[0], qb = q[1], qc = q[2];
    mac6(a, qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, w0);
    mac6(a, qb.z, qb.w, qc.x, qc.y, qc.z, qc.w, w1);
}
__device__ __forceinline__ void reduce6(uint64_t (&a)[6]) {
#pragma unroll
    for (int r = 0; r < 3; r++) { a[r] = mod_p(a[r]); a[3 + r] = mod_b(a[3 + r]); }
}
__device__ __forceinline__ void store_acc(uint64_t* acc, const uint64_t (&a)[6], uint32_t ic, uint32_t z) {
    const uint32_t ii = ic >> 1, c = ic & 1u;
#pragma unroll
    for (uint32_t r = 0; r < 3; r++) acc[((size_t)(6u * ii + 2u * r + c)) * kN + z] = pack((uint32_t)a[r], (uint32_t)a[3 + r]);
}

// one tile = (z, block of 64 columns); tiles numbered z*wpz + icb like the product kernel
template <int UNROLL, bool NT>
__device__ __forceinline__ void do_tile(const uint64_t* __restrict__ db, const uint32_t* __restrict__ qs, uint64_t* __restrict__ acc, uint32_t nic,
                                        uint32_t dim0, uint32_t tile, uint32_t lane) {
    const uint32_t wpz = nic >> 6;
    const uint32_t z = tile / wpz, ic = (tile - z * wpz) * 64u + lane;
    const u64x2* dbp = reinterpret_cast<const u64x2*>(db) + (size_t)tile * dim0 * 64u + lane;
    const uint4* q = reinterpret_cast<const uint4*>(qs) + (size_t)z * dim0 * 3u;
    uint64_t a[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t j0 = 0; j0 < dim0; j0 += 128) {
        const uint32_t jend = min(j0 + 128u, dim0);
#pragma unroll UNROLL
        for (uint32_t j = j0; j < jend; j++) {
            u64x2 w;
            if constexpr (NT) w = __builtin_nontemporal_load(dbp + (size_t)j * 64u);
            else w = dbp[(size_t)j * 64u];
            mac_j(a, q + j * 3u, w.x, w.y);
        }
        reduce6(a);
    }
    store_acc(acc, a, ic, z);
}

template <int UNROLL, bool NT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void sweep_v(const uint64_t* __restrict__ db, const uint32_t* __restrict__ qs, uint64_t* __restrict__ acc,
                                                      uint32_t nic, uint32_t dim0) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t tile = __builtin_amdgcn_readfirstlane(blockIdx.x * WAVES + (threadIdx.x >> 6));
    do_tile<UNROLL, NT>(db, qs, acc, nic, dim0, tile, lane);
}

// persistent: gridDim.x workgroups of WAVES waves stride over all tiles
template <int UNROLL, bool NT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void sweep_p(const uint64_t* __restrict__ db, const uint32_t* __restrict__ qs, uint64_t* __restrict__ acc,
                                                      uint32_t nic, uint32_t dim0, uint32_t ntiles) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w0 = __builtin_amdgcn_readfirstlane(blockIdx.x * WAVES + (threadIdx.x >> 6));
    for (uint32_t tile = w0; tile < ntiles; tile += gridDim.x * WAVES) do_tile<UNROLL, NT>(db, qs, acc, nic, dim0, tile, lane);
}

// two-pointer: each wave owns TWO tiles and alternates loads between them (more independent streams per wave)
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void sweep_2(const uint64_t* __restrict__ db, const uint32_t* __restrict__ qs, uint64_t* __restrict__ acc,
                                               uint32_t nic, uint32_t dim0) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
    // split the j range of one tile over two waves would need a combine; instead: half-length tiles (j split) with atomics is
    // avoided -- here each wave simply handles tile 2*wave and 2*wave+1 back to back
    do_tile<UNROLL, NT>(db, qs, acc, nic, dim0, 2 * wave, lane);
    do_tile<UNROLL, NT>(db, qs, acc, nic, dim0, 2 * wave + 1, lane);
}

__global__ void fill(uint64_t* p, size_t n, uint64_t seed) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint64_t x = seed + i + 0x9E3779B97F4A7C15ull;
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; x ^= x >> 31;
        p[i] = pack((uint32_t)(x & 0xffffffffull) % kP, (uint32_t)(x >> 32) % kB);
    }
}
__global__ void fillq(uint32_t* p, size_t n, uint64_t seed) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint64_t x = seed + i + 0x9E3779B97F4A7C15ull;
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; x ^= x >> 31;
        p[i] = (uint32_t)x % ((i % 6) < 3 ? kP : kB);
    }
}

struct Variant { const char* name; void (*launch)(const uint64_t*, const uint32_t*, uint64_t*, uint32_t, uint32_t); };

#define V(NAME, ...) {NAME, [](const uint64_t* db, const uint32_t* qs, uint64_t* acc, uint32_t nic, uint32_t dim0) { __VA_ARGS__; }}

int main(int argc, char** argv) {
    const uint32_t nu1 = argc > 1 ? atoi(argv[1]) : 8, nu2 = argc > 2 ? atoi(argv[2]) : 7;
    const uint32_t dim0 = 1u << nu1, num_per = 1u << nu2, nic = 2 * num_per, ntiles = kN * (nic / 64);
    const size_t dbw = (size_t)kN * dim0 * nic * 2, qw = (size_t)kN * dim0 * 12, accw = (size_t)num_per * 6 * kN;
    uint64_t *db, *acc, *ref; uint32_t* qs;
    hipMalloc(&db, dbw * 8); hipMalloc(&qs, qw * 4); hipMalloc(&acc, accw * 8); hipMalloc(&ref, accw * 8);
    fill<<<2048, 256>>>(db, dbw, 1); fillq<<<2048, 256>>>(qs, qw, 2);
    hipDeviceSynchronize();
    const double bytes = (double)dbw * 8 + (double)dim0 * 6 * kN * 8 + (double)num_per * 12 * kN * 8;
    std::vector<Variant> vs = {
        V("u4 nt w4 (v1 product)", hipLaunchKernelGGL((sweep_v<4, true, 4>), dim3(kN * (nic / 64) / 4), dim3(256), 0, 0, db, qs, acc, nic, dim0)),
        V("u8 nt w4", hipLaunchKernelGGL((sweep_v<8, true, 4>), dim3(kN * (nic / 64) / 4), dim3(256), 0, 0, db, qs, acc, nic, dim0)),
        V("u4 nt w1", hipLaunchKernelGGL((sweep_v<4, true, 1>), dim3(kN * (nic / 64)), dim3(64), 0, 0, db, qs, acc, nic, dim0)),
        V("u6 nt w1", hipLaunchKernelGGL((sweep_v<6, true, 1>), dim3(kN * (nic / 64)), dim3(64), 0, 0, db, qs, acc, nic, dim0)),
        V("u3 nt w1", hipLaunchKernelGGL((sweep_v<3, true, 1>), dim3(kN * (nic / 64)), dim3(64), 0, 0, db, qs, acc, nic, dim0)),
        V("u4 nt w2", hipLaunchKernelGGL((sweep_v<4, true, 2>), dim3(kN * (nic / 64) / 2), dim3(128), 0, 0, db, qs, acc, nic, dim0)),
        V("u8 nt w2", hipLaunchKernelGGL((sweep_v<8, true, 2>), dim3(kN * (nic / 64) / 2), dim3(128), 0, 0, db, qs, acc, nic, dim0)),
        V("u6 nt w2", hipLaunchKernelGGL((sweep_v<6, true, 2>), dim3(kN * (nic / 64) / 2), dim3(128), 0, 0, db, qs, acc, nic, dim0)),
        V("u6 nt w4", hipLaunchKernelGGL((sweep_v<6, true, 4>), dim3(kN * (nic / 64) / 4), dim3(256), 0, 0, db, qs, acc, nic, dim0)),
        V("u12 nt w4", hipLaunchKernelGGL((sweep_v<12, true, 4>), dim3(kN * (nic / 64) / 4), dim3(256), 0, 0, db, qs, acc, nic, dim0)),
        V("persist 8192w u4 nt w1", hipLaunchKernelGGL((sweep_p<4, true, 1>), dim3(8192), dim3(64), 0, 0, db, qs, acc, nic, dim0, kN * (nic / 64))),
    };
    (void)ntiles;
    std::vector<std::vector<float>> times(vs.size());
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<uint64_t> href(accw), hacc(accw);
    for (int round = 0; round < 6; round++) {
        for (size_t v = 0; v < vs.size(); v++) {
            hipMemset(acc, 0, accw * 8);
            hipEventRecord(e0);
            vs[v].launch(db, qs, acc, nic, dim0);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (round > 0) times[v].push_back(ms);
            if (round == 0) {
                hipMemcpy(v == 0 ? href.data() : hacc.data(), acc, accw * 8, hipMemcpyDeviceToHost);
                if (v > 0 && href != hacc) printf("!! variant %s differs from variant 0\n", vs[v].name);
            }
        }
    }
    for (size_t v = 0; v < vs.size(); v++) {
        std::sort(times[v].begin(), times[v].end());
        float med = times[v][times[v].size() / 2], mn = times[v][0];
        printf("%-28s median %7.1f us (min %7.1f)  %6.1f GB/s  %4.1f%% of 8 TB/s\n", vs[v].name, med * 1e3, mn * 1e3, bytes / (med * 1e-3) / 1e9,
               bytes / (med * 1e-3) / 8e12 * 100);
    }
    return 0;
}
