// What would "pull the multiply-accumulate into its consumer" cost?  
// Consumer-side fusion makes the 256-thread workgroup that next inverse-transforms an output polynomial gather its K-term
// product itself: out[z] = sum_{k<K} key[k][z] * D[b][k][z] over the workgroup's 2048 slots, i.e. 2 K polynomials of 16 KiB
// streamed into ONE workgroup (fold rounds: K = 2 m2 = 48 -> 1.5 MiB per output polynomial; odd expansion ciphertexts:
// K = t_exp_right = 56).  This probe times exactly that gather (16-byte loads, 8 slots per thread as the transforms hold
// them, u64 accumulators, result stored so nothing is optimised away) for the grid sizes of config 2's fold rounds, against
// the split form the product kernels use today (one thread per slot and 4 k-groups per 64 slots: fold_mac_kernel), on the
// same operands.  hipcc --offload-arch=gfx950 -O3 tools/mac_gather_probe.hip -o tools/mac_gather_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr uint32_t kN = 2048;
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

// consumer-side gather: block b owns output polynomial b; key is shared by all blocks, D is private to the block
__global__ __launch_bounds__(256) void gather_kernel(const uint64_t* key, const uint64_t* d, uint64_t* out, uint32_t K) {
    const uint32_t tid = threadIdx.x, b = blockIdx.x;
    uint64_t alo[8] = {}, ahi[8] = {};
    const u64x2* kp = reinterpret_cast<const u64x2*>(key) + tid;
    const u64x2* dp = reinterpret_cast<const u64x2*>(d + (size_t)b * K * kN) + tid;
#pragma unroll 2
    for (uint32_t k = 0; k < K; k++) {
        u64x2 kv[4], dv[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            kv[q] = kp[(size_t)k * (kN / 2) + q * 256];
            dv[q] = __builtin_nontemporal_load(dp + (size_t)k * (kN / 2) + q * 256);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            alo[2 * q] += (uint64_t)(uint32_t)kv[q].x * (uint32_t)dv[q].x;
            ahi[2 * q] += (kv[q].x >> 32) * (dv[q].x >> 32);
            alo[2 * q + 1] += (uint64_t)(uint32_t)kv[q].y * (uint32_t)dv[q].y;
            ahi[2 * q + 1] += (kv[q].y >> 32) * (dv[q].y >> 32);
        }
    }
    u64x2* op = reinterpret_cast<u64x2*>(out + (size_t)b * kN) + tid;
#pragma unroll
    for (int q = 0; q < 4; q++) op[q * 256] = u64x2{alo[2 * q] ^ ahi[2 * q], alo[2 * q + 1] ^ ahi[2 * q + 1]};
}

// the split form: 64 slots x 4 k-groups per workgroup, 32 workgroups per output polynomial (as fold_mac_kernel<1>, one row)
__global__ __launch_bounds__(256) void split_kernel(const uint64_t* key, const uint64_t* d, uint64_t* out, uint32_t K) {
    __shared__ uint64_t sh[3][64][2];
    const uint32_t zz = threadIdx.x & 63u, kg = threadIdx.x >> 6, z = blockIdx.x * 64u + zz, b = blockIdx.y;
    uint64_t lo = 0, hi = 0;
#pragma unroll 4
    for (uint32_t k = kg; k < K; k += 4) {
        const uint64_t kv = key[(size_t)k * kN + z], dv = __builtin_nontemporal_load(&d[((size_t)b * K + k) * kN + z]);
        lo += (uint64_t)(uint32_t)kv * (uint32_t)dv;
        hi += (kv >> 32) * (dv >> 32);
    }
    if (kg) {
        sh[kg - 1][zz][0] = lo;
        sh[kg - 1][zz][1] = hi;
    }
    __syncthreads();
    if (kg == 0) {
        for (int q = 0; q < 3; q++) {
            lo += sh[q][zz][0];
            hi += sh[q][zz][1];
        }
        out[(size_t)b * kN + z] = lo ^ hi;
    }
}

// usage: tools/mac_gather_probe [K=48] [outputs ...]   (round 6: K = 8 with 2 .. 128 outputs = the even tree of expansion rounds 0 .. 5, both rows;
// K = 24 + addend ~ the pair-form fold's narrow rounds)
int main(int argc, char** argv) {
    const uint32_t K = argc > 1 ? (uint32_t)atoi(argv[1]) : 48u;
    std::vector<uint32_t> outs;
    for (int i = 2; i < argc; i++) outs.push_back((uint32_t)atoi(argv[i]));
    if (outs.empty()) outs = {6u, 12u, 48u, 96u, 192u, 384u, 768u};
    uint32_t max_out = 0;
    for (uint32_t n : outs) max_out = n > max_out ? n : max_out;
    uint64_t *key, *d, *out;
    hipMalloc(&key, (size_t)K * kN * 8);
    hipMalloc(&d, (size_t)max_out * K * kN * 8);
    hipMalloc(&out, (size_t)max_out * kN * 8);
    hipMemset(key, 0x11, (size_t)K * kN * 8);
    hipMemset(d, 0x22, (size_t)max_out * K * kN * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("K = %u terms per output polynomial: %.2f MiB gathered per output polynomial\n", K, 2.0 * K * kN * 8 / (1 << 20));
    printf("%10s %28s %28s\n", "outputs", "consumer-side gather (us)", "split over 32 workgroups (us)");
    // output polynomials per launch: config 2's fold rounds hold 6 np' of them, np' = 64 ... 1; one digit job per block in the
    // narrow rounds repeats the gather ell = 8 times (48 blocks for np' = 1)
    for (uint32_t n : outs) {
        float ms[2];
        for (int which = 0; which < 2; which++) {
            const int iters = 20;
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                for (int i = 0; i < iters; i++) {
                    if (which == 0)
                        hipLaunchKernelGGL(gather_kernel, dim3(n), dim3(256), 0, 0, key, d, out, K);
                    else
                        hipLaunchKernelGGL(split_kernel, dim3(kN / 64, n), dim3(256), 0, 0, key, d, out, K);
                }
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms[which], e0, e1);
            }
            ms[which] = ms[which] * 1e3f / iters;
        }
        printf("%10u %28.2f %28.2f\n", n, ms[0], ms[1]);
    }
    return 0;
}
