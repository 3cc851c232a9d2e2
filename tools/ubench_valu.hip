// Micro-benchmark: issue rate of the integer VALU instructions the NTT / sweep kernels are built from.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o gpurun_out/ubench_valu && gpurun_out/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITERS 2048
#define REP8(x) x x x x x x x x

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed) {
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;
    uint32_t b = seed | 1, c = seed * 77 + 5;
    uint64_t w0 = a0, w1 = a1, w2 = a2, w3 = a3, w4 = a4, w5 = a5, w6 = a6, w7 = a7;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7, db = 1.0000001, dc = 1e-9;
    for (int i = 0; i < ITERS; i++) {
        if constexpr (OP == 0) {  // v_add_u32
#define X(r) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r) : "v"(b));
            X(a0) X(a1) X(a2) X(a3) X(a4) X(a5) X(a6) X(a7)
#undef X
        } else if constexpr (OP == 1) {  // v_mul_lo_u32
#define X(r) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r) : "v"(b));
            X(a0) X(a1) X(a2) X(a3) X(a4) X(a5) X(a6) X(a7)
#undef X
        } else if constexpr (OP == 2) {  // v_mul_hi_u32
#define X(r) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(r) : "v"(b));
            X(a0) X(a1) X(a2) X(a3) X(a4) X(a5) X(a6) X(a7)
#undef X
        } else if constexpr (OP == 3) {  // v_mad_u64_u32 (sgpr multiplier like the sweep)
#define X(r, s) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(r) : "v"(s), "s"(b) : "s20", "s21");
            X(w0, a0) X(w1, a1) X(w2, a2) X(w3, a3) X(w4, a4) X(w5, a5) X(w6, a6) X(w7, a7)
#undef X
        } else if constexpr (OP == 4) {  // v_min_u32
#define X(r) asm volatile("v_min_u32 %0, %0, %1" : "+v"(r) : "v"(b));
            X(a0) X(a1) X(a2) X(a3) X(a4) X(a5) X(a6) X(a7)
#undef X
        } else if constexpr (OP == 5) {  // v_mul_u32_u24
#define X(r) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r) : "v"(b));
            X(a0) X(a1) X(a2) X(a3) X(a4) X(a5) X(a6) X(a7)
#undef X
        } else if constexpr (OP == 6) {  // v_mad_u32_u24
#define X(r) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(c));
            X(a0) X(a1) X(a2) X(a3) X(a4) X(a5) X(a6) X(a7)
#undef X
        } else if constexpr (OP == 7) {  // v_fma_f64
#define X(r) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(r) : "v"(db), "v"(dc));
            X(d0) X(d1) X(d2) X(d3) X(d4) X(d5) X(d6) X(d7)
#undef X
        } else if constexpr (OP == 8) {  // v_mul_hi_u32_u24
#define X(r) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(r) : "v"(b));
            X(a0) X(a1) X(a2) X(a3) X(a4) X(a5) X(a6) X(a7)
#undef X
        } else if constexpr (OP == 9) {  // v_mad_u64_u32 with vgpr multiplier
#define X(r, s) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(r) : "v"(s), "v"(b) : "s20", "s21");
            X(w0, a0) X(w1, a1) X(w2, a2) X(w3, a3) X(w4, a4) X(w5, a5) X(w6, a6) X(w7, a7)
#undef X
        } else if constexpr (OP == 10) {  // v_cndmask after v_cmp pair (the compiler's conditional subtract)
#define X(r) asm volatile("v_cmp_gt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r) : "v"(b) : "vcc");
            X(a0) X(a1) X(a2) X(a3) X(a4) X(a5) X(a6) X(a7)
#undef X
        }
    }
    uint32_t r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)(w0 ^ w1 ^ w2 ^ w3 ^ w4 ^ w5 ^ w6 ^ w7) ^ (uint32_t)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
    if (r == 0x12345678) out[threadIdx.x] = r;
}

template <int OP>
void run(const char* name, uint32_t* d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int blocks = 256 * 8;  // 8 blocks of 256 threads per CU: 8 waves per SIMD
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double inst = (double)blocks * 4 /*waves*/ * ITERS * 8;  // wave-instructions
    double per_simd_per_s = inst / (256.0 * 4) / (ms * 1e-3);
    printf("%-28s %8.3f ms  %7.2f Gwave-inst/s/SIMD  -> %5.2f cycles per wave-instruction @2.4GHz\n", name, ms, per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s);
}

int main() {
    uint32_t* d;
    hipMalloc(&d, 4096);
    run<0>("v_add_u32", d);
    run<4>("v_min_u32", d);
    run<1>("v_mul_lo_u32", d);
    run<2>("v_mul_hi_u32", d);
    run<3>("v_mad_u64_u32 (sgpr)", d);
    run<9>("v_mad_u64_u32 (vgpr)", d);
    run<5>("v_mul_u32_u24", d);
    run<8>("v_mul_hi_u32_u24", d);
    run<6>("v_mad_u32_u24", d);
    run<7>("v_fma_f64", d);
    run<10>("v_cmp+v_cndmask pair", d);
    return 0;
}
