// EXPERIMENT, not part of the product build: the 16-coefficients-per-thread transform core measured in round 2 (see ntt_device.h).
// Negacyclic NTT / INTT over Z[x]/(x^2048+1) for the two 28-bit CRT primes: 128 threads (two waves) per polynomial, both
// limbs side by side (PK format, see common.h), 16 coefficients x 2 limbs per thread in VGPRs.
//
// Replaces ntt_forward / ntt_inverse (reference src/core.cpp:247-514).  Same transform and slot order (natural in ->
// bit-reversed out for the forward, the converse for the inverse) and the same twiddle table (regenerated from psi, see
// tables.cpp), organised for CDNA4: the 11 radix-2 stages run as three register-resident passes of 4 + 4 + 3 stages with
// lazy u32 Harvey butterflies, and the passes trade data through a 17 KiB LDS tile -- two exchanges per transform.  (The
// round-1 core held 8 coefficients per thread: four passes, three exchanges; LDS stores are the scarce resource here --
// a ds_write_b64 costs a SIMD 25 cycles against 2.9 for an add, profiles/r02_ubench_valu.txt -- so fewer, fatter passes
// win.)  Outputs are canonical residues in [0, m).
#pragma once
#include "common.h"
#ifndef TWI
#define TWI(i) (i)
#endif

namespace spiral {

constexpr uint32_t kNttThreads = 128;  // threads per polynomial
constexpr int kV = 16;                 // coefficients per thread

struct Tables {
    const uint4* fwd;  // [2048] {W_p, W'_p, W_b, W'_b} indexed like the reference's forward rows (m + i)
    const uint4* inv;  // [2048] same for the inverse rows (h + i), 1/2 folded in
};

// ---- butterflies --------------------------------------------------------------------------------
// Both primes are < 2^28, so a u32 holds values up to 16 m.  The butterflies are therefore fully lazy: no
// conditional subtraction inside a stage, one multiply-based range reduction where the bound would otherwise
// pass 14 m.  (The reference keeps [0, 4m) with a conditional subtract per butterfly, src/core.cpp:274-290;
// the canonical results are identical.)
//
// Shoup product: t = W*y - floor(W'*y / 2^32)*m lies in [0, 2m) for ANY y < 2^32.
__device__ __forceinline__ uint32_t shoup(uint32_t y, uint32_t w, uint32_t ws, uint32_t m) {
    return w * y - __umulhi(y, ws) * m;
}
// [0, 14m] -> [0, 2m): x - floor(x / 2^28) * m  (m > 0.9296 * 2^28, so the quotient is off by at most one)
__device__ __forceinline__ uint32_t lazy_reduce(uint32_t x, uint32_t m) { return x - (x >> 28) * m; }
// [0, 2m) -> [0, m) without a compare / select pair: x - m wraps to a huge value exactly when x < m
__device__ __forceinline__ uint32_t csub_min(uint32_t x, uint32_t m) { return min(x, x - m); }
// forward (Cooley-Tukey): bound grows by 2m per stage
__device__ __forceinline__ void ct_bfly(uint32_t& x, uint32_t& y, uint32_t w, uint32_t ws, uint32_t m) {
    const uint32_t t = shoup(y, w, ws, m);
    const uint32_t x0 = x;
    x = x0 + t;
    y = x0 + 2 * m - t;
}
// inverse (Gentleman-Sande, 1/2 folded per stage, src/core.cpp:445-472): the sum side grows by m/2 per stage,
// the product side is always < 2m.  8m - v keeps the difference positive (v < 8m) and, 8m being even and m odd,
// u + 8m - v has the parity of u + v, which is what the exact halving needs.
__device__ __forceinline__ void gs_bfly(uint32_t& u, uint32_t& v, uint32_t w, uint32_t ws, uint32_t m) {
    const uint32_t t = u + 8 * m - v;
    const uint32_t s = u + v;
    u = (s + ((s & 1u) ? m : 0u)) >> 1;
    v = shoup(t, w, ws, m);
}
__device__ __forceinline__ void ct2(uint32_t* lo, uint32_t* hi, int a, int b, uint4 tw) {
    ct_bfly(lo[a], lo[b], tw.x, tw.y, kP);
    ct_bfly(hi[a], hi[b], tw.z, tw.w, kB);
}
__device__ __forceinline__ void gs2(uint32_t* lo, uint32_t* hi, int a, int b, uint4 tw) {
    gs_bfly(lo[a], lo[b], tw.x, tw.y, kP);
    gs_bfly(hi[a], hi[b], tw.z, tw.w, kB);
}

// ---- register-resident passes ---------------------------------------------------------------------
// A pass runs NST consecutive stages on the 16 values of a thread; stage j pairs registers at distance D0 >> j, and the
// blocks of 2 * (D0 >> j) registers are the twiddle groups of that stage.  The twiddles of a pass arrive preloaded in
// stage-major order: stage 0's 16 / (2 D0) groups, then stage 1's, ...  (D0 = 8, 4 stages: 1 + 2 + 4 + 8 = 15; D0 = 4, 3
// stages: 2 + 4 + 8 = 14.)
struct Tw15 {
    uint4 t[15];
};
// twiddles of the pass whose first stage has index s0 (stage s uses table row 2^s + (coefficient index >> (11 - s)));
// `pre` = the thread-dependent part of that index at the pass's first stage
template <int D0, int NST>
__device__ __forceinline__ Tw15 tw_load(const uint4* tw, uint32_t s0, uint32_t pre) {
    Tw15 r;
    int o = 0;
#pragma unroll
    for (int j = 0; j < NST; j++) {
        const int groups = kV / (2 * (D0 >> j));
        const uint32_t base = (1u << (s0 + j)) + pre * groups;
#pragma unroll
        for (int g = 0; g < groups; g++) r.t[o + g] = tw[TWI(base + g)];
        o += groups;
    }
    return r;
}
template <int D0, int NST, int REDUCE_AFTER = -1>
__device__ __forceinline__ void ct_pass(uint32_t* lo, uint32_t* hi, const Tw15& w) {
    int o = 0;
#pragma unroll
    for (int j = 0; j < NST; j++) {
        const int d = D0 >> j, groups = kV / (2 * d);
#pragma unroll
        for (int g = 0; g < groups; g++)
#pragma unroll
            for (int i = 0; i < d; i++) ct2(lo, hi, g * 2 * d + i, g * 2 * d + i + d, w.t[o + g]);
        o += groups;
        if (j == REDUCE_AFTER) {
#pragma unroll
            for (int k = 0; k < kV; k++) {
                lo[k] = lazy_reduce(lo[k], kP);
                hi[k] = lazy_reduce(hi[k], kB);
            }
        }
    }
}
// the inverse runs the stages of a pass in the opposite order (distance 1 first)
template <int D0, int NST>
__device__ __forceinline__ void gs_pass(uint32_t* lo, uint32_t* hi, const Tw15& w) {
    int o = 0;
#pragma unroll
    for (int j = 0; j < NST; j++) o += kV / (2 * (D0 >> j));
#pragma unroll
    for (int j = NST - 1; j >= 0; j--) {
        const int d = D0 >> j, groups = kV / (2 * d);
        o -= groups;
#pragma unroll
        for (int g = 0; g < groups; g++)
#pragma unroll
            for (int i = 0; i < d; i++) gs2(lo, hi, g * 2 * d + i, g * 2 * d + i + d, w.t[o + g]);
    }
}

// ---- LDS tile -----------------------------------------------------------------------------------
// 2048 packed coefficients, one word of padding per 16: a thread's 16 consecutive words (pass C) then sit at a stride of 17
// words = 34 banks, which spreads the 32 lanes of a ds_read_b64 / ds_write_b64 group over all 64 banks, and the 8-word runs
// of pass B land 136 words apart (four runs per group on four different 16-bank windows).  One mapping for both exchanges,
// so a thread writes its pass-B results to the very words it read and no second barrier is needed.
constexpr uint32_t kLdsWords = kN + (kN >> 4);
__device__ __forceinline__ uint32_t lds_ix(uint32_t i) { return i + (i >> 4); }

// coefficient index held in register k by thread `tid` in each pass
__device__ __forceinline__ uint32_t ix_a(uint32_t tid, int k) { return tid + kNttThreads * k; }                        // stages 0..3: bits 10..7 in k
__device__ __forceinline__ uint32_t ix_b(uint32_t tid, int k) { return ((tid >> 3) << 7) | ((uint32_t)k << 3) | (tid & 7u); }  // stages 4..7: bits 6..3
__device__ __forceinline__ uint32_t ix_c(uint32_t tid, int k) { return (tid << 4) | (uint32_t)k; }                      // stages 8..10: bits 2..0 (slot order)

template <uint32_t (*IX)(uint32_t, int)>
__device__ __forceinline__ void lds_put(uint64_t* sh, uint32_t tid, const uint32_t* lo, const uint32_t* hi) {
#pragma unroll
    for (int k = 0; k < kV; k++) sh[lds_ix(IX(tid, k))] = pack(lo[k], hi[k]);
}
template <uint32_t (*IX)(uint32_t, int)>
__device__ __forceinline__ void lds_get(const uint64_t* sh, uint32_t tid, uint32_t* lo, uint32_t* hi) {
#pragma unroll
    for (int k = 0; k < kV; k++) {
        const uint64_t v = sh[lds_ix(IX(tid, k))];
        lo[k] = lo32(v);
        hi[k] = hi32(v);
    }
}

// ---- PK polynomial <-> the 16 slots 16*tid .. 16*tid+15 of a thread ----------------------------------------------------
// common.h pk_pos: slot s sits at ((s & 7) >> 1) * 512 + 2 * (s >> 3) + (s & 1), so for q < 4 the four words at q*512 + 4*tid
// hold this thread's registers {2q, 2q+1, 8+2q, 9+2q}: eight 16-byte accesses, adjacent in pairs.
typedef unsigned long long pk_u64x2 __attribute__((ext_vector_type(2)));
template <bool NT = false>
__device__ __forceinline__ void pk_load16(const uint64_t* poly, uint32_t tid, uint64_t (&v)[kV]) {
    const pk_u64x2* src = reinterpret_cast<const pk_u64x2*>(poly) + 2u * tid;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const pk_u64x2 x = NT ? __builtin_nontemporal_load(src + q * 256) : src[q * 256];
        const pk_u64x2 y = NT ? __builtin_nontemporal_load(src + q * 256 + 1) : src[q * 256 + 1];
        v[2 * q] = x.x;
        v[2 * q + 1] = x.y;
        v[8 + 2 * q] = y.x;
        v[9 + 2 * q] = y.y;
    }
}
template <bool NT = false>
__device__ __forceinline__ void pk_store16(uint64_t* poly, uint32_t tid, const uint64_t (&v)[kV]) {
    pk_u64x2* dst = reinterpret_cast<pk_u64x2*>(poly) + 2u * tid;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const pk_u64x2 x = {v[2 * q], v[2 * q + 1]}, y = {v[8 + 2 * q], v[9 + 2 * q]};
        if (NT) {
            __builtin_nontemporal_store(x, dst + q * 256);
            __builtin_nontemporal_store(y, dst + q * 256 + 1);
        } else {
            dst[q * 256] = x;
            dst[q * 256 + 1] = y;
        }
    }
}
__device__ __forceinline__ void pk_unpack16(const uint64_t (&v)[kV], uint32_t* lo, uint32_t* hi) {
#pragma unroll
    for (int k = 0; k < kV; k++) {
        lo[k] = lo32(v[k]);
        hi[k] = hi32(v[k]);
    }
}
__device__ __forceinline__ void pk_pack16(const uint32_t* lo, const uint32_t* hi, uint64_t (&v)[kV]) {
#pragma unroll
    for (int k = 0; k < kV; k++) v[k] = pack(lo[k], hi[k]);
}
// physical PK position of register k of thread tid (slot 16*tid + k)
__device__ __forceinline__ uint32_t pk_pos_tk16(uint32_t tid, uint32_t k) { return pk_pos(16u * tid + k); }

// Forward transform of the 2048 coefficients held as (lo,hi)[k] <-> index ix_a(tid,k) = tid + 128k, values < 2m.
// On return (lo,hi)[k] <-> slot ix_c(tid,k) = 16*tid + k, canonical in [0, m).
// Bounds: < 2m in, +2m per stage: < 14m after the 6th stage (the second of pass B) -> reduced to < 2m; < 12m at the end.
// Pass A's twiddles (rows 1..15) are the same for every thread and travel through the scalar cache; those of passes B and C
// are requested one pass ahead, so their latency hides behind the arithmetic and the exchange in between.
__device__ __forceinline__ void ntt_forward_block(uint32_t* lo, uint32_t* hi, uint64_t* sh, const uint4* tw, uint32_t tid) {
    const Tw15 wb = tw_load<8, 4>(tw, 4, tid >> 3);
    const Tw15 wa = tw_load<8, 4>(tw, 0, 0);
    ct_pass<8, 4>(lo, hi, wa);
    lds_put<ix_a>(sh, tid, lo, hi);
    __syncthreads();
    lds_get<ix_b>(sh, tid, lo, hi);
    ct_pass<8, 4, 1>(lo, hi, wb);
    const Tw15 wc = tw_load<4, 3>(tw, 8, tid);
    lds_put<ix_b>(sh, tid, lo, hi);
    __syncthreads();
    lds_get<ix_c>(sh, tid, lo, hi);
    ct_pass<4, 3>(lo, hi, wc);
#pragma unroll
    for (int k = 0; k < kV; k++) {
        lo[k] = csub_min(lazy_reduce(lo[k], kP), kP);
        hi[k] = csub_min(lazy_reduce(hi[k], kB), kB);
    }
}

// Inverse transform: in (lo,hi)[k] <-> slot ix_c(tid,k), values in [0, 2m);
// out (lo,hi)[k] <-> coefficient ix_a(tid,k) = tid + 128k, canonical in [0, m).
// Bounds: sum side < 2m + 11 * m/2 = 7.5m, product side < 2m: every t = u + 8m - v is in (0, 15.5m).
__device__ __forceinline__ void ntt_inverse_block(uint32_t* lo, uint32_t* hi, uint64_t* sh, const uint4* tw, uint32_t tid) {
    const Tw15 wc = tw_load<4, 3>(tw, 8, tid);
    const Tw15 wb = tw_load<8, 4>(tw, 4, tid >> 3);
    gs_pass<4, 3>(lo, hi, wc);
    lds_put<ix_c>(sh, tid, lo, hi);
    __syncthreads();
    lds_get<ix_b>(sh, tid, lo, hi);
    gs_pass<8, 4>(lo, hi, wb);
    const Tw15 wa = tw_load<8, 4>(tw, 0, 0);
    lds_put<ix_b>(sh, tid, lo, hi);
    __syncthreads();
    lds_get<ix_a>(sh, tid, lo, hi);
    gs_pass<8, 4>(lo, hi, wa);
#pragma unroll
    for (int k = 0; k < kV; k++) {
        lo[k] = csub_min(lazy_reduce(lo[k], kP), kP);
        hi[k] = csub_min(lazy_reduce(hi[k], kB), kB);
    }
}

}  // namespace spiral
