// Negacyclic NTT / INTT over Z[x]/(x^2048+1) for the two 28-bit CRT primes, one 256-thread workgroup
// per polynomial, both limbs processed side by side (PK format, see common.h).
//
// Replaces ntt_forward / ntt_inverse (reference src/core.cpp:247-514).  Same transform and slot
// order (natural in -> bit-reversed out for the forward, the converse for the inverse) and the same
// twiddle table (regenerated from psi, see tables.cpp), but organised for CDNA4: every thread keeps
// 8 coefficients x 2 limbs in VGPRs and runs three radix-2 stages per pass with lazy u32 Harvey
// butterflies; passes exchange data through a 16 KiB LDS tile (3 exchanges for 11 stages).
// Outputs are canonical residues in [0, m).
// (16 coefficients per thread -- 128 threads per polynomial, passes of 4 + 4 + 3 stages, two LDS exchanges -- was measured in round 2 and lost:
// 4.5 % fewer VALU instructions but 114 VGPRs and twice the serial work per thread, 9 % slower on wide batches; HISTORY.md.)
#pragma once
#include "common.h"
#ifndef TWI
#define TWI(i) (i)
#endif
// The twiddle rows are fetched through a buffer descriptor (32-bit byte offsets against a wave-uniform base) rather than
// 64-bit flat addresses: the 20 per-thread row loads of a transform then need one shift each instead of a 64-bit add.
struct TwTable {
    __amdgpu_buffer_rsrc_t rs;
    __device__ __forceinline__ explicit TwTable(const uint4* tw) : rs(__builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(tw), 0, 2048 * 16, 0x00020000)) {}
    __device__ __forceinline__ uint4 operator[](uint32_t i) const {
        typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, i * 16u, 0, 0);
        return make_uint4(v.x, v.y, v.z, v.w);
    }
};
// tuning hook (tools/build_variants.sh): -DNTT_PRIO=n raises the wave's issue priority from the LDS stores of a pass up to the
// twiddle loads of the next one (the memory-pipe part of the exchange) and drops it to 0 for the butterflies
#ifdef NTT_PRIO
#define NTT_PRIO_HI() __builtin_amdgcn_s_setprio(NTT_PRIO)
#define NTT_PRIO_LO() __builtin_amdgcn_s_setprio(0)
#else
#define NTT_PRIO_HI() ((void)0)
#define NTT_PRIO_LO() ((void)0)
#endif
#ifdef NTT_ABLATE_BARRIER
#define NTT_SYNC() ((void)0)
#else
#define NTT_SYNC() __syncthreads()
#endif

namespace spiral {

struct Tables {
    const uint4* fwd;  // [2048] {W_p, W'_p, W_b, W'_b} indexed like the reference's forward rows (m + i)
    const uint4* inv;  // [2048] inverse rows (h + i): psi^-i unscaled, row 1 times N^-1, row 0 = N^-1 (tables.cpp)
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
#ifdef NTT_ABLATE_ALU
    const uint32_t t = y ^ w;
#else
    const uint32_t t = shoup(y, w, ws, m);
#endif
    const uint32_t x0 = x;
    x = x0 + t;
    y = x0 + 2 * m - t;
}
// inverse (Gentleman-Sande).  The reference halves in every stage (u' = (u + v)/2, twiddles carrying the 1/2,
// src/core.cpp:445-472); the 11 halvings are one multiplication by N^-1, which is exact modular arithmetic either way, so
// here the stages are unscaled -- u' = u + v, v' = (u - v) w with w = psi^-i -- and the LAST stage multiplies both outputs by
// N^-1 (the difference side through its twiddle, the sum side by the constant kept in row 0 of the table).  That removes
// the parity test, select, add and shift of every butterfly (4 of its 10 instructions); the price is that the sum side
// doubles per stage instead of growing by m/2, so the sum-side registers are range-reduced once per pass (see
// ntt_inverse_block).  VB: the difference is taken as u + VB m - v, VB m being at least the bound of v.
template <uint32_t VB>
__device__ __forceinline__ void gs_bfly(uint32_t& u, uint32_t& v, uint32_t w, uint32_t ws, uint32_t m) {
    const uint32_t t = u + VB * m - v;
    u = u + v;
#ifdef NTT_ABLATE_ALU
    v = t ^ w;
#else
    v = shoup(t, w, ws, m);
#endif
}
// last stage: both outputs scaled by N^-1 (n = {N^-1 mod p, its Shoup companion, N^-1 mod b, companion}; w already carries it)
template <uint32_t VB>
__device__ __forceinline__ void gs_bfly_last(uint32_t& u, uint32_t& v, uint32_t w, uint32_t ws, uint32_t n, uint32_t ns, uint32_t m) {
    const uint32_t t = u + VB * m - v;
    u = shoup(u + v, n, ns, m);
    v = shoup(t, w, ws, m);
}

__device__ __forceinline__ void ct2(uint32_t* lo, uint32_t* hi, int a, int b, uint4 tw) {
    ct_bfly(lo[a], lo[b], tw.x, tw.y, kP);
    ct_bfly(hi[a], hi[b], tw.z, tw.w, kB);
}
template <uint32_t VB>
__device__ __forceinline__ void gs2(uint32_t* lo, uint32_t* hi, int a, int b, uint4 tw) {
    gs_bfly<VB>(lo[a], lo[b], tw.x, tw.y, kP);
    gs_bfly<VB>(hi[a], hi[b], tw.z, tw.w, kB);
}
template <uint32_t VB>
__device__ __forceinline__ void gs2_last(uint32_t* lo, uint32_t* hi, int a, int b, uint4 tw, uint4 n) {
    gs_bfly_last<VB>(lo[a], lo[b], tw.x, tw.y, n.x, n.y, kP);
    gs_bfly_last<VB>(hi[a], hi[b], tw.z, tw.w, n.z, n.w, kB);
}

// three forward stages on 8 register-resident coefficients whose indices differ in the 3 bits the
// stages consume: distance 4, then 2, then 1 in register numbering
template <class TW>
__device__ __forceinline__ void ct_radix8(uint32_t* lo, uint32_t* hi, const TW& tw, uint32_t b0, uint32_t b1, uint32_t b2) {
    uint4 t0 = tw[TWI(b0)];
#pragma unroll
    for (int k = 0; k < 4; k++) ct2(lo, hi, k, k + 4, t0);
    uint4 t1a = tw[TWI(b1)], t1b = tw[TWI(b1 + 1)];
    ct2(lo, hi, 0, 2, t1a);
    ct2(lo, hi, 1, 3, t1a);
    ct2(lo, hi, 4, 6, t1b);
    ct2(lo, hi, 5, 7, t1b);
#pragma unroll
    for (int q = 0; q < 4; q++) ct2(lo, hi, 2 * q, 2 * q + 1, tw[TWI(b2 + q)]);
}
template <class TW>
__device__ __forceinline__ void ct_radix4x2(uint32_t* lo, uint32_t* hi, const TW& tw, uint32_t b1, uint32_t b2) {
    uint4 t1a = tw[TWI(b1)], t1b = tw[TWI(b1 + 1)];
    ct2(lo, hi, 0, 2, t1a);
    ct2(lo, hi, 1, 3, t1a);
    ct2(lo, hi, 4, 6, t1b);
    ct2(lo, hi, 5, 7, t1b);
#pragma unroll
    for (int q = 0; q < 4; q++) ct2(lo, hi, 2 * q, 2 * q + 1, tw[TWI(b2 + q)]);
}
// the 7 twiddles of a radix-8 pass (6 of a radix-4x2 pass), loaded ahead of the pass so that their latency hides
// behind the previous pass's arithmetic and the LDS exchange
struct Tw7 {
    uint4 t[7];
};
template <class TW>
__device__ __forceinline__ Tw7 tw_load8(const TW& tw, uint32_t b0, uint32_t b1, uint32_t b2) {
    Tw7 r;
    r.t[0] = tw[TWI(b0)];
    r.t[1] = tw[TWI(b1)];
    r.t[2] = tw[TWI(b1 + 1)];
#pragma unroll
    for (int q = 0; q < 4; q++) r.t[3 + q] = tw[TWI(b2 + q)];
    return r;
}
template <class TW>
__device__ __forceinline__ Tw7 tw_load4x2(const TW& tw, uint32_t b1, uint32_t b2) {
    Tw7 r;
    r.t[0] = make_uint4(0, 0, 0, 0);
    r.t[1] = tw[TWI(b1)];
    r.t[2] = tw[TWI(b1 + 1)];
#pragma unroll
    for (int q = 0; q < 4; q++) r.t[3 + q] = tw[TWI(b2 + q)];
    return r;
}
__device__ __forceinline__ void ct_radix8_pre(uint32_t* lo, uint32_t* hi, const Tw7& w) {
#pragma unroll
    for (int k = 0; k < 4; k++) ct2(lo, hi, k, k + 4, w.t[0]);
    ct2(lo, hi, 0, 2, w.t[1]);
    ct2(lo, hi, 1, 3, w.t[1]);
    ct2(lo, hi, 4, 6, w.t[2]);
    ct2(lo, hi, 5, 7, w.t[2]);
#pragma unroll
    for (int q = 0; q < 4; q++) ct2(lo, hi, 2 * q, 2 * q + 1, w.t[3 + q]);
}
__device__ __forceinline__ void ct_radix4x2_pre(uint32_t* lo, uint32_t* hi, const Tw7& w) {
    ct2(lo, hi, 0, 2, w.t[1]);
    ct2(lo, hi, 1, 3, w.t[1]);
    ct2(lo, hi, 4, 6, w.t[2]);
    ct2(lo, hi, 5, 7, w.t[2]);
#pragma unroll
    for (int q = 0; q < 4; q++) ct2(lo, hi, 2 * q, 2 * q + 1, w.t[3 + q]);
}

// inverse passes (stage distances 1, 2, 4 in register numbering), inputs < 2m.  Bounds in units of m after each stage:
//   distance 1: sums (0,2,4,6) < 4, products (1,3,5,7) < 2
//   distance 2: (0,4) < 8, (1,5) < 4, (2,3,6,7) < 2
//   distance 4: 0 < 16, 1 < 8, (2,3) < 4, (4..7) < 2            -- 16 m < 2^32 for both primes
__device__ __forceinline__ void gs_stage12(uint32_t* lo, uint32_t* hi, const Tw7& w) {
#pragma unroll
    for (int q = 0; q < 4; q++) gs2<2>(lo, hi, 2 * q, 2 * q + 1, w.t[3 + q]);
    gs2<4>(lo, hi, 0, 2, w.t[1]);
    gs2<2>(lo, hi, 1, 3, w.t[1]);
    gs2<4>(lo, hi, 4, 6, w.t[2]);
    gs2<2>(lo, hi, 5, 7, w.t[2]);
}
// [0, 16m) -> [0, 2m): lazy_reduce alone leaves up to 2.13 m for inputs beyond 14 m (prime b)
__device__ __forceinline__ uint32_t lazy_reduce16(uint32_t x, uint32_t m) {
    const uint32_t r = lazy_reduce(x, m);
    return min(r, r - 2 * m);
}
__device__ __forceinline__ void gs_radix8_pre(uint32_t* lo, uint32_t* hi, const Tw7& w) {
    gs_stage12(lo, hi, w);
    gs2<8>(lo, hi, 0, 4, w.t[0]);
    gs2<4>(lo, hi, 1, 5, w.t[0]);
    gs2<2>(lo, hi, 2, 6, w.t[0]);
    gs2<2>(lo, hi, 3, 7, w.t[0]);
    lo[0] = lazy_reduce16(lo[0], kP);
    hi[0] = lazy_reduce16(hi[0], kB);
#pragma unroll
    for (int k = 1; k < 4; k++) {
        lo[k] = lazy_reduce(lo[k], kP);
        hi[k] = lazy_reduce(hi[k], kB);
    }
}
// the last pass: its distance-4 stage applies N^-1 (n = row 0 of the table); every output is a Shoup product, < 2m
__device__ __forceinline__ void gs_radix8_last(uint32_t* lo, uint32_t* hi, const Tw7& w, uint4 n) {
    gs_stage12(lo, hi, w);
    gs2_last<8>(lo, hi, 0, 4, w.t[0], n);
    gs2_last<4>(lo, hi, 1, 5, w.t[0], n);
    gs2_last<2>(lo, hi, 2, 6, w.t[0], n);
    gs2_last<2>(lo, hi, 3, 7, w.t[0], n);
}
__device__ __forceinline__ void gs_radix4x2_pre(uint32_t* lo, uint32_t* hi, const Tw7& w) {
    gs_stage12(lo, hi, w);
#pragma unroll
    for (int k = 0; k < 8; k += 4) {  // registers 0, 4 < 8m and 1, 5 < 4m
        lo[k] = lazy_reduce(lo[k], kP);
        hi[k] = lazy_reduce(hi[k], kB);
        lo[k + 1] = lazy_reduce(lo[k + 1], kP);
        hi[k + 1] = lazy_reduce(hi[k + 1], kB);
    }
}

// ---- LDS tile -----------------------------------------------------------------------------------
// 2048 packed coefficients; +4 words of padding per 32 keeps the stride-32 and stride-8 access
// patterns of passes C and D off a single bank
#ifndef NTT_LDS_EXTRA_WORDS  // tuning only: extra LDS per workgroup, to limit the workgroups resident on a CU
#define NTT_LDS_EXTRA_WORDS 0
#endif
constexpr uint32_t kLdsWords = kN + (kN >> 5) * 4 + NTT_LDS_EXTRA_WORDS;
__device__ __forceinline__ uint32_t lds_ix(uint32_t i) { return i + ((i >> 5) << 2); }

// coefficient index held in register k by thread `tid` in each pass
__device__ __forceinline__ uint32_t ix_a(uint32_t tid, int k) { return tid + 256u * k; }
__device__ __forceinline__ uint32_t ix_b(uint32_t tid, int k) { return ((tid >> 5) << 8) + 32u * k + (tid & 31u); }
__device__ __forceinline__ uint32_t ix_c(uint32_t tid, int k) { return ((tid >> 2) << 5) + 4u * k + (tid & 3u); }
__device__ __forceinline__ uint32_t ix_d(uint32_t tid, int k) { return 8u * tid + k; }

template <uint32_t (*IX)(uint32_t, int)>
__device__ __forceinline__ void lds_put(uint64_t* sh, uint32_t tid, const uint32_t* lo, const uint32_t* hi) {
#ifdef NTT_ABLATE_LDS
    return;
#endif
#pragma unroll
    for (int k = 0; k < 8; k++) sh[lds_ix(IX(tid, k))] = pack(lo[k], hi[k]);
}
template <uint32_t (*IX)(uint32_t, int)>
__device__ __forceinline__ void lds_get(const uint64_t* sh, uint32_t tid, uint32_t* lo, uint32_t* hi) {
#ifdef NTT_ABLATE_LDS
    return;
#endif
#pragma unroll
    for (int k = 0; k < 8; k++) {
        uint64_t v = sh[lds_ix(IX(tid, k))];
        lo[k] = lo32(v);
        hi[k] = hi32(v);
    }
}

// ---- PK polynomial <-> the 8 slots of a thread (common.h pk_pos): four 16-byte accesses -------------------------
typedef unsigned long long pk_u64x2 __attribute__((ext_vector_type(2)));
template <bool NT = false>
__device__ __forceinline__ void pk_load8(const uint64_t* poly, uint32_t tid, uint64_t (&v)[8]) {
    const pk_u64x2* src = reinterpret_cast<const pk_u64x2*>(poly) + tid;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const pk_u64x2 x = NT ? __builtin_nontemporal_load(src + q * 256) : src[q * 256];
        v[2 * q] = x.x;
        v[2 * q + 1] = x.y;
    }
}
template <bool NT = false>
__device__ __forceinline__ void pk_store8(uint64_t* poly, uint32_t tid, const uint64_t (&v)[8]) {
    pk_u64x2* dst = reinterpret_cast<pk_u64x2*>(poly) + tid;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const pk_u64x2 x = {v[2 * q], v[2 * q + 1]};
        if (NT)
            __builtin_nontemporal_store(x, dst + q * 256);
        else
            dst[q * 256] = x;
    }
}
__device__ __forceinline__ void pk_unpack8(const uint64_t (&v)[8], uint32_t* lo, uint32_t* hi) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
        lo[k] = lo32(v[k]);
        hi[k] = hi32(v[k]);
    }
}
__device__ __forceinline__ void pk_pack8(const uint32_t* lo, const uint32_t* hi, uint64_t (&v)[8]) {
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = pack(lo[k], hi[k]);
}

// Forward transform of the 2048 coefficients held as (lo,hi)[k] <-> index ix_a(tid,k), values < 2m.
// On return (lo,hi)[k] <-> slot ix_d(tid,k) = 8*tid + k, canonical in [0, m) -- or, with CANONICAL = false, only reduced
// to [0, 2m): what a transform of gadget digits may leave when its only readers are the u64 multiply-accumulate kernels
// (a product of a canonical residue with a value < 2m is < 2^57; up to 127 of them fit the accumulator).
// Bounds: < 2m in, +2m per stage: < 14m after passes A and B (6 stages).  Pass C's first stage multiplies registers 4..7
// (any u32 is a valid Shoup operand and the product is < 2m) and only adds to registers 0..3, so only those four are
// reduced (to < 2m) after the exchange; from there +2m per stage again: < 12m after the remaining 5 stages.
// Pass A's twiddles (rows 1..7) are wave-uniform and come through the scalar cache; the per-thread rows of the other
// passes are fetched through a buffer descriptor, one pass ahead of their use.
// [0, 2m) -> [0, m) for a thread's 8 x 2 values
__device__ __forceinline__ void canonicalize8(uint32_t* lo, uint32_t* hi) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
        lo[k] = csub_min(lo[k], kP);
        hi[k] = csub_min(hi[k], kB);
    }
}
template <bool CANONICAL = true>
__device__ __forceinline__ void ntt_forward_block(uint32_t* lo, uint32_t* hi, uint64_t* sh, const uint4* tw, uint32_t tid) {
    const TwTable tb(tw);
    Tw7 wb = tw_load8(tb, 8 + (tid >> 5), 16 + 2 * (tid >> 5), 32 + 4 * (tid >> 5));
#ifndef NTT_ABLATE_AB  // ablation (tools/build_variants.sh): the first six stages' butterflies off, exchanges kept -- the bound on moving them off the vector ALU
    ct_radix8(lo, hi, tw, 1, 2, 4);
#endif
    NTT_PRIO_HI();
    lds_put<ix_a>(sh, tid, lo, hi);
    NTT_SYNC();
    lds_get<ix_b>(sh, tid, lo, hi);
    Tw7 wc = tw_load8(tb, 64 + (tid >> 2), 128 + 2 * (tid >> 2), 256 + 4 * (tid >> 2));
    NTT_PRIO_LO();
#ifndef NTT_ABLATE_AB
    ct_radix8_pre(lo, hi, wb);
#endif
    NTT_PRIO_HI();
    lds_put<ix_b>(sh, tid, lo, hi);
    NTT_SYNC();
    lds_get<ix_c>(sh, tid, lo, hi);
    Tw7 wd = tw_load4x2(tb, 512 + 2 * tid, 1024 + 4 * tid);
    NTT_PRIO_LO();
#pragma unroll
    for (int k = 0; k < 4; k++) {
        lo[k] = lazy_reduce(lo[k], kP);
        hi[k] = lazy_reduce(hi[k], kB);
    }
    ct_radix8_pre(lo, hi, wc);
    NTT_PRIO_HI();
#ifndef NTT_ABLATE_QUAD_EXCHANGE  // ablation: the one exchange that stays inside a quad of lanes (C <-> D) removed outright -- the bound on replacing it by DPP shuffles
    lds_put<ix_c>(sh, tid, lo, hi);
    NTT_SYNC();
    lds_get<ix_d>(sh, tid, lo, hi);
#endif
    NTT_PRIO_LO();
    ct_radix4x2_pre(lo, hi, wd);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        lo[k] = lazy_reduce(lo[k], kP);
        hi[k] = lazy_reduce(hi[k], kB);
        if constexpr (CANONICAL) {
            lo[k] = csub_min(lo[k], kP);
            hi[k] = csub_min(hi[k], kB);
        }
    }
}

// Inverse transform: in (lo,hi)[k] <-> slot ix_d(tid,k), values in [0, 2m);
// out (lo,hi)[k] <-> coefficient ix_a(tid,k) = tid + 256k, in [0, 2m) (CANONICAL: [0, m)).
// `tw` is the device inverse table (tables.cpp): row 0 = N^-1, row 1 = psi^-(N/2) N^-1, rows >= 2 = psi^-i (no halving).
// Every pass starts from values < 2m and range-reduces its sum-side registers before the exchange (gs_radix*_pre).
template <bool CANONICAL = true>
__device__ __forceinline__ void ntt_inverse_block(uint32_t* lo, uint32_t* hi, uint64_t* sh, const uint4* tw, uint32_t tid) {
    // as in the forward transform: the next pass's twiddles are in flight during the current pass
    const TwTable tb(tw);
    Tw7 wd = tw_load4x2(tb, 512 + 2 * tid, 1024 + 4 * tid);
    Tw7 wc = tw_load8(tb, 64 + (tid >> 2), 128 + 2 * (tid >> 2), 256 + 4 * (tid >> 2));
    gs_radix4x2_pre(lo, hi, wd);
    NTT_PRIO_HI();
#ifndef NTT_ABLATE_QUAD_EXCHANGE
    lds_put<ix_d>(sh, tid, lo, hi);
    __syncthreads();
    lds_get<ix_c>(sh, tid, lo, hi);
#endif
    Tw7 wb = tw_load8(tb, 8 + (tid >> 5), 16 + 2 * (tid >> 5), 32 + 4 * (tid >> 5));
    NTT_PRIO_LO();
    gs_radix8_pre(lo, hi, wc);
    NTT_PRIO_HI();
    lds_put<ix_c>(sh, tid, lo, hi);
    __syncthreads();
    lds_get<ix_b>(sh, tid, lo, hi);
    Tw7 wa = tw_load8(tw, 1, 2, 4);
    const uint4 ninv = tw[0];
    NTT_PRIO_LO();
    gs_radix8_pre(lo, hi, wb);
    NTT_PRIO_HI();
    lds_put<ix_b>(sh, tid, lo, hi);
    __syncthreads();
    lds_get<ix_a>(sh, tid, lo, hi);
    NTT_PRIO_LO();
    gs_radix8_last(lo, hi, wa, ninv);
    if constexpr (CANONICAL) canonicalize8(lo, hi);
}

// Two polynomials per workgroup pass, one twiddle fetch: the per-thread twiddle rows (20 of a transform's 28 vector-memory
// accesses) depend only on the thread, not on the polynomial, so a workgroup that transforms two polynomials side by side --
// two register sets, two LDS tiles, the same Tw7 -- halves the twiddle traffic per transform and gives every wave two
// independent butterfly chains to interleave.  Same arithmetic, bounds and results as the one-polynomial blocks above.
template <bool CANONICAL = true>
__device__ __forceinline__ void ntt_forward_block2(uint32_t* lo0, uint32_t* hi0, uint32_t* lo1, uint32_t* hi1, uint64_t* sh0, uint64_t* sh1,
                                                   const uint4* tw, uint32_t tid) {
    const TwTable tb(tw);
    Tw7 wb = tw_load8(tb, 8 + (tid >> 5), 16 + 2 * (tid >> 5), 32 + 4 * (tid >> 5));
#ifndef NTT_ABLATE_AB
    {
        const Tw7 wa = tw_load8(tw, 1, 2, 4);
        ct_radix8_pre(lo0, hi0, wa);
        ct_radix8_pre(lo1, hi1, wa);
    }
#endif
    lds_put<ix_a>(sh0, tid, lo0, hi0);
    lds_put<ix_a>(sh1, tid, lo1, hi1);
    NTT_SYNC();
    lds_get<ix_b>(sh0, tid, lo0, hi0);
    lds_get<ix_b>(sh1, tid, lo1, hi1);
    Tw7 wc = tw_load8(tb, 64 + (tid >> 2), 128 + 2 * (tid >> 2), 256 + 4 * (tid >> 2));
#ifndef NTT_ABLATE_AB
    ct_radix8_pre(lo0, hi0, wb);
    ct_radix8_pre(lo1, hi1, wb);
#endif
    lds_put<ix_b>(sh0, tid, lo0, hi0);
    lds_put<ix_b>(sh1, tid, lo1, hi1);
    NTT_SYNC();
    lds_get<ix_c>(sh0, tid, lo0, hi0);
    lds_get<ix_c>(sh1, tid, lo1, hi1);
    Tw7 wd = tw_load4x2(tb, 512 + 2 * tid, 1024 + 4 * tid);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        lo0[k] = lazy_reduce(lo0[k], kP);
        hi0[k] = lazy_reduce(hi0[k], kB);
        lo1[k] = lazy_reduce(lo1[k], kP);
        hi1[k] = lazy_reduce(hi1[k], kB);
    }
    ct_radix8_pre(lo0, hi0, wc);
    ct_radix8_pre(lo1, hi1, wc);
#ifndef NTT_ABLATE_QUAD_EXCHANGE
    lds_put<ix_c>(sh0, tid, lo0, hi0);
    lds_put<ix_c>(sh1, tid, lo1, hi1);
    NTT_SYNC();
    lds_get<ix_d>(sh0, tid, lo0, hi0);
    lds_get<ix_d>(sh1, tid, lo1, hi1);
#endif
    ct_radix4x2_pre(lo0, hi0, wd);
    ct_radix4x2_pre(lo1, hi1, wd);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        lo0[k] = lazy_reduce(lo0[k], kP);
        hi0[k] = lazy_reduce(hi0[k], kB);
        lo1[k] = lazy_reduce(lo1[k], kP);
        hi1[k] = lazy_reduce(hi1[k], kB);
        if constexpr (CANONICAL) {
            lo0[k] = csub_min(lo0[k], kP);
            hi0[k] = csub_min(hi0[k], kB);
            lo1[k] = csub_min(lo1[k], kP);
            hi1[k] = csub_min(hi1[k], kB);
        }
    }
}
// CRT lift of a coefficient whose residues are (x mod p canonical, y mod b in [0, 2b)) to [0, Q): Garner form as
// common.h crt_compose (src/poly.cpp:344-353), the product by p^-1 mod b as a Shoup product so that it takes three
// multiplies instead of a 64-bit remainder, and the b-side consumed lazily.
constexpr uint32_t kPinvBShoup = 1676084570u;  // floor(kPinvB * 2^32 / kB)
__device__ __forceinline__ uint64_t crt_compose_lazy(uint32_t x, uint32_t y2) {
    const uint32_t d = y2 + 2 * kB - x;  // == y - x (mod b), positive: x < p < 2b
    const uint32_t k = csub_min(shoup(d, kPinvB, kPinvBShoup, kB), kB);
    return (uint64_t)x + (uint64_t)kP * k;
}
// the inverse transform's lazy outputs -> CRT-lifted coefficients
__device__ __forceinline__ void crt_lift8(const uint32_t* lo, const uint32_t* hi, uint64_t (&v)[8]) {
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = crt_compose_lazy(csub_min(lo[r], kP), hi[r]);
}

}  // namespace spiral
