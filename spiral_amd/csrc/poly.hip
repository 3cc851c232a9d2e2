// Pointwise polynomial kernels on PK (packed NTT) and RAW buffers, gfx950.
// One thread per NTT slot / coefficient, 256-thread workgroups, 8 workgroups per polynomial.
// All NTT-domain outputs are canonical residues in [0, m).
#include <cstdlib>
#include "common.h"
#include "kernels.h"

namespace spiral {

// w*y mod m in [0, 2m) from the Shoup companion ws = floor(w * 2^32 / m), any y < 2^32 (as ntt_device.h shoup)
__device__ __forceinline__ uint32_t shoup32(uint32_t y, uint32_t w, uint32_t ws, uint32_t m) { return w * y - __umulhi(y, ws) * m; }

constexpr uint32_t kTpb = 256;
constexpr uint32_t kBpp = kN / kTpb;  // blocks per polynomial

struct Acc2 {
    uint64_t lo = 0, hi = 0;
    __device__ __forceinline__ void mac(uint64_t a, uint64_t b) {
        lo += (uint64_t)lo32(a) * lo32(b);  // u64 accumulation without intermediate reduce,
        hi += (uint64_t)hi32(a) * hi32(b);  // as src/poly.cpp:62 (<= 256 terms of < 2^56)
    }
    __device__ __forceinline__ uint64_t reduced() const { return pack(mod_p(lo), mod_b(hi)); }
};
__device__ __forceinline__ uint64_t add_pk(uint64_t a, uint64_t b) {  // canonical operands
    return pack(csub(lo32(a) + lo32(b), kP), csub(hi32(a) + hi32(b), kB));
}

// ---- generic MatPoly multiply (src/poly.cpp:34-78) ---------------------------------------------------
__global__ __launch_bounds__(kTpb) void matmul_kernel(MatmulParams p) {
    const uint32_t z = blockIdx.x * kTpb + threadIdx.x;
    const uint32_t rc = blockIdx.y, r = rc / p.cs, c = rc - r * p.cs, bt = blockIdx.z;
    const uint64_t* a = p.a + ((size_t)bt * p.a_batch + (size_t)r * p.ms) * kN + z;
    const uint64_t* b = p.b + ((size_t)bt * p.b_batch + c) * kN + z;
    Acc2 acc;
    for (uint32_t m = 0; m < p.ms; m++) acc.mac(a[(size_t)m * kN], b[(size_t)m * p.cs * kN]);
    p.out[((size_t)bt * p.out_batch + rc) * kN + z] = acc.reduced();
}
void launch_matmul(const MatmulParams& p, uint32_t batch, hipStream_t s) {
    if (batch == 0) return;
    hipLaunchKernelGGL(matmul_kernel, dim3(kBpp, p.rs * p.cs, batch), dim3(kTpb), 0, s, p);
}

// ---- fold product (cpu_mul_query_by_ct x2 + add, src/spiral.cpp:464-582, 1361-1383) ------------------
// out[i][r][c] = sum_{mm < K} key[r][mm] * D[i][mm][c], K = 2*m2 (Q_neg half then Q half).  One workgroup =
// 64 slots x 4 k-groups; every D word feeds 3 rows and every key word 2 columns; partial sums meet in LDS.
struct FoldMacParamsCore {
    const uint64_t* key;  // [3][K]
    const uint64_t* d;    // [np][K][2]
    uint64_t* out;        // [np][3][2]
    uint32_t K;
    uint32_t ks;          // polynomials between key rows (>= K)
    const uint64_t* add;  // optional addend [np][3][2] (the pair form: out = C[i] + Q * D'), fields any u32
};
template <class L>
struct FoldMacParamsT : FoldMacParamsCore {
    using Core = FoldMacParamsCore;
    using NoLanesT = FoldMacParamsT<NoLanes>;
    L lanes;          // every pointer per query lane (each query folds with its own keys)
};
using FoldMacParams = FoldMacParamsT<Lanes>;
// B ciphertexts per workgroup share every key word loaded (the key is common to all ciphertexts of a round)
template <uint32_t B, class L>
__global__ __launch_bounds__(kTpb) void fold_mac_kernel(FoldMacParamsT<L> p) {
    __shared__ uint64_t sh[3][64][12 * B];
    const uint32_t zz = threadIdx.x & 63u, kg = threadIdx.x >> 6, z = blockIdx.x * 64u + zz, i0 = blockIdx.y * B;
    {
        const int64_t lane = p.lanes.here();
        lane_shift(p.key, lane);
        lane_shift(p.d, lane);
        lane_shift(p.out, lane);
        lane_shift(p.add, lane);
    }
    const uint64_t* dp = p.d + (size_t)i0 * p.K * 2 * kN + z;
    const uint64_t* kp = p.key + z;
    Acc2 acc[B][3][2];
    if (p.add != nullptr && kg == 0) {  // requested before the product loop
#pragma unroll
        for (uint32_t b = 0; b < B; b++)
#pragma unroll
            for (uint32_t rc = 0; rc < 6; rc++) {
                const uint64_t a = p.add[((size_t)(i0 + b) * 6 + rc) * kN + z];
                acc[b][rc >> 1][rc & 1].lo = lo32(a);
                acc[b][rc >> 1][rc & 1].hi = hi32(a);
            }
    }
    if constexpr (B == 1) {
        // The latency-bound rounds (np < 16): "request all 12 terms of the k-group (K = 6 t_GSW = 48), then multiply".  With the loads inside
        // a loop whose every iteration may be the last, the compiler waits for each term before requesting the next -- 12 dependent
        // round trips per thread.  Indices beyond K are clamped for the request and their products skipped.  7.0 / 6.3 / 6.3 / 6.2 ->
        // 6.6 / 5.6 / 5.5 / 5.4 us for the four narrow rounds of config 2.  (The same rewrite of the bandwidth-bound B = 2 rounds and of
        // the expansion's product kernels costs occupancy and measured 1-2 us SLOWER per launch: they keep the plain loop.)
        constexpr uint32_t U = 12;
        for (uint32_t m0 = kg; m0 < p.K; m0 += 4 * U) {
            uint64_t kv[U][3], dv[U][2];
#pragma unroll
            for (uint32_t u = 0; u < U; u++) {
                const uint32_t mm = min(m0 + 4 * u, p.K - 1);
#pragma unroll
                for (uint32_t r = 0; r < 3; r++) kv[u][r] = kp[((size_t)r * p.ks + mm) * kN];
                dv[u][0] = __builtin_nontemporal_load(&dp[(size_t)mm * 2 * kN]);
                dv[u][1] = __builtin_nontemporal_load(&dp[((size_t)mm * 2 + 1) * kN]);
            }
#pragma unroll
            for (uint32_t u = 0; u < U; u++) {
                if (m0 + 4 * u < p.K) {
#pragma unroll
                    for (uint32_t r = 0; r < 3; r++) {
                        acc[0][r][0].mac(kv[u][r], dv[u][0]);
                        acc[0][r][1].mac(kv[u][r], dv[u][1]);
                    }
                }
            }
        }
    } else {
#pragma unroll 4
    for (uint32_t mm = kg; mm < p.K; mm += 4) {
        uint64_t kv[3];
#pragma unroll
        for (uint32_t r = 0; r < 3; r++) kv[r] = kp[((size_t)r * p.ks + mm) * kN];
#pragma unroll
        for (uint32_t b = 0; b < B; b++) {
#ifndef MAC_PLAIN_LOADS  // streamed operand: read once, must not push the shared W / key rows out of L2 (-17 us on expand + convert)
            const uint64_t d0 = __builtin_nontemporal_load(&dp[((size_t)b * p.K + mm) * 2 * kN]), d1 = __builtin_nontemporal_load(&dp[(((size_t)b * p.K + mm) * 2 + 1) * kN]);
#else
            const uint64_t d0 = dp[((size_t)b * p.K + mm) * 2 * kN], d1 = dp[(((size_t)b * p.K + mm) * 2 + 1) * kN];
#endif
#pragma unroll
            for (uint32_t r = 0; r < 3; r++) {
                acc[b][r][0].mac(kv[r], d0);
                acc[b][r][1].mac(kv[r], d1);
            }
        }
    }
    }
    if (kg > 0) {
#pragma unroll
        for (uint32_t b = 0; b < B; b++)
#pragma unroll
            for (uint32_t r = 0; r < 3; r++)
#pragma unroll
                for (uint32_t c = 0; c < 2; c++) {
                    sh[kg - 1][zz][b * 12 + (r * 2 + c) * 2] = acc[b][r][c].lo;
                    sh[kg - 1][zz][b * 12 + (r * 2 + c) * 2 + 1] = acc[b][r][c].hi;
                }
    }
    __syncthreads();
    if (kg == 0) {
#pragma unroll
        for (uint32_t b = 0; b < B; b++)
#pragma unroll
            for (uint32_t r = 0; r < 3; r++)
#pragma unroll
                for (uint32_t c = 0; c < 2; c++) {
                    Acc2 a = acc[b][r][c];
#pragma unroll
                    for (int q = 0; q < 3; q++) {  // K <= 256 terms in total (m2 * q fits u64, src/spiral.cpp:465)
                        a.lo += sh[q][zz][b * 12 + (r * 2 + c) * 2];
                        a.hi += sh[q][zz][b * 12 + (r * 2 + c) * 2 + 1];
                    }
                    p.out[((size_t)(i0 + b) * 6 + r * 2 + c) * kN + z] = a.reduced();
                }
    }
}
// The reference's two-product round from the matrices Q alone (the resident server keeps no Q_neg): with Q_neg = G2 - Q slot by slot (src/spiral.cpp:2361-2379)
//     Q_neg D_L + Q D_H  =  G2 D_L + Q (D_H - D_L),      (G2 D_L)[r][c] = sum_k 2^(bits k) D_L[3 k + r][c]      (G2[r][3 k + r'] = 2^(bits k) iff r' == r)
// in exact arithmetic mod p and mod b: the same canonical residues as the stored-Q_neg form, for any gadget dimension (no recomposition is assumed: D_L, D_H are
// whatever digits the round's loader produced).  The fallback form -- fold_root's first round, the stage API's fold() without transform-domain words,
// option fold_pair = 0, gadgets without the pair identity.
// d: the two-product operand layout D[i][(L | H) half][m2 rows][2 columns]; its words may be lazy ([0, 2m)): canonicalised on load.
template <class L>
__global__ __launch_bounds__(kTpb) void fold_mac_two_kernel(FoldMacParamsT<L> p, uint32_t ell, uint32_t bits) {
    // 64 slots x 4 k-groups per workgroup as fold_mac_kernel<1> -- these rounds are narrow (fold_root's first round holds G / 2 <= 8 ciphertexts) and
    // latency-bound, so a k-group requests ALL its terms' operands before the first multiply (terms beyond m2 are clamped and skipped)
    __shared__ uint64_t sh[3][64][12];
    const uint32_t zz = threadIdx.x & 63u, kg = threadIdx.x >> 6, z = blockIdx.x * 64u + zz, i = blockIdx.y, m2 = p.K;
    {
        const int64_t lane = p.lanes.here();
        lane_shift(p.key, lane);
        lane_shift(p.d, lane);
        lane_shift(p.out, lane);
    }
    const uint64_t* dl = p.d + (size_t)i * 2 * m2 * 2 * kN + z;
    const uint64_t* dh = dl + (size_t)m2 * 2 * kN;
    const uint64_t* kp = p.key + z;
    Acc2 acc[3][2];
    constexpr uint32_t U = 6;  // terms in flight per k-group: m2 = 3 ell <= 24 for ell <= 8 in one trip
    for (uint32_t m0 = kg; m0 < m2; m0 += 4 * U) {
        uint64_t kv[U][3], lw[U][2], hw[U][2];
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            const uint32_t mm = min(m0 + 4 * u, m2 - 1);
#pragma unroll
            for (uint32_t r = 0; r < 3; r++) kv[u][r] = kp[((size_t)r * p.ks + mm) * kN];
#pragma unroll
            for (uint32_t c = 0; c < 2; c++) {
                lw[u][c] = dl[((size_t)mm * 2 + c) * kN];
                hw[u][c] = dh[((size_t)mm * 2 + c) * kN];
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            const uint32_t mm = m0 + 4 * u;
            if (mm < m2) {
                uint64_t lv[2], df[2];
#pragma unroll
                for (uint32_t c = 0; c < 2; c++) {
                    const uint32_t lp = csub(lo32(lw[u][c]), kP), lb = csub(hi32(lw[u][c]), kB), hp = csub(lo32(hw[u][c]), kP), hb = csub(hi32(hw[u][c]), kB);
                    lv[c] = pack(lp, lb);
                    df[c] = pack(csub(hp + kP - lp, kP), csub(hb + kB - lb, kB));  // D_H - D_L, canonical
                }
#pragma unroll
                for (uint32_t r = 0; r < 3; r++) {
                    acc[r][0].mac(kv[u][r], df[0]);
                    acc[r][1].mac(kv[u][r], df[1]);
                }
                const uint32_t k = mm / 3u, r = mm - 3u * k, sh_ = bits * k;  // row 3 k + r of D_L is digit k of the ciphertext's row r
                const uint64_t g2 = sh_ < 64u ? pack(mod_p(1ull << (sh_ & 63u)), mod_b(1ull << (sh_ & 63u))) : 0ull;
#pragma unroll
                for (uint32_t rr = 0; rr < 3; rr++)  // (a select per row instead of a dynamically indexed accumulator)
                    if (rr == r) {
                        acc[rr][0].mac(g2, lv[0]);
                        acc[rr][1].mac(g2, lv[1]);
                    }
            }
        }
    }
    (void)ell;  // (m2 = 3 ell: at most 4 ell <= 112 canonical products per accumulator over the four k-groups, far inside the u64 bound)
    if (kg > 0) {
#pragma unroll
        for (uint32_t rc = 0; rc < 6; rc++) {
            sh[kg - 1][zz][rc * 2] = acc[rc >> 1][rc & 1].lo;
            sh[kg - 1][zz][rc * 2 + 1] = acc[rc >> 1][rc & 1].hi;
        }
    }
    __syncthreads();
    if (kg == 0) {
#pragma unroll
        for (uint32_t rc = 0; rc < 6; rc++) {
            Acc2 a = acc[rc >> 1][rc & 1];
#pragma unroll
            for (int q = 0; q < 3; q++) {
                a.lo += sh[q][zz][rc * 2];
                a.hi += sh[q][zz][rc * 2 + 1];
            }
            p.out[((size_t)i * 6 + rc) * kN + z] = a.reduced();
        }
    }
}
void launch_fold_mac_two(const uint64_t* q, const uint64_t* d, uint64_t* out, uint32_t m2, uint32_t ell, uint32_t bits, uint32_t np, hipStream_t s, const Lanes& lanes) {
    if (np == 0) return;
    FoldMacParams p{};
    static_cast<FoldMacParamsCore&>(p) = FoldMacParamsCore{q, d, out, m2, m2, nullptr};
    p.lanes = lanes;
    if (lanes.n > 1)
        hipLaunchKernelGGL(fold_mac_two_kernel<Lanes>, dim3(kN / 64, np, lanes.n), dim3(kTpb), 0, s, p, ell, bits);
    else
        hipLaunchKernelGGL(fold_mac_two_kernel<NoLanes>, dim3(kN / 64, np, 1), dim3(kTpb), 0, s, no_lanes(p), ell, bits);
}
void launch_fold_mac(const uint64_t* key, const uint64_t* d, uint64_t* out, uint32_t K, uint32_t np, hipStream_t s, uint32_t key_stride, const uint64_t* addend,
                     const Lanes& lanes) {
    if (np == 0) return;
    FoldMacParams p{};
    static_cast<FoldMacParamsCore&>(p) = FoldMacParamsCore{key, d, out, K, key_stride ? key_stride : K, addend};
    p.lanes = lanes;
    const bool two = np * lanes.n >= 16 && np % 2 == 0;  // wide rounds (all query lanes together): 2 ciphertexts per workgroup
    const dim3 grid(kN / 64, two ? np / 2 : np, lanes.n);
    if (lanes.n > 1) {  // (one query: the instantiations without lane arguments, kernels.h NoLanes)
        if (two)
            hipLaunchKernelGGL((fold_mac_kernel<2, Lanes>), grid, dim3(kTpb), 0, s, p);
        else
            hipLaunchKernelGGL((fold_mac_kernel<1, Lanes>), grid, dim3(kTpb), 0, s, p);
    } else {
        if (two)
            hipLaunchKernelGGL((fold_mac_kernel<2, NoLanes>), grid, dim3(kTpb), 0, s, no_lanes(p));
        else
            hipLaunchKernelGGL((fold_mac_kernel<1, NoLanes>), grid, dim3(kTpb), 0, s, no_lanes(p));
    }
}

// ---- add / mul_by_const (src/poly.cpp:138-155, 190-211) ----------------------------------------------
__global__ __launch_bounds__(kTpb) void add_kernel(const uint64_t* a, const uint64_t* b, uint64_t* out) {
    const size_t i = (size_t)blockIdx.x * kTpb + threadIdx.x;
    uint64_t x = a[i], y = b[i];
    out[i] = pack(mod_p((uint64_t)lo32(x) + lo32(y)), mod_b((uint64_t)hi32(x) + hi32(y)));
}
void launch_add(const uint64_t* a, const uint64_t* b, uint64_t* out, uint32_t npolys, hipStream_t s) {
    if (npolys) hipLaunchKernelGGL(add_kernel, dim3(npolys * kBpp), dim3(kTpb), 0, s, a, b, out);
}
__global__ __launch_bounds__(kTpb) void mul_by_const_kernel(const uint64_t* single, const uint64_t* a, uint64_t* out) {
    const uint32_t z = blockIdx.x * kTpb + threadIdx.x;
    const size_t i = (size_t)blockIdx.y * kN + z;
    uint64_t x = a[i], w = single[z];
    out[i] = pack(mod_p((uint64_t)lo32(x) * lo32(w)), mod_b((uint64_t)hi32(x) * hi32(w)));
}
void launch_mul_by_const(const uint64_t* single, const uint64_t* a, uint64_t* out, uint32_t npolys, hipStream_t s) {
    if (npolys) hipLaunchKernelGGL(mul_by_const_kernel, dim3(kBpp, npolys), dim3(kTpb), 0, s, single, a, out);
}

// ---- raw-domain helpers (src/poly.cpp:240-283, src/util.cpp:114-144) ------------------------------------
__global__ __launch_bounds__(kTpb) void automorph_kernel(const uint64_t* in, uint64_t* out, uint32_t t) {
    const uint32_t i = blockIdx.x * kTpb + threadIdx.x;
    const size_t base = (size_t)blockIdx.y * kN;
    const uint64_t prod = (uint64_t)i * t;
    const uint32_t pos = (uint32_t)(prod & (kN - 1));
    const uint64_t v = in[base + i];
    out[base + pos] = ((prod >> kLogN) & 1ull) ? kQ - v : v;  // Q - a: 0 -> Q
}
void launch_automorph(const uint64_t* in, uint64_t* out, uint32_t npolys, uint32_t t, hipStream_t s) {
    if (npolys) hipLaunchKernelGGL(automorph_kernel, dim3(kBpp, npolys), dim3(kTpb), 0, s, in, out, t);
}
__global__ __launch_bounds__(kTpb) void invert_kernel(const uint64_t* in, uint64_t* out) {
    const size_t i = (size_t)blockIdx.x * kTpb + threadIdx.x;
    out[i] = kQ - in[i];
}
void launch_invert(const uint64_t* in, uint64_t* out, uint32_t npolys, hipStream_t s) {
    if (npolys) hipLaunchKernelGGL(invert_kernel, dim3(npolys * kBpp), dim3(kTpb), 0, s, in, out);
}
// in raw [rdim][cols][N] -> out raw [mx][cols][N], row = j + k*rdim
__global__ __launch_bounds__(kTpb) void gadget_invert_kernel(const uint64_t* in, uint64_t* out, uint32_t mx, uint32_t rdim, uint32_t cols) {
    const uint32_t z = blockIdx.x * kTpb + threadIdx.x;
    const uint32_t jc = blockIdx.y, j = jc / cols, c = jc - j * cols;
    const uint32_t ne = mx / rdim, bits = get_bits_per(ne);
    const uint64_t mask = (1ull << bits) - 1;
    const uint64_t v = in[((size_t)j * cols + c) * kN + z];
    for (uint32_t k = 0; k < ne; k++) {
        uint32_t sh = k * bits;
        out[((size_t)(j + k * rdim) * cols + c) * kN + z] = sh >= 64 ? 0ull : ((v >> sh) & mask);
    }
}
void launch_gadget_invert(const uint64_t* in, uint64_t* out, uint32_t mx, uint32_t rdim, uint32_t cols, hipStream_t s) {
    hipLaunchKernelGGL(gadget_invert_kernel, dim3(kBpp, rdim * cols), dim3(kTpb), 0, s, in, out, mx, rdim, cols);
}

// ---- response modulus switch (src/poly.cpp:578-601): rescale_dev lives in common.h ----------------------------
__global__ __launch_bounds__(kTpb) void rescale_kernel(const uint64_t* in, uint64_t* out, uint32_t n, uint64_t inp_mod, uint64_t out_mod) {
    const uint32_t i = blockIdx.x * kTpb + threadIdx.x;
    if (i < n) out[i] = rescale_dev(in[i] % kQ, inp_mod, out_mod);
}
// the response switch in one launch: elements [0, n0) -> out_mod0 (row 0 -> q'), [n0, n) -> out_mod1 (the rest -> 4p)
template <class L>
__global__ __launch_bounds__(kTpb) void rescale2_kernel(const uint64_t* in, uint64_t* out, uint32_t n0, uint32_t n, uint64_t inp_mod, uint64_t out_mod0,
                                                        uint64_t out_mod1, L lanes) {
    const uint32_t i = blockIdx.x * kTpb + threadIdx.x;
    lane_shift(in, lanes.here());
    lane_shift(out, lanes.here());
    if (i < n) out[i] = rescale_dev(in[i] % kQ, inp_mod, i < n0 ? out_mod0 : out_mod1);
}
// wire form of a switched response (the bit stream write_arbitrary_bits builds, src/core.cpp:32-52, along modswitch's walk,
// src/spiral.cpp:40-76): values [0, n0) at w0 bits each, then values [n0, n0 + n1) at w1 bits; one thread per 64-bit output
// word gathers the fields that overlap it (both segments are whole words: N = 2048 values per polynomial)
__global__ __launch_bounds__(kTpb) void response_wire_kernel(const uint64_t* in, uint64_t* out, uint32_t n0, uint32_t w0, uint32_t n1, uint32_t w1) {
    const uint32_t k = blockIdx.x * kTpb + threadIdx.x;
    const uint32_t words0 = n0 / 64u * w0, words1 = n1 / 64u * w1;
    if (k >= words0 + words1) return;
    const bool second = k >= words0;
    const uint32_t w = second ? w1 : w0;
    const uint64_t bit0 = (uint64_t)(second ? k - words0 : k) * 64u;  // first bit of this word inside its segment
    const uint64_t* v = in + (second ? n0 : 0);
    const uint32_t nv = second ? n1 : n0;
    const uint64_t mask = (1ull << w) - 1;
    uint64_t word = 0;
    for (uint32_t i = (uint32_t)(bit0 / w); i < nv && (uint64_t)i * w < bit0 + 64; i++) {
        const uint64_t x = v[i] & mask;
        const int64_t sh = (int64_t)((uint64_t)i * w) - (int64_t)bit0;  // position of the field's bit 0 relative to this word
        word |= sh >= 0 ? x << sh : x >> (-sh);
    }
    out[k] = word;
}
void launch_response_wire(const uint64_t* in, uint64_t* out, uint32_t n0, uint32_t w0, uint32_t n1, uint32_t w1, hipStream_t s) {
    const uint32_t words = n0 / 64u * w0 + n1 / 64u * w1;
    hipLaunchKernelGGL(response_wire_kernel, dim3((words + kTpb - 1) / kTpb), dim3(kTpb), 0, s, in, out, n0, w0, n1, w1);
}
void launch_rescale2(const uint64_t* in, uint64_t* out, uint32_t n0, uint32_t n, uint64_t inp_mod, uint64_t out_mod0, uint64_t out_mod1, hipStream_t s,
                     const Lanes& lanes) {
    if (n == 0) return;
    if (lanes.n > 1)
        hipLaunchKernelGGL(rescale2_kernel<Lanes>, dim3((n + kTpb - 1) / kTpb, 1, lanes.n), dim3(kTpb), 0, s, in, out, n0, n, inp_mod, out_mod0, out_mod1, lanes);
    else
        hipLaunchKernelGGL(rescale2_kernel<NoLanes>, dim3((n + kTpb - 1) / kTpb, 1, 1), dim3(kTpb), 0, s, in, out, n0, n, inp_mod, out_mod0, out_mod1, NoLanes{});
}
void launch_rescale(const uint64_t* in, uint64_t* out, uint32_t n, uint64_t inp_mod, uint64_t out_mod, hipStream_t s) {
    if (n) hipLaunchKernelGGL(rescale_kernel, dim3((n + kTpb - 1) / kTpb), dim3(kTpb), 0, s, in, out, n, inp_mod, out_mod);
}

// ---- coefficient expansion (src/spiral.cpp:1664-1743) -------------------------------------------------------
template <class P>
__device__ __forceinline__ void expand_mac_lane(P& p) {  // query lane blockIdx.z: every query has its own ciphertexts, keys and scratch
    const int64_t lane = p.lanes.here();
    lane_shift(p.cv, lane);
    lane_shift(p.w_e, lane);
    lane_shift(p.w_o, lane);
    lane_shift(p.g, lane);
    lane_shift(p.a1, lane);
}
// whole-round MAC: 64 slots x 4 k-groups per workgroup, partial sums combined through LDS
template <class L>
__global__ __launch_bounds__(kTpb) void expand_mac_round_kernel(ExpandMacParamsT<L> p) {
    __shared__ uint64_t sh[3][64][4];
    expand_mac_lane(p);
    const uint32_t zz = threadIdx.x & 63u, kg = threadIdx.x >> 6, z = blockIdx.x * 64u + zz, a = blockIdx.y;
    const bool odd = a >= p.cnt_e;
    const uint32_t tdim = odd ? p.t_o : p.t_e;
    const uint32_t i = p.act.index(a, p.cnt_e);
    const size_t gbase = odd ? (size_t)p.cnt_e * p.t_e + (size_t)(a - p.cnt_e) * p.t_o : (size_t)a * p.t_e;
    const uint64_t* w = (odd ? p.w_o : p.w_e) + z;
    const uint64_t* gp = p.g + gbase * kN + z;
    // next round's neg1 words are cold: request them before the MAC loop
    const bool make_next = kg == 0 && p.neg1n != nullptr && (!odd || (i >> 1) + (p.next_num_in >> 1) < p.next_cnt_o);
    uint64_t nw = 0, nws = 0;
    if (make_next) {
        nw = p.neg1n[z];
        nws = p.neg1ns[z];
    }
    // so are the words the finishing wave adds to the product: requested now, they arrive under the MAC loop
    uint64_t* c = p.cv + (size_t)i * 2 * kN + z;
    uint64_t old0 = 0, old1 = 0, a1v = 0;
    if (kg == 0) {
        old0 = c[0];
        old1 = c[kN];
        a1v = p.a1[((size_t)a * 2u + 1u) * kN + z];
    }
    Acc2 acc0, acc1;
#pragma unroll 14  // t = 56 is 14 terms per k-group: all of them in flight at once (this kernel serves the latency-bound rounds)
    for (uint32_t k = kg; k < tdim; k += 4) {
#ifndef MAC_PLAIN_LOADS  // streamed operand: read once, must not push the shared W / key rows out of L2 (-17 us on expand + convert)
        const uint64_t gv = __builtin_nontemporal_load(&gp[(size_t)k * kN]);
#else
        const uint64_t gv = gp[(size_t)k * kN];
#endif
        acc0.mac(w[(size_t)k * kN], gv);
        acc1.mac(w[(size_t)(tdim + k) * kN], gv);
    }
    if (kg > 0) {
        sh[kg - 1][zz][0] = acc0.lo;
        sh[kg - 1][zz][1] = acc0.hi;
        sh[kg - 1][zz][2] = acc1.lo;
        sh[kg - 1][zz][3] = acc1.hi;
    }
    __syncthreads();
    if (kg == 0) {
#pragma unroll
        for (int q = 0; q < 3; q++) {  // <= 56 terms of < 2^56 in total: no overflow
            acc0.lo += sh[q][zz][0];
            acc0.hi += sh[q][zz][1];
            acc1.lo += sh[q][zz][2];
            acc1.hi += sh[q][zz][3];
        }
        const uint64_t c0 = add_pk(old0, acc0.reduced());
        const uint64_t c1 = add_pk(add_pk(old1, acc1.reduced()), a1v);
        c[0] = c0;
        c[kN] = c1;
        if (make_next) {
            uint64_t* n = p.cv + (size_t)(i + p.next_num_in) * 2 * kN + z;
            n[0] = pack(csub(shoup32(lo32(c0), lo32(nw), lo32(nws), kP), kP), csub(shoup32(hi32(c0), hi32(nw), hi32(nws), kB), kB));
            n[kN] = pack(csub(shoup32(lo32(c1), lo32(nw), lo32(nws), kP), kP), csub(shoup32(hi32(c1), hi32(nw), hi32(nws), kB), kB));
        }
    }
}
// the same with two adjacent slots per thread (16-byte loads, 1 KiB per wave instruction) for the wide rounds
typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
template <class L>
__global__ __launch_bounds__(kTpb) void expand_mac_round_wide_kernel(ExpandMacParamsT<L> p) {
    __shared__ uint64_t sh[3][64][8];
    expand_mac_lane(p);
    const uint32_t zz = threadIdx.x & 63u, kg = threadIdx.x >> 6, z = (blockIdx.x * 64u + zz) * 2u, a = blockIdx.y;
    const bool odd = a >= p.cnt_e;
    const uint32_t tdim = odd ? p.t_o : p.t_e;
    const uint32_t i = p.act.index(a, p.cnt_e);
    const size_t gbase = odd ? (size_t)p.cnt_e * p.t_e + (size_t)(a - p.cnt_e) * p.t_o : (size_t)a * p.t_e;
    const uint64_t* w = (odd ? p.w_o : p.w_e) + z;
    const uint64_t* gp = p.g + gbase * kN + z;
    const bool make_next = kg == 0 && p.neg1n != nullptr && (!odd || (i >> 1) + (p.next_num_in >> 1) < p.next_cnt_o);
    u64x2_t nw = {0, 0}, nws = {0, 0};
    if (make_next) {
        nw = *reinterpret_cast<const u64x2_t*>(p.neg1n + z);
        nws = *reinterpret_cast<const u64x2_t*>(p.neg1ns + z);
    }
    uint64_t* c = p.cv + (size_t)i * 2 * kN + z;
    u64x2_t old0 = {0, 0}, old1 = {0, 0}, a1v = {0, 0};
    if (kg == 0) {  // the finishing wave's addends, in flight under the MAC loop
        old0 = *reinterpret_cast<const u64x2_t*>(c);
        old1 = *reinterpret_cast<const u64x2_t*>(c + kN);
        a1v = *reinterpret_cast<const u64x2_t*>(p.a1 + ((size_t)a * 2u + 1u) * kN + z);
    }
    Acc2 acc0[2], acc1[2];
#pragma unroll 7
    for (uint32_t k = kg; k < tdim; k += 4) {
        const u64x2_t gv = __builtin_nontemporal_load(reinterpret_cast<const u64x2_t*>(gp + (size_t)k * kN));
        const u64x2_t w0 = *reinterpret_cast<const u64x2_t*>(w + (size_t)k * kN), w1 = *reinterpret_cast<const u64x2_t*>(w + (size_t)(tdim + k) * kN);
        acc0[0].mac(w0.x, gv.x);
        acc0[1].mac(w0.y, gv.y);
        acc1[0].mac(w1.x, gv.x);
        acc1[1].mac(w1.y, gv.y);
    }
    if (kg > 0) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            sh[kg - 1][zz][4 * h + 0] = acc0[h].lo;
            sh[kg - 1][zz][4 * h + 1] = acc0[h].hi;
            sh[kg - 1][zz][4 * h + 2] = acc1[h].lo;
            sh[kg - 1][zz][4 * h + 3] = acc1[h].hi;
        }
    }
    __syncthreads();
    if (kg == 0) {
        uint64_t c0[2], c1[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
#pragma unroll
            for (int q = 0; q < 3; q++) {  // <= 56 terms of < 2^56 in total: no overflow
                acc0[h].lo += sh[q][zz][4 * h + 0];
                acc0[h].hi += sh[q][zz][4 * h + 1];
                acc1[h].lo += sh[q][zz][4 * h + 2];
                acc1[h].hi += sh[q][zz][4 * h + 3];
            }
            c0[h] = add_pk(h ? old0.y : old0.x, acc0[h].reduced());
            c1[h] = add_pk(add_pk(h ? old1.y : old1.x, acc1[h].reduced()), h ? a1v.y : a1v.x);
        }
        *reinterpret_cast<u64x2_t*>(c) = u64x2_t{c0[0], c0[1]};
        *reinterpret_cast<u64x2_t*>(c + kN) = u64x2_t{c1[0], c1[1]};
        if (make_next) {
            uint64_t* n = p.cv + (size_t)(i + p.next_num_in) * 2 * kN + z;
            uint64_t n0[2], n1[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const uint64_t ww = h ? nw.y : nw.x, ws = h ? nws.y : nws.x;
                n0[h] = pack(csub(shoup32(lo32(c0[h]), lo32(ww), lo32(ws), kP), kP), csub(shoup32(hi32(c0[h]), hi32(ww), hi32(ws), kB), kB));
                n1[h] = pack(csub(shoup32(lo32(c1[h]), lo32(ww), lo32(ws), kP), kP), csub(shoup32(hi32(c1[h]), hi32(ww), hi32(ws), kB), kB));
            }
            *reinterpret_cast<u64x2_t*>(n) = u64x2_t{n0[0], n0[1]};
            *reinterpret_cast<u64x2_t*>(n + kN) = u64x2_t{n1[0], n1[1]};
        }
    }
}
// The widest rounds: CT ciphertexts of one parity per workgroup, so that a W word is fetched once for CT digit streams (the
// single-ciphertext kernels above read 2 W words per digit word: two thirds of their L2 traffic).  Same split over 4
// k-groups; wave c finishes ciphertext c from the other waves' partial sums (LDS [k-group][ct][word][lane]: conflict-free).
template <int CT, class L>
__global__ __launch_bounds__(kTpb) void expand_mac_round_batch_kernel(ExpandMacParamsT<L> p) {
    __shared__ uint64_t sh[4][CT][8][64];
    expand_mac_lane(p);
    const uint32_t zz = threadIdx.x & 63u, kg = threadIdx.x >> 6, z = (blockIdx.x * 64u + zz) * 2u;
    const uint32_t groups_e = (p.cnt_e + CT - 1) / CT;
    const bool odd = blockIdx.y >= groups_e;
    const uint32_t a0 = odd ? p.cnt_e + (blockIdx.y - groups_e) * CT : blockIdx.y * CT;
    const uint32_t n = min((uint32_t)CT, (odd ? p.cnt_e + p.cnt_o : p.cnt_e) - a0);
    const uint32_t tdim = odd ? p.t_o : p.t_e;
    const size_t gbase = odd ? (size_t)p.cnt_e * p.t_e + (size_t)(a0 - p.cnt_e) * p.t_o : (size_t)a0 * p.t_e;
    const uint64_t* w = (odd ? p.w_o : p.w_e) + z;
    const uint64_t* gp = p.g + gbase * kN + z;
    // wave kg finishes ciphertext kg of the group (CT <= 4): what it adds to the product -- the old ciphertext, NTT(automorph(c_1)),
    // the next round's neg1 words -- is requested now and arrives under the MAC loop
    static_assert(CT <= 4, "one ciphertext per finishing wave");
    const bool fin = kg < (uint32_t)CT && kg < n;
    const uint32_t fa = a0 + kg, fi = p.act.index(fin ? fa : a0, p.cnt_e);
    const bool make_next = fin && p.neg1n != nullptr && (!odd || (fi >> 1) + (p.next_num_in >> 1) < p.next_cnt_o);
    uint64_t* cp = p.cv + (size_t)fi * 2 * kN + z;
    u64x2_t old0 = {0, 0}, old1 = {0, 0}, a1v = {0, 0}, nw = {0, 0}, nws = {0, 0};
    if (fin) {
        old0 = *reinterpret_cast<const u64x2_t*>(cp);
        old1 = *reinterpret_cast<const u64x2_t*>(cp + kN);
        a1v = *reinterpret_cast<const u64x2_t*>(p.a1 + ((size_t)fa * 2u + 1u) * kN + z);
    }
    if (make_next) {
        nw = *reinterpret_cast<const u64x2_t*>(p.neg1n + z);
        nws = *reinterpret_cast<const u64x2_t*>(p.neg1ns + z);
    }
    Acc2 acc[CT][2][2];  // [ciphertext][output row][slot of the pair]
#pragma unroll 2
    for (uint32_t k = kg; k < tdim; k += 4) {
        const u64x2_t w0 = *reinterpret_cast<const u64x2_t*>(w + (size_t)k * kN), w1 = *reinterpret_cast<const u64x2_t*>(w + (size_t)(tdim + k) * kN);
#pragma unroll
        for (int c = 0; c < CT; c++) {
            const uint32_t cc = min((uint32_t)c, n - 1u);  // a short last group re-reads its last ciphertext; the result is not stored
            const u64x2_t gv = __builtin_nontemporal_load(reinterpret_cast<const u64x2_t*>(gp + ((size_t)cc * tdim + k) * kN));
            acc[c][0][0].mac(w0.x, gv.x);
            acc[c][0][1].mac(w0.y, gv.y);
            acc[c][1][0].mac(w1.x, gv.x);
            acc[c][1][1].mac(w1.y, gv.y);
        }
    }
#pragma unroll
    for (int c = 0; c < CT; c++)
        if ((uint32_t)(c & 3) != kg) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    sh[kg][c][4 * r + 2 * h + 0][zz] = acc[c][r][h].lo;
                    sh[kg][c][4 * r + 2 * h + 1][zz] = acc[c][r][h].hi;
                }
        }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CT; c++) {
        if ((uint32_t)c != kg || !fin) continue;
        const uint32_t i = fi;
#pragma unroll
        for (int q = 0; q < 4; q++)  // <= 56 terms of < 2^56 in total: no overflow
            if ((uint32_t)q != kg) {
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        acc[c][r][h].lo += sh[q][c][4 * r + 2 * h + 0][zz];
                        acc[c][r][h].hi += sh[q][c][4 * r + 2 * h + 1][zz];
                    }
            }
        uint64_t c0[2], c1[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            c0[h] = add_pk(h ? old0.y : old0.x, acc[c][0][h].reduced());
            c1[h] = add_pk(add_pk(h ? old1.y : old1.x, acc[c][1][h].reduced()), h ? a1v.y : a1v.x);
        }
        *reinterpret_cast<u64x2_t*>(cp) = u64x2_t{c0[0], c0[1]};
        *reinterpret_cast<u64x2_t*>(cp + kN) = u64x2_t{c1[0], c1[1]};
        if (make_next) {
            uint64_t* nx = p.cv + (size_t)(i + p.next_num_in) * 2 * kN + z;
            uint64_t n0[2], n1[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const uint64_t ww = h ? nw.y : nw.x, ws = h ? nws.y : nws.x;
                n0[h] = pack(csub(shoup32(lo32(c0[h]), lo32(ww), lo32(ws), kP), kP), csub(shoup32(hi32(c0[h]), hi32(ww), hi32(ws), kB), kB));
                n1[h] = pack(csub(shoup32(lo32(c1[h]), lo32(ww), lo32(ws), kP), kP), csub(shoup32(hi32(c1[h]), hi32(ww), hi32(ws), kB), kB));
            }
            *reinterpret_cast<u64x2_t*>(nx) = u64x2_t{n0[0], n0[1]};
            *reinterpret_cast<u64x2_t*>(nx + kN) = u64x2_t{n1[0], n1[1]};
        }
    }
}
void launch_expand_mac_round(const ExpandMacParams& p, hipStream_t s) {
    const uint32_t cnt = p.cnt_e + p.cnt_o;
    if (cnt == 0) return;
    static const uint32_t wide_min = [] {
        const char* e = tuning_env("SPIRAL_MAC_WIDE_MIN");  // tuning only
        return e ? (uint32_t)strtoul(e, nullptr, 10) : 64u;
    }();
    static const uint32_t ct2_min = [] {
        const char* e = tuning_env("SPIRAL_MAC_CT2_MIN");  // tuning only
        return e ? (uint32_t)strtoul(e, nullptr, 10) : 64u;
    }();
    static const uint32_t ct4_min = [] {
        const char* e = tuning_env("SPIRAL_MAC_CT4_MIN");  // tuning only
        return e ? (uint32_t)strtoul(e, nullptr, 10) : 128u;
    }();
    // (the thresholds count the ciphertexts of all query lanes: what matters is how many workgroups the launch has; the groups of a batch kernel
    // still come from one lane -- they share that lane's W)
    const uint32_t eff = cnt * p.lanes.n;
// (one query: the instantiations without lane arguments, kernels.h NoLanes)
    const bool one = p.lanes.n == 1;
    const auto q = no_lanes(p);
    const dim3 b(kTpb);
    if (eff >= ct4_min && cnt >= 8) {
        const dim3 g(kN / 128, (p.cnt_e + 3) / 4 + (p.cnt_o + 3) / 4, p.lanes.n);
        if (one) hipLaunchKernelGGL((expand_mac_round_batch_kernel<4, NoLanes>), g, b, 0, s, q);
        else hipLaunchKernelGGL((expand_mac_round_batch_kernel<4, Lanes>), g, b, 0, s, p);
    } else if (eff >= ct2_min && cnt >= 4) {
        const dim3 g(kN / 128, (p.cnt_e + 1) / 2 + (p.cnt_o + 1) / 2, p.lanes.n);
        if (one) hipLaunchKernelGGL((expand_mac_round_batch_kernel<2, NoLanes>), g, b, 0, s, q);
        else hipLaunchKernelGGL((expand_mac_round_batch_kernel<2, Lanes>), g, b, 0, s, p);
    } else if (eff >= wide_min) {
        const dim3 g(kN / 128, cnt, p.lanes.n);
        if (one) hipLaunchKernelGGL(expand_mac_round_wide_kernel<NoLanes>, g, b, 0, s, q);
        else hipLaunchKernelGGL(expand_mac_round_wide_kernel<Lanes>, g, b, 0, s, p);
    } else {
        const dim3 g(kN / 64, cnt, p.lanes.n);
        if (one) hipLaunchKernelGGL(expand_mac_round_kernel<NoLanes>, g, b, 0, s, q);
        else hipLaunchKernelGGL(expand_mac_round_kernel<Lanes>, g, b, 0, s, p);
    }
}

// ---- scalToMat (src/spiral.cpp:1834-1885) ----------------------------------------------------------------------
// prod[r][c] = sum_k W[r][2k + c] * g[k]  (special_distribute makes column c see only W's columns 2k+c),
// out = prod + pad(cv row 1) at (1,0) and (2,1)
__device__ __forceinline__ void scal2mat_slot(const uint64_t* w, const uint64_t* g, uint32_t t_conv, uint64_t cv1, uint32_t z, uint64_t out[3][2]) {
    Acc2 acc[3][2];
    if (t_conv <= 4) {
        // few digits (the base path's t_conv = 4): every operand of the product -- the digits and their 6 t_conv W words -- is requested before the first
        // multiply.  With the W loads inside the per-digit loop a wave made one dependent round trip per digit (a wave of the conversion launch lived 11 us)
        uint64_t gv[4], wv[4][3][2];
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) {
            const uint32_t k = min(u, t_conv - 1);
            gv[u] = __builtin_nontemporal_load(&g[(size_t)k * kN]);
#pragma unroll
            for (uint32_t r = 0; r < 3; r++) {
                const uint64_t* wr = w + ((size_t)r * 2 * t_conv + 2 * k) * kN + z;
                wv[u][r][0] = wr[0];
                wv[u][r][1] = wr[kN];
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < 4; u++)
            if (u < t_conv) {
#pragma unroll
                for (uint32_t r = 0; r < 3; r++) {
                    acc[r][0].mac(wv[u][r][0], gv[u]);
                    acc[r][1].mac(wv[u][r][1], gv[u]);
                }
            }
#pragma unroll
        for (uint32_t r = 0; r < 3; r++)
#pragma unroll
            for (uint32_t c = 0; c < 2; c++) {
                uint64_t v = acc[r][c].reduced();
                if ((r == 1 && c == 0) || (r == 2 && c == 1)) v = add_pk(v, cv1);
                out[r][c] = v;
            }
        return;
    }
    // the digits stream from HBM once (t_conv polynomials per ciphertext, 1.9 GB at the SpiralStream sets' t_conv = 56): eight of them
    // are requested before the first is used -- with one 8-byte load in flight per thread the product ran at 1.8 TB/s
    constexpr uint32_t kAhead = 8;
    for (uint32_t k0 = 0; k0 < t_conv; k0 += kAhead) {
        uint64_t gv[kAhead];
#pragma unroll
        for (uint32_t u = 0; u < kAhead; u++) {
            const uint32_t k = min(k0 + u, t_conv - 1);
#ifndef S2M_PLAIN_LOADS  // digits are read once
            gv[u] = __builtin_nontemporal_load(&g[(size_t)k * kN]);
#else
            gv[u] = g[(size_t)k * kN];
#endif
        }
#pragma unroll
        for (uint32_t u = 0; u < kAhead; u++) {
            const uint32_t k = k0 + u;
            if (k < t_conv) {
#pragma unroll
                for (uint32_t r = 0; r < 3; r++) {
                    const uint64_t* wr = w + ((size_t)r * 2 * t_conv + 2 * k) * kN + z;
                    acc[r][0].mac(wr[0], gv[u]);
                    acc[r][1].mac(wr[kN], gv[u]);
                }
            }
        }
    }
#pragma unroll
    for (uint32_t r = 0; r < 3; r++)
#pragma unroll
        for (uint32_t c = 0; c < 2; c++) {
            uint64_t v = acc[r][c].reduced();
            if ((r == 1 && c == 0) || (r == 2 && c == 1)) v = add_pk(v, cv1);
            out[r][c] = v;
        }
}
template <class P>
__device__ __forceinline__ void scal2mat_lane(P& p) {  // query lane blockIdx.z
    const int64_t lane = p.lanes.here();
    lane_shift(p.w, lane);
    lane_shift(p.g, lane);
    lane_shift(p.cv, lane);
    lane_shift(p.out, lane);
    lane_shift(p.qs, lane);  // (u32 records in a u64-word arena: the shift is in bytes)
}
template <class P>
__device__ __forceinline__ void gsw_lane(P& p) {
    const int64_t lane = p.lanes.here();
    lane_shift(p.w, lane);
    lane_shift(p.v, lane);
    lane_shift(p.chat, lane);
    lane_shift(p.cv, lane);
    lane_shift(p.gsw, lane);
    lane_shift(p.key, lane);
}
template <class L>
__global__ __launch_bounds__(kTpb) void scal2mat_kernel(Scal2MatParamsT<L> p) {
    scal2mat_lane(p);
    const uint32_t z = blockIdx.x * kTpb + threadIdx.x, a = blockIdx.y;
    const uint64_t cv1 = p.cv[((size_t)p.cv_pos(a) * 2 + 1) * kN + z];
    uint64_t out[3][2];
    scal2mat_slot(p.w, p.g + (size_t)a * p.t_conv * kN + z, p.t_conv, cv1, z, out);
    if (p.out) {
#pragma unroll
        for (uint32_t r = 0; r < 3; r++)
#pragma unroll
            for (uint32_t c = 0; c < 2; c++) p.out[(((size_t)a * 3 + r) * 2 + c) * kN + z] = out[r][c];
    }
    if (p.qs) {  // sweep query record of (z, j): {p rows 0..2 | b rows 0..2} for m = 0, then m = 1
        uint4* rec = reinterpret_cast<uint4*>(p.qs + ((size_t)z * (p.jm_total / 2) + p.j_base + a) * 12);
        rec[0] = make_uint4(lo32(out[0][0]), lo32(out[1][0]), lo32(out[2][0]), hi32(out[0][0]));
        rec[1] = make_uint4(hi32(out[1][0]), hi32(out[2][0]), lo32(out[0][1]), lo32(out[1][1]));
        rec[2] = make_uint4(lo32(out[2][1]), hi32(out[0][1]), hi32(out[1][1]), hi32(out[2][1]));
    }
}
// the same product for 16 ciphertexts x 16 slots per workgroup, records only: the 48-byte sweep records of one slot and
// consecutive j are adjacent in memory, so the workgroup transposes its results through LDS and writes 768-byte runs
// (one thread per slot and ciphertext writes 16-byte pieces 12 KiB apart instead)
__device__ __forceinline__ void scal2mat_rec_body(const Scal2MatParamsCore& p, uint32_t bx, uint32_t by) {
    __shared__ uint4 sh[16][16][3];  // [slot][ct][piece]
    const uint32_t zl = threadIdx.x & 15u, al = threadIdx.x >> 4, z0 = bx * 16u, a0 = by * 16u;
    {
        const uint32_t z = z0 + zl, a = a0 + al;
        const uint64_t cv1 = p.cv[((size_t)p.cv_pos(a) * 2 + 1) * kN + z];
        uint64_t out[3][2];
        scal2mat_slot(p.w, p.g + (size_t)a * p.t_conv * kN + z, p.t_conv, cv1, z, out);
        sh[zl][al][0] = make_uint4(lo32(out[0][0]), lo32(out[1][0]), lo32(out[2][0]), hi32(out[0][0]));
        sh[zl][al][1] = make_uint4(hi32(out[1][0]), hi32(out[2][0]), lo32(out[0][1]), lo32(out[1][1]));
        sh[zl][al][2] = make_uint4(lo32(out[2][1]), hi32(out[0][1]), hi32(out[1][1]), hi32(out[2][1]));
    }
    __syncthreads();
    const uint4* flat = &sh[0][0][0];
#pragma unroll
    for (uint32_t m = 0; m < 3; m++) {
        const uint32_t q = threadIdx.x + 256u * m, zz = q / 48u, within = q - zz * 48u;  // 48 pieces per slot
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        u32x4* rec = reinterpret_cast<u32x4*>(p.qs + ((size_t)(z0 + zz) * (p.jm_total / 2) + p.j_base + a0) * 12);
        const uint4 v = flat[q];
#ifdef S2M_PLAIN_STORE
        rec[within] = u32x4{v.x, v.y, v.z, v.w};
#else
        // streaming store: the 25 MB of records should not sit dirty in the caches when the sweep starts -- dirty lines
        // written back into a saturated read stream cost the sweep ~20 us (tools/sweep_in_situ.py)
        __builtin_nontemporal_store(u32x4{v.x, v.y, v.z, v.w}, rec + within);
#endif
    }
}
// Large t_conv (the SpiralStream sets: 56 digit polynomials per ciphertext, 1.9 GB of digits): 64 slots x 16 ciphertexts per
// workgroup, a thread takes one slot of FOUR ciphertexts, so a W word is fetched once per four digit words (the 16 x 16 tile above
// fetches six per digit word and ran at 1.8 TB/s of digits, bound by the L1/L2 request rate) and a wave reads 512 contiguous bytes.
__device__ __forceinline__ void scal2mat_rec4_body(const Scal2MatParamsCore& p, uint32_t bx, uint32_t by) {
    __shared__ uint4 sh[64][16][3];  // [slot][ct][piece]
    const uint32_t zl = threadIdx.x & 63u, cg = threadIdx.x >> 6, z = bx * 64u + zl, a0 = by * 16u, tc = p.t_conv;
    Acc2 acc[4][3][2];
    const uint64_t* g = p.g + (size_t)(a0 + cg * 4u) * tc * kN + z;
    const uint64_t* w = p.w + z;
#pragma unroll 2
    for (uint32_t k = 0; k < tc; k++) {
        uint64_t gv[4], wv[3][2];
#pragma unroll
        for (uint32_t c = 0; c < 4; c++) gv[c] = __builtin_nontemporal_load(&g[((size_t)c * tc + k) * kN]);
#pragma unroll
        for (uint32_t r = 0; r < 3; r++) {
            wv[r][0] = w[((size_t)r * 2 * tc + 2 * k) * kN];
            wv[r][1] = w[((size_t)r * 2 * tc + 2 * k + 1) * kN];
        }
#pragma unroll
        for (uint32_t c = 0; c < 4; c++)
#pragma unroll
            for (uint32_t r = 0; r < 3; r++) {
                acc[c][r][0].mac(wv[r][0], gv[c]);
                acc[c][r][1].mac(wv[r][1], gv[c]);
            }
    }
#pragma unroll
    for (uint32_t c = 0; c < 4; c++) {
        const uint32_t al = cg * 4u + c;
        const uint64_t cv1 = p.cv[((size_t)p.cv_pos(a0 + al) * 2 + 1) * kN + z];
        uint64_t out[3][2];
#pragma unroll
        for (uint32_t r = 0; r < 3; r++)
#pragma unroll
            for (uint32_t cc = 0; cc < 2; cc++) {
                uint64_t v = acc[c][r][cc].reduced();
                if ((r == 1 && cc == 0) || (r == 2 && cc == 1)) v = add_pk(v, cv1);
                out[r][cc] = v;
            }
        sh[zl][al][0] = make_uint4(lo32(out[0][0]), lo32(out[1][0]), lo32(out[2][0]), hi32(out[0][0]));
        sh[zl][al][1] = make_uint4(hi32(out[1][0]), hi32(out[2][0]), lo32(out[0][1]), lo32(out[1][1]));
        sh[zl][al][2] = make_uint4(lo32(out[2][1]), hi32(out[0][1]), hi32(out[1][1]), hi32(out[2][1]));
    }
    __syncthreads();
    const uint4* flat = &sh[0][0][0];
#pragma unroll
    for (uint32_t m = 0; m < 12; m++) {
        const uint32_t q = threadIdx.x + 256u * m, zz = q / 48u, within = q - zz * 48u;  // 48 pieces per slot
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        u32x4* rec = reinterpret_cast<u32x4*>(p.qs + ((size_t)(bx * 64u + zz) * (p.jm_total / 2) + p.j_base + a0) * 12);
        const uint4 v = flat[q];
        __builtin_nontemporal_store(u32x4{v.x, v.y, v.z, v.w}, rec + within);
    }
}
static uint32_t scal2mat_wide_min() {
    static const uint32_t v = [] {
        const char* e = tuning_env("SPIRAL_S2M_WIDE_MIN");  // tuning only: smallest t_conv that takes the 64 x 16 tile
        return e ? (uint32_t)strtoul(e, nullptr, 10) : 16u;
    }();
    return v;
}
template <class L>
__global__ __launch_bounds__(kTpb) void scal2mat_rec4_kernel(Scal2MatParamsT<L> p) {
    scal2mat_lane(p);
    scal2mat_rec4_body(p, blockIdx.x, blockIdx.y);
}
template <class L>
__global__ __launch_bounds__(kTpb) void scal2mat_rec_kernel(Scal2MatParamsT<L> p) {
    scal2mat_lane(p);
    scal2mat_rec_body(p, blockIdx.x, blockIdx.y);
}
static bool scal2mat_rec_ok(const Scal2MatParams& p) { return p.count && p.count % 16 == 0 && p.qs && !p.out; }
// The 64 x 16 tile (a W word fetched once per four digit words, 512-byte runs per wave) for large t_conv -- and for ANY t_conv once the launch is big
// enough to fill the chip with its fewer, larger workgroups (48 KiB of LDS, 160 VGPRs: three per CU): the conversion of an eight-query batch at
// t_conv = 4 takes 167 us with it against 202 us with the 16 x 16 tile, which wins at one query (512 workgroups of the wide tile would leave half the CUs idle)
static bool scal2mat_wide(const Scal2MatParams& p) { return p.t_conv >= scal2mat_wide_min() || (kN / 64u) * (p.count / 16u) * p.lanes.n >= 2048u; }
void launch_scal2mat(const Scal2MatParams& p, hipStream_t s) {
    if (scal2mat_rec_ok(p)) {
        if (scal2mat_wide(p)) {
            if (p.lanes.n > 1) hipLaunchKernelGGL(scal2mat_rec4_kernel<Lanes>, dim3(kN / 64, p.count / 16, p.lanes.n), dim3(kTpb), 0, s, p);
            else hipLaunchKernelGGL(scal2mat_rec4_kernel<NoLanes>, dim3(kN / 64, p.count / 16, 1), dim3(kTpb), 0, s, no_lanes(p));
        } else {
            if (p.lanes.n > 1) hipLaunchKernelGGL(scal2mat_rec_kernel<Lanes>, dim3(kN / 16, p.count / 16, p.lanes.n), dim3(kTpb), 0, s, p);
            else hipLaunchKernelGGL(scal2mat_rec_kernel<NoLanes>, dim3(kN / 16, p.count / 16, 1), dim3(kTpb), 0, s, no_lanes(p));
        }
        return;
    }
    if (p.count == 0) return;
    if (p.lanes.n > 1) hipLaunchKernelGGL(scal2mat_kernel<Lanes>, dim3(kBpp, p.count, p.lanes.n), dim3(kTpb), 0, s, p);
    else hipLaunchKernelGGL(scal2mat_kernel<NoLanes>, dim3(kBpp, p.count, 1), dim3(kTpb), 0, s, no_lanes(p));
}

// ---- regevToGSW (src/spiral.cpp:1985-2025) ------------------------------------------------------------------------
// HOIST: the V product's operands requested up front (the conversion's own launch in a batch: 104 VGPRs); the merged single-query launch keeps the loop (78)
template <bool HOIST>
__device__ __forceinline__ void regev_to_gsw_body(const GswParamsCore& p, uint32_t bx, uint32_t by) {
    const uint32_t z = bx * kTpb + threadIdx.x, di = by;  // di = d*ell + i
    const uint32_t d = di / p.ell, i = di - d * p.ell, tc = p.t_conv;
    const uint64_t* chat = p.chat + (size_t)di * 2 * tc * kN + z;
    const uint64_t cv1 = p.cv[((size_t)p.cv_pos(di) * 2 + 1) * kN + z];
    uint64_t s2m[3][2];
    scal2mat_slot(p.w, chat, tc, cv1, z, s2m);
    Acc2 accv[3];
    if (HOIST && tc <= 4) {
        // as scal2mat_slot: all 2 t_conv digit words and their 6 t_conv V words requested before the first multiply (inside the loop the
        // compiler waits for each term before requesting the next: eight dependent round trips per thread at t_conv = 4)
        uint64_t cvv[8], vv[8][3];
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) {
            const uint32_t k = min(u, 2 * tc - 1);
            cvv[u] = chat[(size_t)k * kN];
#pragma unroll
            for (uint32_t r = 0; r < 3; r++) vv[u][r] = p.v[((size_t)r * 2 * tc + k) * kN + z];
        }
#pragma unroll
        for (uint32_t u = 0; u < 8; u++)
            if (u < 2 * tc) {
#pragma unroll
                for (uint32_t r = 0; r < 3; r++) accv[r].mac(vv[u][r], cvv[u]);
            }
    } else {
        for (uint32_t k = 0; k < 2 * tc; k++) {
            uint64_t cvv = chat[(size_t)k * kN];
#pragma unroll
            for (uint32_t r = 0; r < 3; r++) accv[r].mac(p.v[((size_t)r * 2 * tc + k) * kN + z], cvv);
        }
    }
    const uint32_t cols = 3 * p.ell;
    uint64_t* g = p.gsw ? p.gsw + (size_t)(p.dims - 1 - d) * 3 * cols * kN + z : nullptr;  // (the resident server keeps the matrices only as the Q half of the fold key)
    // (the host-buffer fold seam's key: key[r][mm] = G2[r][mm] - gsw[r][mm] (= Q_neg, src/spiral.cpp:2361-2379; the NTT is linear and a constant c is c in
    // every slot), key[r][m2 + mm] = gsw[r][mm]; G2[r][3i + c] = 2^(bits*i) iff c == r.  The resident server passes no key: it folds with the matrices alone)
    uint64_t* key = p.key ? p.key + (size_t)(p.dims - 1 - d) * 3 * (2 * cols) * kN + z : nullptr;
    const uint32_t sh = get_bits_per(p.ell) * i;
    const uint32_t g2p = sh < 64 ? mod_p(1ull << sh) : 0u, g2b = sh < 64 ? mod_b(1ull << sh) : 0u;
#pragma unroll
    for (uint32_t r = 0; r < 3; r++) {
        const uint64_t col[3] = {accv[r].reduced(), s2m[r][0], s2m[r][1]};
#pragma unroll
        for (uint32_t c = 0; c < 3; c++) {
            const uint64_t q = col[c];
            if (g) g[((size_t)r * cols + 3 * i + c) * kN] = q;
            if (key) {
                const uint32_t gp = c == r ? g2p : 0u, gb = c == r ? g2b : 0u;
                key[((size_t)r * 2 * cols + 3 * i + c) * kN] = pack(csub(gp + kP - lo32(q), kP), csub(gb + kB - hi32(q), kB));
                key[((size_t)r * 2 * cols + cols + 3 * i + c) * kN] = q;
            }
        }
    }
}
template <class L>
__global__ __launch_bounds__(kTpb) void regev_to_gsw_kernel(GswParamsT<L> p) {
    gsw_lane(p);
    regev_to_gsw_body<true>(p, blockIdx.x, blockIdx.y);
}
// the two conversion products are independent: one launch, the first n1 blocks ScalToMat, the rest Regev->GSW
template <bool WIDE, class L>
__global__ __launch_bounds__(kTpb) void convert_products_kernel(Scal2MatParamsT<L> sp, GswParamsT<L> gp, uint32_t n1) {
    scal2mat_lane(sp);
    gsw_lane(gp);
#ifdef CONVERT_GSW_FIRST  // (variant: the latency-bound Regev->GSW workgroups dispatched first, the streaming ScalToMat ones fill in behind them)
    const uint32_t n2 = gridDim.x - n1, b = blockIdx.x < n2 ? blockIdx.x + n1 : blockIdx.x - n2;
#else
    const uint32_t b = blockIdx.x;
#endif
    if (b < n1) {
        if constexpr (WIDE)
            scal2mat_rec4_body(sp, b % (kN / 64u), b / (kN / 64u));
        else
            scal2mat_rec_body(sp, b % (kN / 16u), b / (kN / 16u));
    } else {
        const uint32_t bb = b - n1;
        regev_to_gsw_body<false>(gp, bb % kBpp, bb / kBpp);
    }
}
void launch_convert_products(const Scal2MatParams& sp, const GswParams& gp, hipStream_t s) {
    if (scal2mat_rec_ok(sp) && gp.dims) {
        const bool wide = scal2mat_wide(sp);
        const uint32_t n1 = (kN / (wide ? 64u : 16u)) * (sp.count / 16u), n2 = kBpp * gp.dims * gp.ell;
        const dim3 g(n1 + n2, 1, sp.lanes.n);
        if (sp.lanes.n > 1) {
            if (wide) hipLaunchKernelGGL((convert_products_kernel<true, Lanes>), g, dim3(kTpb), 0, s, sp, gp, n1);
            else hipLaunchKernelGGL((convert_products_kernel<false, Lanes>), g, dim3(kTpb), 0, s, sp, gp, n1);
        } else {  // one query: no lane arguments (kernels.h NoLanes)
            if (wide) hipLaunchKernelGGL((convert_products_kernel<true, NoLanes>), g, dim3(kTpb), 0, s, no_lanes(sp), no_lanes(gp), n1);
            else hipLaunchKernelGGL((convert_products_kernel<false, NoLanes>), g, dim3(kTpb), 0, s, no_lanes(sp), no_lanes(gp), n1);
        }
    } else {
        launch_scal2mat(sp, s);
        launch_regev_to_gsw(gp, s);
    }
}
void launch_regev_to_gsw(const GswParams& p, hipStream_t s) {
    if (p.dims == 0) return;
    if (p.lanes.n > 1) hipLaunchKernelGGL(regev_to_gsw_kernel<Lanes>, dim3(kBpp, p.dims * p.ell, p.lanes.n), dim3(kTpb), 0, s, p);
    else hipLaunchKernelGGL(regev_to_gsw_kernel<NoLanes>, dim3(kBpp, p.dims * p.ell, 1), dim3(kTpb), 0, s, no_lanes(p));
}

// ---- fold key from the reference's reoriented matrices (the resident path writes its keys in regev_to_gsw_kernel) --------
__global__ __launch_bounds__(kTpb) void fold_key_from_reoriented_kernel(const uint64_t* q_re, const uint64_t* qneg_re, uint64_t* key, uint32_t m2) {
    const uint32_t z = blockIdx.x * kTpb + threadIdx.x, rm = blockIdx.y, r = rm / m2, mm = rm - r * m2;
    const size_t src = (size_t)z * (3 * m2) + rm;  // z in the reference's slot order
    const uint64_t q = q_re[src], qn = qneg_re[src];
    uint64_t* k = key + (size_t)r * (2 * m2) * kN + pk_pos(z);
    k[(size_t)mm * kN] = pack(lo32(qn) % kP, hi32(qn) % kB);
    k[(size_t)(m2 + mm) * kN] = pack(lo32(q) % kP, hi32(q) % kB);
}
void launch_fold_key_from_reoriented(const uint64_t* q_re, const uint64_t* qneg_re, uint64_t* key, uint32_t m2, hipStream_t s) {
    hipLaunchKernelGGL(fold_key_from_reoriented_kernel, dim3(kBpp, 3 * m2), dim3(kTpb), 0, s, q_re, qneg_re, key, m2);
}

// ---- exchange of the GSW-bit ciphertexts of a sharded expansion (kernels.h) -------------------------------------------------
__global__ __launch_bounds__(kTpb) void gsw_bits_copy_kernel(uint64_t* cv, uint64_t* buf, uint32_t rank, uint32_t n_ranks, uint32_t n_bits, uint32_t n_max, int unpack) {
    const uint32_t z = blockIdx.x * kTpb + threadIdx.x, row = blockIdx.y & 1u, slot = blockIdx.y >> 1;  // slot = r * n_max + a (unpack) or a (pack)
    const uint32_t r = unpack ? slot / n_max : rank, a = unpack ? slot - r * n_max : slot, i = a * n_ranks + r;
    if (i >= n_bits) return;
    uint64_t* c = cv + ((size_t)(2u * i + 1u) * 2u + row) * kN + z;
    uint64_t* b = buf + ((size_t)slot * 2u + row) * kN + z;
    if (unpack)
        *c = *b;
    else
        *b = *c;
}
void launch_gsw_bits_pack(const uint64_t* cv, uint64_t* block, uint32_t rank, uint32_t n_ranks, uint32_t n_bits, hipStream_t s) {
    const uint32_t n_max = (n_bits + n_ranks - 1) / n_ranks;
    if (n_max) hipLaunchKernelGGL(gsw_bits_copy_kernel, dim3(kBpp, 2 * n_max), dim3(kTpb), 0, s, const_cast<uint64_t*>(cv), block, rank, n_ranks, n_bits, n_max, 0);
}
void launch_gsw_bits_unpack(uint64_t* cv, const uint64_t* gathered, uint32_t n_ranks, uint32_t n_bits, hipStream_t s) {
    const uint32_t n_max = (n_bits + n_ranks - 1) / n_ranks;
    if (n_max) hipLaunchKernelGGL(gsw_bits_copy_kernel, dim3(kBpp, 2 * n_max * n_ranks), dim3(kTpb), 0, s, cv, const_cast<uint64_t*>(gathered), 0u, n_ranks, n_bits, n_max, 1);
}

}  // namespace spiral
