// First-dimension sweep for BATCHES of queries on the matrix cores (the batched form of multiplyQueryByDatabase, reference
// src/spiral.cpp:628-999, which answers one query per call).
//
// Per NTT slot z and prime the sweep is a (nic x JM) by (JM x 3) integer product mod m (sweep.hip).  With B queries the right-hand side
// has 3 B columns and the 64-bit VALU MADs of sweep_kernel<0, B> become the limit (B = 4: 440 us against the 295 us the database stream
// takes).  Here both operands are split into signed 8-bit limbs and the products run as v_mfma_i32_16x16x64_i8:
//
//   database residue a in [0, m)   ->  a'' = a or a - m  in [-0x808080, 2^28 - 0x808080),   a'' + 0x808080 = w < 2^28,
//                                      limbs  s_i = byte_i(w) - 128  (i < 3, signed bytes)  and  u_3 = w >> 24  (4 bits, 0 .. 15):
//                                      a'' = s_0 + 2^8 s_1 + 2^16 s_2 + 2^24 u_3                                    -- still 28 bits
//   query residue v in [0, m)      ->  w = v + 0x808080, t_i = byte_i(w) - 128 (i < 3), t_3 = w >> 24 (0 .. 16)     -- four signed bytes
//
//   sum_k a''_k v_k = sum_{i, l} 2^{8 (i + l)} sum_k s_{i,k} t_{l,k}      exactly, every inner sum an int32 (|.| <= K 2^14, K = JM <= 2^12)
//
// so per (z, prime) the product is M = 4 limbs x nic rows, K = JM, N = 12 B columns (query, row, limb) of i8 MACs: 16 limb products per
// 28-bit product, 25.8 G MACs per query at config 2 -- 50 us of the i8 matrix rate for B = 4, 100 us for B = 8: the kernel is bound by the
// database stream again, for up to eight queries per pass.  The results are recombined, reduced mod m once per output and are bit-identical
// to the VALU sweep's.
//
// Database image for this kernel ("limb planes", built from the packed image by db_limb_planes_kernel; the same 3.5 bytes per residue):
//   [z][block of 16 columns][prime][piece of 128 terms k = 2 j + m][7 x 1 KiB]
//   1 KiB = 64 lanes x 16 bytes = one MFMA A operand (lane l: column l & 15, terms (l >> 4) * 16 + 0 .. 15 of a 64-term chunk):
//   limb 0 chunk 0, limb 0 chunk 1, limb 1 chunk 0, limb 1 chunk 1, limb 2 chunk 0, limb 2 chunk 1, nibbles (low nibble = chunk 0, high = chunk 1)
// A wave owns (z, 16 columns) and streams its 2 x K/128 x 7 KiB front to back, one 1 KiB load instruction at a time.
// The B operands (query limbs) are built in LDS from the lanes' ordinary 48-byte query records by the workgroup (all its waves share z) and read
// back with conflict-free ds_read_b128.
#include <atomic>
#include <cstdlib>
#include "common.h"
#include "kernels.h"

namespace spiral {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t kLimbBias = 0x808080u;

__device__ __forceinline__ uint32_t limb_word(uint32_t a, uint32_t m) {  // w = a'' + bias of residue a mod m
    return (a >= (1u << 28) - kLimbBias ? a - m : a) + kLimbBias;
}
template <int E>
__device__ __forceinline__ void limbs_of_group(const uint32_t (&d)[28], uint32_t (&wp)[16], uint32_t (&wb)[16]) {
    if constexpr (E < 16) {
        constexpr int JJ = E >> 1, M = E & 1;
        wp[E] = limb_word(field28<4 * JJ + 2 * M>(d), kP);
        wb[E] = limb_word(field28<4 * JJ + 2 * M + 1>(d), kB);
        limbs_of_group<E + 1>(d, wp, wb);
    }
}

// packed image -> limb planes.  One wave per (z, block of 16 columns, piece of 128 terms); lane l = (column l & 15, term block l >> 4)
// reads the two 112-byte groups (8 j each) that hold its 2 x 16 terms and writes its 16 bytes of each of the 2 x 7 pieces.
__global__ __launch_bounds__(256) void db_limb_planes_kernel(const uint64_t* __restrict__ packed, uint4* __restrict__ limbs, uint32_t nic, uint32_t dim0, uint32_t nz) {
    const uint32_t lane = threadIdx.x & 63u, wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t nk2 = dim0 >> 6, nblk16 = nic >> 4;
    const uint32_t kc2 = wave % nk2, icb = (wave / nk2) % nblk16, z = wave / (nk2 * nblk16);
    if (z >= nz) return;
    const uint32_t ic = icb * 16u + (lane & 15u), kblk = lane >> 4, groups = dim0 >> 3;
    const uint4* src = reinterpret_cast<const uint4*>(packed) + ((size_t)(z * (nic >> 6) + (ic >> 6)) * groups) * 7u * 64u + (ic & 63u);
    uint32_t lp[3][2][4] = {}, lb[3][2][4] = {}, np[4] = {}, nb[4] = {};
#pragma unroll
    for (uint32_t c = 0; c < 2; c++) {
        const uint32_t g = kc2 * 8u + c * 4u + kblk;
        uint32_t d[28];
#pragma unroll
        for (uint32_t k = 0; k < 7; k++) {
            const uint4 v = src[((size_t)g * 7u + k) * 64u];
            d[4 * k] = v.x, d[4 * k + 1] = v.y, d[4 * k + 2] = v.z, d[4 * k + 3] = v.w;
        }
        uint32_t wp[16], wb[16];
        limbs_of_group<0>(d, wp, wb);
#pragma unroll
        for (uint32_t e = 0; e < 16; e++) {
            const uint32_t xp = wp[e] ^ kLimbBias, xb = wb[e] ^ kLimbBias, sh = 8u * (e & 3u);
#pragma unroll
            for (uint32_t i = 0; i < 3; i++) {
                lp[i][c][e >> 2] |= ((xp >> (8u * i)) & 0xFFu) << sh;
                lb[i][c][e >> 2] |= ((xb >> (8u * i)) & 0xFFu) << sh;
            }
            np[e >> 2] |= (wp[e] >> 24) << (sh + 4u * c);
            nb[e >> 2] |= (wb[e] >> 24) << (sh + 4u * c);
        }
    }
    uint4* dst = limbs + ((size_t)(z * nblk16 + icb) * 2u * nk2 + kc2) * 7u * 64u + lane;  // prime p; prime b is nk2 pieces further
#pragma unroll
    for (uint32_t i = 0; i < 3; i++)
#pragma unroll
        for (uint32_t c = 0; c < 2; c++) {
            dst[(size_t)(2u * i + c) * 64u] = make_uint4(lp[i][c][0], lp[i][c][1], lp[i][c][2], lp[i][c][3]);
            dst[((size_t)nk2 * 7u + 2u * i + c) * 64u] = make_uint4(lb[i][c][0], lb[i][c][1], lb[i][c][2], lb[i][c][3]);
        }
    dst[(size_t)6u * 64u] = make_uint4(np[0], np[1], np[2], np[3]);
    dst[((size_t)nk2 * 7u + 6u) * 64u] = make_uint4(nb[0], nb[1], nb[2], nb[3]);
}

// limb planes -> packed image: the inverse map, same wave and lane assignment (the two forms are bijective on residues below the moduli, so
// packed -> limbs -> packed reproduces every byte: tests/test_gpu_parity.py::test_db_format_round_trip)
template <int T>
__device__ __forceinline__ void put_field28(uint32_t (&d)[28], uint32_t v) {
    constexpr uint32_t bit = 28u * T, w = bit >> 5, sh = bit & 31u;
    d[w] |= v << sh;
    if constexpr (sh > 4) d[w + 1] |= v >> (32u - sh);
}
template <int E>
__device__ __forceinline__ void group_of_limbs(uint32_t (&d)[28], const uint32_t (&rp)[16], const uint32_t (&rb)[16]) {
    if constexpr (E < 16) {
        constexpr int JJ = E >> 1, M = E & 1;
        put_field28<4 * JJ + 2 * M>(d, rp[E]);
        put_field28<4 * JJ + 2 * M + 1>(d, rb[E]);
        group_of_limbs<E + 1>(d, rp, rb);
    }
}
__device__ __forceinline__ uint32_t residue_of_limb_word(uint32_t w, uint32_t m) {
    const int32_t a = (int32_t)w - (int32_t)kLimbBias;
    return a < 0 ? (uint32_t)(a + (int32_t)m) : (uint32_t)a;
}
__global__ __launch_bounds__(256) void db_limb_unplanes_kernel(const uint4* __restrict__ limbs, uint64_t* __restrict__ packed, uint32_t nic, uint32_t dim0, uint32_t nz) {
    const uint32_t lane = threadIdx.x & 63u, wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t nk2 = dim0 >> 6, nblk16 = nic >> 4;
    const uint32_t kc2 = wave % nk2, icb = (wave / nk2) % nblk16, z = wave / (nk2 * nblk16);
    if (z >= nz) return;
    const uint32_t ic = icb * 16u + (lane & 15u), kblk = lane >> 4, groups = dim0 >> 3;
    const uint4* src = limbs + ((size_t)(z * nblk16 + icb) * 2u * nk2 + kc2) * 7u * 64u + lane;
    uint4* dst = reinterpret_cast<uint4*>(packed) + ((size_t)(z * (nic >> 6) + (ic >> 6)) * groups) * 7u * 64u + (ic & 63u);
    const uint4 np4 = src[(size_t)6u * 64u], nb4 = src[((size_t)nk2 * 7u + 6u) * 64u];
    const uint32_t np[4] = {np4.x, np4.y, np4.z, np4.w}, nb[4] = {nb4.x, nb4.y, nb4.z, nb4.w};
#pragma unroll
    for (uint32_t c = 0; c < 2; c++) {
        uint32_t lp[3][4], lb[3][4];
#pragma unroll
        for (uint32_t i = 0; i < 3; i++) {
            const uint4 vp = src[(size_t)(2u * i + c) * 64u], vb = src[((size_t)nk2 * 7u + 2u * i + c) * 64u];
            lp[i][0] = vp.x, lp[i][1] = vp.y, lp[i][2] = vp.z, lp[i][3] = vp.w;
            lb[i][0] = vb.x, lb[i][1] = vb.y, lb[i][2] = vb.z, lb[i][3] = vb.w;
        }
        uint32_t rp[16], rb[16];
#pragma unroll
        for (uint32_t e = 0; e < 16; e++) {
            const uint32_t sh = 8u * (e & 3u);
            uint32_t wp = 0, wb = 0;
#pragma unroll
            for (uint32_t i = 0; i < 3; i++) {
                wp |= ((lp[i][e >> 2] >> sh) & 0xFFu) << (8u * i);
                wb |= ((lb[i][e >> 2] >> sh) & 0xFFu) << (8u * i);
            }
            wp = (wp ^ kLimbBias) | (((np[e >> 2] >> (sh + 4u * c)) & 0xFu) << 24);
            wb = (wb ^ kLimbBias) | (((nb[e >> 2] >> (sh + 4u * c)) & 0xFu) << 24);
            rp[e] = residue_of_limb_word(wp, kP);
            rb[e] = residue_of_limb_word(wb, kB);
        }
        uint32_t d[28] = {};
        group_of_limbs<0>(d, rp, rb);
        const uint32_t g = kc2 * 8u + c * 4u + kblk;
#pragma unroll
        for (uint32_t k = 0; k < 7; k++) dst[((size_t)g * 7u + k) * 64u] = make_uint4(d[4 * k], d[4 * k + 1], d[4 * k + 2], d[4 * k + 3]);
    }
}

struct SweepLanes {
    const uint32_t* qs[kMaxLanes];
    uint64_t* acc[kMaxLanes];
};

template <typename T>
__device__ __forceinline__ T pick_lane(const T (&a)[kMaxLanes], uint32_t q) {  // (a dynamic index would put the argument struct in scratch)
    T p = a[0];
#pragma unroll
    for (uint32_t i = 1; i < kMaxLanes; i++) p = q == i ? a[i] : p;
    return p;
}

// x mod M for x < 2^59: quotient estimated in double (exact to +-1: x / M < 2^32 and a double carries 53 bits), remainder fixed up in 32 bits
template <uint32_t M>
__device__ __forceinline__ uint32_t mod_est(uint64_t x) {
    const double xd = (double)hi32(x) * 4294967296.0 + (double)lo32(x);
    const uint32_t q = (uint32_t)(xd * (1.0 / (double)M));
    uint32_t r = lo32(x) - q * M;  // the true remainder is in (-M, 2M): it fits 32 bits whatever the high words were
    r = (int32_t)r < 0 ? r + M : r;
    return r >= M ? r - M : r;
}
constexpr uint32_t pow256_mod(uint32_t m, int w) {
    uint64_t x = 1;
    for (int i = 0; i < w; i++) x = (x << 8) % m;
    return (uint32_t)x;
}
template <int CTRL>
__device__ __forceinline__ int64_t quad_dpp(int64_t x) {  // quad_perm exchange of a 64-bit value: CTRL 0xB1 = lane ^ 1, 0x4E = lane ^ 2
    const int lo = __builtin_amdgcn_mov_dpp((int)(uint32_t)x, CTRL, 0xF, 0xF, true), hi = __builtin_amdgcn_mov_dpp((int)(x >> 32), CTRL, 0xF, 0xF, true);
    return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}

// acc[i][t][e] = c_{i,l}: database limb i, query limb l = lane & 3, row (lane >> 4) * 4 + e of the column block, column n = 16 t + (lane & 15) =
// 12 q + 4 r + l.  The result for (row e, query q, row r) is  sum_{i,l} 2^{8 (i + l)} c_{i,l}  mod M  =  sum_l U_l mod M  with
// U_l = sum_i c_{i,l} T_{i+l},  T_w = 2^{8 w} mod M  (|U_l| < 2^56 for K <= 2^12: 64-bit MADs in lane l, no reduction); the four U_l of a quad are
// summed so that lane e ends up with the sum for row e (two quad_perm exchange steps: keep the rows of my parity, then of my half), one
// reduction mod M per lane and tile.
template <int NT, uint32_t M>
__device__ __forceinline__ void combine_limbs(const v4i (&acc)[4][NT], uint32_t lane, uint32_t (&res)[NT]) {
    const uint32_t l = lane & 3u;
    int32_t tw[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint32_t v = pow256_mod(M, i);
#pragma unroll
        for (int k = 1; k < 4; k++) v = l == (uint32_t)k ? pow256_mod(M, i + k) : v;
        tw[i] = (int32_t)v;
    }
    const bool odd = (l & 1u) != 0, hi = (l & 2u) != 0;
#pragma unroll
    for (int t = 0; t < NT; t++) {
        int64_t u[4];
#pragma unroll
        for (int e = 0; e < 4; e++)
            u[e] = (int64_t)acc[0][t][e] * tw[0] + (int64_t)acc[1][t][e] * tw[1] + (int64_t)acc[2][t][e] * tw[2] + (int64_t)acc[3][t][e] * tw[3];
        // step 1 with lane ^ 1: I keep rows e = (l & 1) and (l & 1) + 2 and give the other two
        const int64_t s0 = (odd ? u[1] : u[0]) + quad_dpp<0xB1>(odd ? u[0] : u[1]), s1 = (odd ? u[3] : u[2]) + quad_dpp<0xB1>(odd ? u[2] : u[3]);
        // step 2 with lane ^ 2: I keep row e = l (s0 in the low half of the quad, s1 in the high half)
        const int64_t sum = (hi ? s1 : s0) + quad_dpp<0x4E>(hi ? s0 : s1);
        res[t] = mod_est<M>((uint64_t)(sum + (int64_t)((uint64_t)M << 30)));  // |sum| < 2^57.2 < M 2^30
    }
}

// The query limbs of one piece (work item, prime, 128 terms) are staged in one of two LDS buffers laid out [chunk of 64 terms][n-tile][lane][16 B],
// lane l of a tile = (column n & 15 = l & 15, term block l >> 4), column n = 12 q + 4 r + limb.  A thread takes R items = (query, row, four
// consecutive terms): four 28-bit values -> 4 x 4 limb bytes, transposed with v_perm into one dword per limb.  What a thread's items are does not
// change from piece to piece: source pointer (the query's records + the item's offset inside a piece's block) and LDS offset are computed once.
// The loads are issued one piece ahead of the stores so that no wave waits for them.
template <int NT, int R>
struct RecPlan {
    const uint32_t* src[R];
    uint32_t lds[R];  // dword index in a limb buffer; ~0u: this slot has no item
    uint32_t v[R][4];
    __device__ __forceinline__ void init(const SweepLanes& bt, uint32_t nb) {
        const uint32_t n_items = nb * 96u;  // 3 rows x 32 groups of four terms per query
#pragma unroll
        for (int i = 0; i < R; i++) {
            const uint32_t t0 = threadIdx.x + (uint32_t)i * blockDim.x, t = min(t0, n_items - 1u);
            const uint32_t r = t % 3u, q = (t / 3u) % nb, k4 = t / (3u * nb);  // terms 4 k4 .. + 3 = (j, m = 0, 1), (j + 1, m = 0, 1), j = 2 k4
            src[i] = pick_lane(bt.qs, q) + (size_t)k4 * 24u + r;
            const uint32_t n = q * 12u + r * 4u, k = k4 * 4u;
            lds[i] = t0 < n_items ? ((((k >> 6) * NT + (n >> 4)) * 64u + ((k >> 4) & 3u) * 16u + (n & 15u)) << 2) + ((k & 15u) >> 2) : ~0u;
        }
    }
    // every slot loads (threads beyond the last item repeat it): a fixed number of loads per piece keeps the compiler's vmcnt bookkeeping exact
    __device__ __forceinline__ void issue(uint32_t block) {  // block: u32 offset of (z, first j of the piece, prime) in a query's records
#pragma unroll
        for (int i = 0; i < R; i++) {
            const uint32_t* rec = src[i] + block;
#ifdef SWEEP_ABL_QUAD  // ablation (tools/build_variants.sh): the time the kernel would take if a thread's four values arrived as one 16-byte load of ready limbs; results are garbage
            const uint4 x = *reinterpret_cast<const uint4*>(reinterpret_cast<uintptr_t>(rec) & ~(uintptr_t)15);
            v[i][0] = x.x, v[i][1] = x.y, v[i][2] = x.z, v[i][3] = x.w;
#else
            v[i][0] = rec[0], v[i][1] = rec[6], v[i][2] = rec[12], v[i][3] = rec[18];
#endif
        }
    }
    __device__ __forceinline__ void store(uint4* buf) const {
        uint32_t* out = reinterpret_cast<uint32_t*>(buf);
#pragma unroll
        for (int i = 0; i < R; i++) {
            if (lds[i] != ~0u) {
#ifdef SWEEP_ABL_QUAD
                uint32_t* oa = out + lds[i];
                oa[0] = v[i][0], oa[4] = v[i][1], oa[8] = v[i][2], oa[12] = v[i][3];
                continue;
#endif
                const uint32_t x0 = (v[i][0] + kLimbBias) ^ kLimbBias, x1 = (v[i][1] + kLimbBias) ^ kLimbBias, x2 = (v[i][2] + kLimbBias) ^ kLimbBias,
                               x3 = (v[i][3] + kLimbBias) ^ kLimbBias;
                const uint32_t a_lo = __builtin_amdgcn_perm(x1, x0, 0x05010400u), a_hi = __builtin_amdgcn_perm(x1, x0, 0x07030602u);
                const uint32_t b_lo = __builtin_amdgcn_perm(x3, x2, 0x05010400u), b_hi = __builtin_amdgcn_perm(x3, x2, 0x07030602u);
                uint32_t* o = out + lds[i];
                o[0] = __builtin_amdgcn_perm(b_lo, a_lo, 0x05040100u);  // limbs 0 .. 3 are columns n .. n + 3: the next lanes' 16 bytes (12 q + 4 r
                o[4] = __builtin_amdgcn_perm(b_lo, a_lo, 0x07060302u);  // never straddles a tile)
                o[8] = __builtin_amdgcn_perm(b_hi, a_hi, 0x05040100u);
                o[12] = __builtin_amdgcn_perm(b_hi, a_hi, 0x07060302u);
            }
        }
    }
};

// NT: n-tiles of 16 columns, 12 per query (NT = 1 .. 6 for up to 1, 2, 4, 5, 6, 8 queries).  512 threads = 8 waves = 128 columns of one z.
// Workgroups are persistent, one per CU: each takes a contiguous run of work items w = (group of 128 columns, z), z fastest, and a wave's stream of
// pieces and the pieces' query limbs run ahead across item boundaries, so the database stream never drains (a workgroup per item, with its limbs
// built up front, reached 430 - 500 us).  One loop trip = one piece, straight-line: every global load of a trip (7 of the database, 4 R of
// records) is issued unconditionally, because a load behind a branch makes the compiler's s_waitcnt insertion fall back to vmcnt(0) at the join.
// The only conditional memory operations are the accumulator stores, once per 2^zs_log items.  (Two and four pieces per trip -- 14 and 28 KiB in
// flight per wave -- measured 2 - 5 % slower than one: the registers are better spent elsewhere.)
// dynamic LDS = 2 limb buffers x 2 NT KiB + 8 x NT x 64 x 2^zs_log result words.
template <int NT>
__global__ __launch_bounds__(512, 2) void sweep_mfma_kernel(const uint4* __restrict__ dbl, SweepLanes bt, uint32_t nb, uint32_t nic, uint32_t dim0, uint32_t g_log,
                                                            uint32_t ls_log, uint32_t n_work, uint32_t zs_log) {
    extern __shared__ __attribute__((aligned(16))) uint4 bq[];
    constexpr int R = NT >= 5 ? 2 : 1;  // 96 items per query and piece over 512 threads: one each up to five queries
    constexpr uint32_t buf_sz = 2u * NT * 64u, W = 8;
    constexpr bool kZeroC = NT >= 6;
    const uint32_t lane = threadIdx.x & 63u, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nk2 = dim0 >> 6, ppi = 2u * nk2, ppi_log = 31u - (uint32_t)__builtin_clz(ppi);  // ppi: pieces per item (both primes), a power of two
    const uint32_t per = (n_work + gridDim.x - 1u) / gridDim.x, w0 = min(blockIdx.x * per, n_work), w1 = min(w0 + per, n_work);
    if (w0 == w1) return;
    const uint32_t total = (w1 - w0) << ppi_log;
    const u32x4* const db0 = reinterpret_cast<const u32x4*>(dbl) + lane;
    auto piece_ptr = [&](uint32_t g) -> const u32x4* {  // piece g of this workgroup's run (clamped to its last one)
        g = min(g, total - 1u);
        const uint32_t w = w0 + (g >> ppi_log), p = g & (ppi - 1u), z = w & (kN - 1u), icb = (w >> kLogN) * W + wv;
        return db0 + ((((size_t)z * (nic >> 4) + icb) << ppi_log) + p) * (7u * 64u);
    };
    RecPlan<NT, R> rec;
    rec.init(bt, nb);
    auto rec_issue = [&](uint32_t g) {  // the record loads of piece g of the run (clamped to its last one)
        g = min(g, total - 1u);
        const uint32_t w = w0 + (g >> ppi_log), p = g & (ppi - 1u), z = w & (kN - 1u);
        rec.issue((z * dim0 + (p >= nk2 ? p - nk2 : p) * 64u) * 12u + (p >= nk2 ? 3u : 0u));
    };
    u32x4 d[7];
    {
        const u32x4* src = piece_ptr(0u);
#pragma unroll
        for (uint32_t k = 0; k < 7; k++) d[k] = __builtin_nontemporal_load(src + k * 64u);
    }
    rec_issue(0u);
    rec.store(bq);
    rec_issue(1u);
    uint32_t res0[NT];
    v4i acc[4][NT];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int t = 0; t < NT; t++) acc[i][t] = v4i{0, 0, 0, 0};
    for (uint32_t g = 0; g < total; g++) {
        __syncthreads();  // piece g's buffer is complete; everyone is done reading the other one
        if (g + 1u < total) rec.store(bq + ((g + 1u) & 1u) * buf_sz);
        rec_issue(g + 2u);
        const uint4* bbuf = bq + (g & 1u) * buf_sz + lane;
        v4i a[4][2];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int c = 0; c < 2; c++) a[i][c] = v4i{(int)d[2 * i + c].x, (int)d[2 * i + c].y, (int)d[2 * i + c].z, (int)d[2 * i + c].w};
        const u32x4 nib = d[6];
        a[3][0] = v4i{(int)(nib.x & 0x0F0F0F0Fu), (int)(nib.y & 0x0F0F0F0Fu), (int)(nib.z & 0x0F0F0F0Fu), (int)(nib.w & 0x0F0F0F0Fu)};
        a[3][1] = v4i{(int)((nib.x >> 4) & 0x0F0F0F0Fu), (int)((nib.y >> 4) & 0x0F0F0F0Fu), (int)((nib.z >> 4) & 0x0F0F0F0Fu), (int)((nib.w >> 4) & 0x0F0F0F0Fu)};
        // the first piece of a prime starts the sums.  NT = 6: its chunk-0 products take the constant 0 as C (the second copy of the loop costs the
        // narrower kernels 3 %, the 96 v_movs it saves cost this one 6 %); NT < 6: the registers are zeroed after the prime's last piece
        if (kZeroC && (g & (nk2 - 1u)) == 0) {
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const uint4 bv = bbuf[t * 64];
                const v4i b = v4i{(int)bv.x, (int)bv.y, (int)bv.z, (int)bv.w};
#pragma unroll
                for (int i = 0; i < 4; i++) acc[i][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i][0], b, v4i{0, 0, 0, 0}, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const uint4 bv = bbuf[t * 64];
                const v4i b = v4i{(int)bv.x, (int)bv.y, (int)bv.z, (int)bv.w};
#pragma unroll
                for (int i = 0; i < 4; i++) acc[i][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i][0], b, acc[i][t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < NT; t++) {
            const uint4 bv = bbuf[(NT + t) * 64];
            const v4i b = v4i{(int)bv.x, (int)bv.y, (int)bv.z, (int)bv.w};
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i][1], b, acc[i][t], 0, 0, 0);
        }
        {
            const u32x4* src = piece_ptr(g + 1u);  // the next piece, in flight under this one's products
#pragma unroll
            for (uint32_t k = 0; k < 7; k++) d[k] = __builtin_nontemporal_load(src + k * 64u);
        }
        const uint32_t p_end = (g + 1u) & (ppi - 1u);
        if (p_end == nk2) {
            combine_limbs<NT, kP>(acc, lane, res0);
        } else if (p_end == 0) {
            uint32_t res1[NT];
            combine_limbs<NT, kB>(acc, lane, res1);
            // A lane's result (column ic, query q, row r) is one 8-byte word of accumulator polynomial (q; 6 ii + 2 r + c) at slot z: 16 KiB from
            // the next lane's.  The wave parks the words of 2^zs_log consecutive z in its own LDS rows [tile][lane][z] and then writes them as
            // 8 * 2^zs_log-byte runs, 16 bytes per lane (scattered 8-byte stores cost 80 - 130 us per launch at config 2).
            const uint32_t w = w0 + (g >> ppi_log), z = w & (kN - 1u), zs = 1u << zs_log, zi = z & (zs - 1u);
            uint64_t* const st = reinterpret_cast<uint64_t*>(bq + 2u * buf_sz) + (size_t)wv * (NT * 64u << zs_log);
#pragma unroll
            for (int t = 0; t < NT; t++) st[(((uint32_t)t * 64u + lane) << zs_log) + zi] = pack(res0[t], res1[t]);
            if (zi == zs - 1u) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const uint32_t icb = (w >> kLogN) * W + wv, half_log = zs_log - 1u;  // 2^half_log 16-byte pieces per entry
                for (uint32_t idx = lane; idx < (NT * 64u) << half_log; idx += 64u) {
                    const uint32_t e = idx >> half_log, part = idx & ((1u << half_log) - 1u), t = e >> 6, ls = e & 63u;  // entry = (tile, source lane)
                    const uint32_t ic = icb * 16u + (ls >> 4) * 4u + (ls & 3u), i0 = ic >> 1, c = ic & 1u, ii = acc_pos(i0, g_log, ls_log);
                    const uint32_t qr = t * 4u + ((ls & 15u) >> 2), q = qr / 3u, r = qr - q * 3u;
                    const uint4 v = reinterpret_cast<const uint4*>(st)[idx];
                    if (q < nb) *reinterpret_cast<uint4*>(pick_lane(bt.acc, q) + ((size_t)(6u * ii + 2u * r + c)) * kN + (z - zi) + 2u * part) = v;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (!kZeroC && (p_end == nk2 || p_end == 0)) {
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int t = 0; t < NT; t++) acc[i][t] = v4i{0, 0, 0, 0};
        }
    }
}

bool sweep_mfma_ok(uint32_t num_per, uint32_t jm_total) {
    // whole workgroups of 128 columns, whole pieces of 128 terms; K = 2 dim0 <= 2^12 (combine_limbs' 64-bit sums); the kernel walks a work item's pieces
    // with shifts and masks (ppi_log, nk2 - 1): the first dimension -- a shard [j0, j1) may be any range -- must be a power of two, other shards take the
    // vector-ALU passes
    const uint32_t nic = 2 * num_per, dim0 = jm_total / 2;
    return nic >= 128 && db_packed(nic, dim0) && (dim0 & 63u) == 0 && (dim0 & (dim0 - 1u)) == 0 && dim0 <= 2048u;
}
void launch_db_limb_planes(const uint64_t* db_packed_img, uint64_t* db_limbs, uint32_t num_per, uint32_t jm_total, hipStream_t s, uint32_t nz) {
    const uint32_t nic = 2 * num_per, dim0 = jm_total / 2;
    const size_t waves = (size_t)nz * (nic >> 4) * (dim0 >> 6);
    if (waves) hipLaunchKernelGGL(db_limb_planes_kernel, dim3((uint32_t)((waves + 3) / 4)), dim3(256), 0, s, db_packed_img, reinterpret_cast<uint4*>(db_limbs), nic, dim0, nz);
}
void launch_db_limb_unplanes(const uint64_t* db_limbs, uint64_t* db_packed_img, uint32_t num_per, uint32_t jm_total, hipStream_t s, uint32_t nz) {
    const uint32_t nic = 2 * num_per, dim0 = jm_total / 2;
    const size_t waves = (size_t)nz * (nic >> 4) * (dim0 >> 6);
    if (waves) hipLaunchKernelGGL(db_limb_unplanes_kernel, dim3((uint32_t)((waves + 3) / 4)), dim3(256), 0, s, reinterpret_cast<const uint4*>(db_limbs), db_packed_img, nic, dim0, nz);
}
hipError_t launch_sweep_mfma(const uint64_t* db_limbs, const uint32_t* const* qs, uint64_t* const* acc, uint32_t n, uint32_t num_per, uint32_t jm_total, uint32_t g_log,
                             hipStream_t s, uint32_t k_log) {
    const uint32_t nic = 2 * num_per, dim0 = jm_total / 2;
    uint32_t ls_log = 0;
    while ((1u << ls_log) < num_per) ls_log++;
    ls_log -= g_log + k_log;
    SweepLanes bt{};
    for (uint32_t b = 0; b < kMaxLanes; b++) {
        bt.qs[b] = qs[b < n ? b : 0];
        bt.acc[b] = acc[b < n ? b : 0];
    }
    const uint32_t nt = (12u * n + 15u) / 16u;
    // one workgroup per CU (two per CU with half the staging measured 5 % slower); per = 8 nic / 128 items each
    const uint32_t n_work = kN * (nic >> 7), n_wg = 256u, per = n_work / n_wg;
    // results of 2^zs_log consecutive z are staged per wave (8 x nt x 64 x 2^zs_log words) beside the two limb buffers: 8 if that fits the CU's LDS
    const size_t lds_b = (size_t)2u * 2u * nt * 1024u;
    uint32_t zs_log = 3;
    while (zs_log > 1 && ((1u << zs_log) > per || lds_b + ((size_t)8u * nt * 64u * 8u << zs_log) > 160u * 1024u)) zs_log--;
    const size_t lds = lds_b + ((size_t)8u * nt * 64u * 8u << zs_log);
    const dim3 grid(n_wg), block(512);
    const uint4* dbl = reinterpret_cast<const uint4*>(db_limbs);
    // more than 64 KiB of dynamic LDS has to be asked for per kernel AND per device: a process may drive servers on several GPUs, from several
    // threads, so the opt-in is remembered per (instance, device) in an atomic bit mask (devices beyond 63 ask every time)
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
#define SWEEP_MFMA(NTV)                                                                                                                            \
    do {                                                                                                                                           \
        static std::atomic<uint64_t> big{0};                                                                                                       \
        const uint64_t bit = dev < 64 ? 1ull << dev : 0ull;                                                                                        \
        if (!(big.load(std::memory_order_relaxed) & bit)) {                                                                                        \
            e = hipFuncSetAttribute((const void*)sweep_mfma_kernel<NTV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                  \
            if (e != hipSuccess) return e;                                                                                                         \
            big.fetch_or(bit, std::memory_order_relaxed);                                                                                          \
        }                                                                                                                                          \
        hipLaunchKernelGGL((sweep_mfma_kernel<NTV>), grid, block, lds, s, dbl, bt, n, nic, dim0, g_log, ls_log, n_work, zs_log);                  \
    } while (0)
    switch (nt) {
        case 1: SWEEP_MFMA(1); break;
        case 2: SWEEP_MFMA(2); break;
        case 3: SWEEP_MFMA(3); break;
        case 4: SWEEP_MFMA(4); break;
        case 5: SWEEP_MFMA(5); break;
        case 6: SWEEP_MFMA(6); break;
        default: return hipErrorInvalidValue;
    }
#undef SWEEP_MFMA
    return hipGetLastError();  // a launch that was refused (LDS, grid) is reported here, not at some later synchronisation
}

}  // namespace spiral
