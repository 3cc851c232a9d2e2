// SpiralPack / SpiralStreamPack kernels (reference src/testing.cpp, `--high-rate`): base_dim x 1 scalar Regev
// ciphertexts against 1 x 1 plaintexts.  The first-dimension sweep is again the HBM-bound kernel: 4 integer MADs
// per 8-byte database word.
#include "common.h"
#include "kernels.h"

namespace spiral {

typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
constexpr uint32_t kTpb = 256, kBpp = kN / kTpb;

__device__ __forceinline__ void mac4(uint64_t (&a)[4], uint4 q, uint64_t w) {
    const uint32_t bl = lo32(w), bh = hi32(w);
    a[0] += (uint64_t)q.x * bl;
    a[1] += (uint64_t)q.y * bl;
    a[2] += (uint64_t)q.z * bh;
    a[3] += (uint64_t)q.w * bh;
}
__device__ __forceinline__ void reduce4(uint64_t (&a)[4]) {
    a[0] = mod_p(a[0]);
    a[1] = mod_p(a[1]);
    a[2] = mod_b(a[2]);
    a[3] = mod_b(a[3]);
}
__device__ __forceinline__ void store_acc1(uint64_t* acc, const uint64_t (&a)[4], uint32_t ii, uint32_t z) {
    acc[((size_t)ii * 2u) * kN + z] = pack((uint32_t)a[0], (uint32_t)a[2]);
    acc[((size_t)ii * 2u + 1u) * kN + z] = pack((uint32_t)a[1], (uint32_t)a[3]);
}

// fast path (packed database): one wave per (z, block of 64 plaintext columns), lane = column; per 16 j seven 16-byte loads
// (112 packed bytes = 16 words), the query records of those j are wave-uniform (SGPRs).  As the base sweep (sweep.hip): 16
// waves of a workgroup work on consecutive z and trade their results through LDS to store full 128-byte lines, and the
// block -> tile mapping keeps the workgroups that share a z-group's query records on one XCD.
constexpr uint32_t kSweep1Z = 16, kSweep1Row = 64 * 2 + 1;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int W>
__device__ __forceinline__ void mac4_packed(uint64_t (&a)[4], uint4 q, const uint32_t (&d)[28]) {
    const uint32_t bl = field28<2 * W>(d), bh = field28<2 * W + 1>(d);
    a[0] += (uint64_t)q.x * bl;
    a[1] += (uint64_t)q.y * bl;
    a[2] += (uint64_t)q.z * bh;
    a[3] += (uint64_t)q.w * bh;
}
// MODE 0 (num_per >= 64): a wave is one slot z and 64 plaintext columns, query records wave-uniform.  MODE 1, 2: a wave is
// P = 64/num_per consecutive slots x num_per columns and every lane needs the records of its own z: loaded per lane
// (MODE 1) or staged per wave in LDS (MODE 2, P <= 8), as sweep.hip.
constexpr uint32_t kQ1Stage = 8 * 16;  // uint4 per wave: up to 8 slots x 16 j
template <int MODE>
__global__ __launch_bounds__(kSweep1Z * 64) void sweep1_kernel(const uint64_t* __restrict__ db, const uint32_t* __restrict__ qs,
                                                               uint64_t* __restrict__ acc, uint32_t num_per, uint32_t dim0, size_t db_stride,
                                                               size_t acc_stride) {
    constexpr bool WIDE = MODE == 0;
    constexpr uint32_t kShWords = MODE == 2 ? (kSweep1Z * kQ1Stage * 2 > kSweep1Z * kSweep1Row ? kSweep1Z * kQ1Stage * 2 : kSweep1Z * kSweep1Row)
                                            : kSweep1Z * kSweep1Row;
    __shared__ __attribute__((aligned(16))) uint64_t sh[kShWords];  // results; MODE 2: first the record staging (aliased)
    db += (size_t)blockIdx.y * db_stride;  // blockIdx.y = trial: all trials of a query in one launch (same query records)
    acc += (size_t)blockIdx.y * acc_stride;
    const uint32_t lane = threadIdx.x & 63u, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t w = WIDE ? 64u : num_per, pz = 64u / w, nblk = num_per / w;
    uint32_t work = blockIdx.x;
    if ((gridDim.x & 7u) == 0) work = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);  // XCD-aware, see sweep.hip
    const uint32_t zg = work / nblk, iib = work - zg * nblk, groups = dim0 >> 4;
    // !WIDE: one workgroup per tile, its waves split the tile's j range and add their partial sums through LDS (sweep.hip)
    const uint32_t ztile = WIDE ? zg * kSweep1Z + wv : zg, tile = ztile * nblk + iib;
    const uint32_t gper = WIDE ? groups : (groups + kSweep1Z - 1u) / kSweep1Z;
    const uint32_t gfirst = WIDE ? 0u : min(wv * gper, groups), glast = WIDE ? groups : min(gfirst + gper, groups);
    const uint32_t z = WIDE ? ztile : ztile * pz + lane / w;
    const u32x4* dbp = reinterpret_cast<const u32x4*>(db) + (size_t)tile * groups * 7u * 64u + lane;
    const uint4* q = reinterpret_cast<const uint4*>(qs) + (size_t)z * dim0;
    uint64_t a[4] = {0, 0, 0, 0};
    for (uint32_t g0 = gfirst; g0 < glast; g0 += 16) {  // 16 groups = 256 j = 256 terms per accumulator between reductions
        const uint32_t gend = min(g0 + 16u, glast);
#pragma unroll 2
        for (uint32_t g = g0; g < gend; g++) {
            uint32_t d[28];
#pragma unroll
            for (uint32_t k = 0; k < 7; k++) {
                const u32x4 v = __builtin_nontemporal_load(dbp + ((size_t)g * 7u + k) * 64u);
                d[4 * k] = v.x;
                d[4 * k + 1] = v.y;
                d[4 * k + 2] = v.z;
                d[4 * k + 3] = v.w;
            }
            const uint4* qg = q + (size_t)g * 16u;
            if constexpr (MODE == 2) {
                uint4* qst = reinterpret_cast<uint4*>(sh) + wv * kQ1Stage;
                const uint4* qsrc = reinterpret_cast<const uint4*>(qs);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();  // the previous group's reads of the staging area
                for (uint32_t t = lane; t < pz * 16u; t += 64u) {
                    const uint32_t zi = t >> 4, off = t & 15u;
                    qst[t] = qsrc[(size_t)(ztile * pz + zi) * dim0 + (size_t)g * 16u + off];
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                qg = qst + (lane / w) * 16u;
            }
            mac4_packed<0>(a, qg[0], d);
            mac4_packed<1>(a, qg[1], d);
            mac4_packed<2>(a, qg[2], d);
            mac4_packed<3>(a, qg[3], d);
            mac4_packed<4>(a, qg[4], d);
            mac4_packed<5>(a, qg[5], d);
            mac4_packed<6>(a, qg[6], d);
            mac4_packed<7>(a, qg[7], d);
            mac4_packed<8>(a, qg[8], d);
            mac4_packed<9>(a, qg[9], d);
            mac4_packed<10>(a, qg[10], d);
            mac4_packed<11>(a, qg[11], d);
            mac4_packed<12>(a, qg[12], d);
            mac4_packed<13>(a, qg[13], d);
            mac4_packed<14>(a, qg[14], d);
            mac4_packed<15>(a, qg[15], d);
        }
        reduce4(a);
    }
    if constexpr (MODE == 2) __syncthreads();  // every wave is done with its staging area before results overwrite it
    sh[wv * kSweep1Row + lane * 2u] = pack((uint32_t)a[0], (uint32_t)a[2]);
    sh[wv * kSweep1Row + lane * 2u + 1u] = pack((uint32_t)a[1], (uint32_t)a[3]);
    __syncthreads();
    if constexpr (WIDE) {  // 128 (column, row) results x 16 consecutive z: thread -> (result, z) with z fastest; acc[ii][r][z]
#pragma unroll
        for (uint32_t m = 0; m < 2; m++) {
            const uint32_t idx = threadIdx.x + kSweep1Z * 64u * m, res = idx / kSweep1Z, zz = idx - res * kSweep1Z;
            acc[((size_t)(iib * 64u) * 2u + res) * kN + zg * kSweep1Z + zz] = sh[zz * kSweep1Row + res];
        }
    } else if (threadIdx.x < 128u) {  // 64 lanes x 2 results of this tile, each the sum of the waves' partials
        const uint32_t sl = threadIdx.x >> 1, r = threadIdx.x & 1u, zz = sl / w, col = sl - zz * w;
        uint64_t sp = 0, sb = 0;
#pragma unroll
        for (uint32_t v = 0; v < kSweep1Z; v++) {
            const uint64_t x = sh[v * kSweep1Row + threadIdx.x];
            sp += lo32(x);
            sb += hi32(x);
        }
        acc[((size_t)col * 2u + r) * kN + zg * pz + zz] = pack(mod_p(sp), mod_b(sb));
    }
}
__global__ __launch_bounds__(256) void sweep1_small_kernel(const uint64_t* __restrict__ db, const uint32_t* __restrict__ qs, uint64_t* __restrict__ acc,
                                                           uint32_t num_per, uint32_t dim0) {
    const uint32_t g = blockIdx.x * 256u + threadIdx.x, z = g / num_per, ii = g - z * num_per;
    if (z >= kN) return;
    const uint4* q = reinterpret_cast<const uint4*>(qs) + (size_t)z * dim0;
    uint64_t a[4] = {0, 0, 0, 0};
    for (uint32_t j = 0; j < dim0; j++) {
        mac4(a, q[j], db[db1_word_index(z, j, ii, num_per, dim0)]);
        if ((j & 255u) == 255u) reduce4(a);
    }
    reduce4(a);
    store_acc1(acc, a, ii, z);
}
void launch_sweep1(const uint64_t* db, const uint32_t* qs1, uint64_t* acc, uint32_t num_per, uint32_t dim0, uint32_t trials, size_t db_stride,
                   size_t acc_stride, hipStream_t s) {
    if (trials == 0) return;
    if (db1_packed(num_per, dim0)) {
        if (num_per >= 64)
            hipLaunchKernelGGL(sweep1_kernel<0>, dim3((kN / kSweep1Z) * (num_per >> 6), trials), dim3(kSweep1Z * 64), 0, s, db, qs1, acc, num_per, dim0,
                               db_stride, acc_stride);
        else if (num_per >= 8)  // one workgroup per tile of 64/num_per <= 8 slots, its waves split the j range; records staged in LDS
            hipLaunchKernelGGL(sweep1_kernel<2>, dim3(kN / (64 / num_per), trials), dim3(kSweep1Z * 64), 0, s, db, qs1, acc, num_per, dim0, db_stride,
                               acc_stride);
        else
            hipLaunchKernelGGL(sweep1_kernel<1>, dim3(kN / (64 / num_per), trials), dim3(kSweep1Z * 64), 0, s, db, qs1, acc, num_per, dim0, db_stride,
                               acc_stride);
    } else {
        for (uint32_t t = 0; t < trials; t++)
            hipLaunchKernelGGL(sweep1_small_kernel, dim3((kN * num_per + 255) / 256), dim3(256), 0, s, db + t * db_stride, qs1, acc + t * acc_stride, num_per,
                               dim0);
    }
}

__global__ __launch_bounds__(kTpb) void qs1_from_cv_kernel(const uint64_t* cv, uint32_t* qs, uint32_t dim0, uint32_t idx_factor) {
    const uint32_t z = blockIdx.x * kTpb + threadIdx.x, j = blockIdx.y;
    const uint64_t* c = cv + (size_t)j * idx_factor * 2 * kN + z;
    const uint64_t r0 = c[0], r1 = c[kN];
    reinterpret_cast<uint4*>(qs)[(size_t)z * dim0 + j] = make_uint4(lo32(r0), lo32(r1), hi32(r0), hi32(r1));
}
void launch_qs1_from_cv(const uint64_t* cv, uint32_t* qs1, uint32_t dim0, uint32_t idx_factor, hipStream_t s) {
    hipLaunchKernelGGL(qs1_from_cv_kernel, dim3(kBpp, dim0), dim3(kTpb), 0, s, cv, qs1, dim0, idx_factor);
}
// reference layout (z, j, m = 0, r): z*(dim0*2) + j*2 + r
__global__ __launch_bounds__(256) void qs1_from_reoriented_kernel(const uint64_t* re, uint32_t* qs, uint32_t dim0) {
    const size_t g = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (g >= (size_t)kN * dim0) return;
    const uint64_t r0 = re[g * 2], r1 = re[g * 2 + 1];
    const uint32_t z = (uint32_t)(g / dim0), j = (uint32_t)(g - (size_t)z * dim0);  // z in the reference's slot order
    reinterpret_cast<uint4*>(qs)[(size_t)pk_pos(z) * dim0 + j] = make_uint4(lo32(r0) % kP, lo32(r1) % kP, hi32(r0) % kB, hi32(r1) % kB);
}
void launch_qs1_from_reoriented(const uint64_t* re, uint32_t* qs1, uint32_t dim0, hipStream_t s) {
    const size_t n = (size_t)kN * dim0;
    hipLaunchKernelGGL(qs1_from_reoriented_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, re, qs1, dim0);
}

__global__ __launch_bounds__(256) void db1_relayout_kernel(const uint64_t* __restrict__ ref, uint64_t* __restrict__ dev, uint32_t num_per, uint32_t dim0) {
    const size_t o = (size_t)blockIdx.x * 256u + threadIdx.x, per_z = (size_t)num_per * dim0;
    const uint32_t z = (uint32_t)(o / per_z);
    if (z >= kN) return;
    const size_t rem = o - (size_t)z * per_z;
    const uint32_t ii = (uint32_t)(rem / dim0), j = (uint32_t)(rem % dim0);
    const uint64_t v = ref[o];
    db1_put_word(dev, pk_pos(z), j, ii, num_per, dim0, pack(lo32(v) % kP, hi32(v) % kB));
}
void launch_db1_relayout(const uint64_t* ref, uint64_t* dev, uint32_t num_per, uint32_t dim0, hipStream_t s) {
    const size_t words = (size_t)kN * num_per * dim0;
    hipLaunchKernelGGL(db1_relayout_kernel, dim3((uint32_t)((words + 255) / 256)), dim3(256), 0, s, ref, dev, num_per, dim0);
}

__global__ __launch_bounds__(kTpb) void pack_gsw_assemble_kernel(const uint64_t* tmp, const uint64_t* cv, uint64_t* gsw, uint32_t ell) {
    const uint32_t z = blockIdx.x * kTpb + threadIdx.x, ij = blockIdx.y, i = ij / ell, j = ij - i * ell, cols = 2 * ell;
#pragma unroll
    for (uint32_t r = 0; r < 2; r++) {
        uint64_t* g = gsw + ((size_t)(i * 2 + r) * cols) * kN + z;
        g[(size_t)(2 * j) * kN] = tmp[((size_t)ij * 2 + r) * kN + z];
        g[(size_t)(2 * j + 1) * kN] = cv[((size_t)(2 * ij + 1) * 2 + r) * kN + z];
    }
}
void launch_pack_gsw_assemble(const uint64_t* tmp, const uint64_t* cv, uint64_t* gsw, uint32_t ell, uint32_t nu2, hipStream_t s) {
    hipLaunchKernelGGL(pack_gsw_assemble_kernel, dim3(kBpp, nu2 * ell), dim3(kTpb), 0, s, tmp, cv, gsw, ell);
}

// direct upload (src/testing.cpp:966-989): gsw[i][r][col] = uploaded ct (dim0 + i*2ell + col), row r
__global__ __launch_bounds__(kTpb) void pack_gsw_from_upload_kernel(const uint64_t* query, uint64_t* gsw, uint32_t dim0, uint32_t ell) {
    const uint32_t z = blockIdx.x * kTpb + threadIdx.x, cols = 2 * ell, ic = blockIdx.y, i = ic / cols, col = ic - i * cols;
#pragma unroll
    for (uint32_t r = 0; r < 2; r++) gsw[((size_t)(i * 2 + r) * cols + col) * kN + z] = query[((size_t)(dim0 + ic) * 2 + r) * kN + z];
}
void launch_pack_gsw_from_upload(const uint64_t* query, uint64_t* gsw, uint32_t dim0, uint32_t ell, uint32_t nu2, hipStream_t s) {
    hipLaunchKernelGGL(pack_gsw_from_upload_kernel, dim3(kBpp, nu2 * 2 * ell), dim3(kTpb), 0, s, query, gsw, dim0, ell);
}

// folding_neg = gadget + NTT(Q - INTT(F)) = gadget - F slot-wise; gadget[r][col] = 2^(bits * col/2) when col % 2 == r
__global__ __launch_bounds__(kTpb) void pack_fold_key_kernel(const uint64_t* gsw, uint64_t* key, uint32_t ell, uint32_t nu2) {
    const uint32_t z = blockIdx.x * kTpb + threadIdx.x, cols = 2 * ell, rc = blockIdx.y, r = rc / cols, col = rc - r * cols, cur = blockIdx.z;
    const uint64_t f = gsw[((size_t)((nu2 - 1 - cur) * 2 + r) * cols + col) * kN + z];
    uint32_t gp = 0, gb = 0;
    if ((col & 1u) == r) {
        const uint32_t sh = get_bits_per(ell) * (col >> 1);
        if (sh < 64) {
            gp = mod_p(1ull << sh);
            gb = mod_b(1ull << sh);
        }
    }
    uint64_t* k = key + ((size_t)(cur * 2 + r) * 2 * cols) * kN + z;
    k[(size_t)col * kN] = pack(csub(gp + kP - lo32(f), kP), csub(gb + kB - hi32(f), kB));
    k[(size_t)(cols + col) * kN] = f;
}
void launch_pack_fold_key(const uint64_t* gsw, uint64_t* key, uint32_t ell, uint32_t nu2, hipStream_t s) {
    hipLaunchKernelGGL(pack_fold_key_kernel, dim3(kBpp, 4 * ell, nu2), dim3(kTpb), 0, s, gsw, key, ell, nu2);
}

__global__ __launch_bounds__(kTpb) void pack_mac_kernel(const uint64_t* v_w, const uint64_t* ginv, const uint64_t* ct2, uint64_t* result, uint32_t out_n,
                                                        uint32_t t_conv) {
    const uint32_t z = blockIdx.x * kTpb + threadIdx.x, rc = blockIdx.y, row = rc / out_n, c = rc - row * out_n, rows = out_n + 1;
    // the reference multiplies per r (t_conv <= 56 terms, one reduction, src/poly.cpp:62) and adds the out_n products mod m
    // (src/testing.cpp:225-238); a u64 holds 256 terms of < 2^56, so the sum over r is reduced whenever the next r's terms
    // would pass that (out_n * t_conv goes up to 16 * 56 = 896)
    uint64_t lo = 0, hi = 0;
    uint32_t terms = 0;
    for (uint32_t r = 0; r < out_n; r++) {
        const uint64_t* w = v_w + (((size_t)r * rows + row) * t_conv) * kN + z;
        const uint64_t* g = ginv + ((size_t)(r * out_n + c) * t_conv) * kN + z;
        if (terms + t_conv > 255u) {  // the reduced value counts as one term
            lo = mod_p(lo);
            hi = mod_b(hi);
            terms = 1;
        }
        for (uint32_t k = 0; k < t_conv; k++) {
            const uint64_t a = w[(size_t)k * kN], b = g[(size_t)k * kN];
            lo += (uint64_t)lo32(a) * lo32(b);
            hi += (uint64_t)hi32(a) * hi32(b);
        }
        terms += t_conv;
    }
    uint32_t rp = mod_p(lo), rb = mod_b(hi);
    if (row >= 1) {
        const uint64_t x = ct2[(size_t)((row - 1) * out_n + c) * kN + z];
        rp = csub(rp + lo32(x), kP);
        rb = csub(rb + hi32(x), kB);
    }
    result[(size_t)rc * kN + z] = pack(rp, rb);
}
void launch_pack_mac(const uint64_t* v_w, const uint64_t* ginv, const uint64_t* ct2, uint64_t* result, uint32_t out_n, uint32_t t_conv, hipStream_t s) {
    hipLaunchKernelGGL(pack_mac_kernel, dim3(kBpp, (out_n + 1) * out_n), dim3(kTpb), 0, s, v_w, ginv, ct2, result, out_n, t_conv);
}

// ---- foldCiphertextsDim1 product (src/testing.cpp:596-624): out[b][r] = sum_m key[r][m] * d[b][m], r < 2, m < K = 4*ell.
// Split-K over 4 k-groups + LDS like the base path's fold_mac_kernel; B ciphertexts per workgroup share the key words; the
// digit operand is streamed once (non-temporal loads).
struct PackAcc {
    uint64_t lo = 0, hi = 0;
    __device__ __forceinline__ void mac(uint64_t a, uint64_t b) {
        lo += (uint64_t)lo32(a) * lo32(b);
        hi += (uint64_t)hi32(a) * hi32(b);
    }
};
template <uint32_t B>
__global__ __launch_bounds__(kTpb) void pack_fold_mac_kernel(const uint64_t* __restrict__ key, const uint64_t* __restrict__ d, uint64_t* __restrict__ out,
                                                             uint32_t K, uint32_t ks, const uint64_t* __restrict__ add, uint32_t np, uint32_t add_stride) {
    __shared__ uint64_t sh[3][64][4 * B];
    const uint32_t zz = threadIdx.x & 63u, kg = threadIdx.x >> 6, z = blockIdx.x * 64u + zz, b0 = blockIdx.y * B;
    const uint64_t* dp = d + (size_t)b0 * K * kN + z;
    const uint64_t* kp = key + z;
    PackAcc acc[B][2];
    if (add != nullptr && kg == 0) {  // pair form: the low ciphertext of the pair, requested before the product loop
#pragma unroll
        for (uint32_t b = 0; b < B; b++) {
            const uint32_t t = (b0 + b) / np, i = (b0 + b) - t * np;
#pragma unroll
            for (uint32_t r = 0; r < 2; r++) {
                const uint64_t a = add[((size_t)(t * add_stride + i) * 2 + r) * kN + z];
                acc[b][r].lo = lo32(a);
                acc[b][r].hi = hi32(a);
            }
        }
    }
#pragma unroll 4
    for (uint32_t m = kg; m < K; m += 4) {
        const uint64_t k0 = kp[(size_t)m * kN], k1 = kp[(size_t)(ks + m) * kN];
#pragma unroll
        for (uint32_t b = 0; b < B; b++) {
            const uint64_t dv = __builtin_nontemporal_load(&dp[((size_t)b * K + m) * kN]);
            acc[b][0].mac(k0, dv);
            acc[b][1].mac(k1, dv);
        }
    }
    if (kg > 0) {
#pragma unroll
        for (uint32_t b = 0; b < B; b++)
#pragma unroll
            for (uint32_t r = 0; r < 2; r++) {
                sh[kg - 1][zz][b * 4 + r * 2] = acc[b][r].lo;
                sh[kg - 1][zz][b * 4 + r * 2 + 1] = acc[b][r].hi;
            }
    }
    __syncthreads();
    if (kg == 0) {
#pragma unroll
        for (uint32_t b = 0; b < B; b++)
#pragma unroll
            for (uint32_t r = 0; r < 2; r++) {
                uint64_t lo = acc[b][r].lo, hi = acc[b][r].hi;
#pragma unroll
                for (int q = 0; q < 3; q++) {  // K <= 256 terms of < 2^56 in total
                    lo += sh[q][zz][b * 4 + r * 2];
                    hi += sh[q][zz][b * 4 + r * 2 + 1];
                }
                out[((size_t)(b0 + b) * 2 + r) * kN + z] = pack(mod_p(lo), mod_b(hi));
            }
    }
}
void launch_pack_fold_mac(const uint64_t* key, const uint64_t* d, uint64_t* out, uint32_t K, uint32_t count, hipStream_t s, uint32_t key_stride,
                          const uint64_t* addend, uint32_t np, uint32_t add_stride) {
    if (count == 0) return;
    const uint32_t ks = key_stride ? key_stride : K;
    if (count % 4 == 0 && count >= 64)
        hipLaunchKernelGGL(pack_fold_mac_kernel<4>, dim3(kN / 64, count / 4), dim3(kTpb), 0, s, key, d, out, K, ks, addend, np, add_stride);
    else
        hipLaunchKernelGGL(pack_fold_mac_kernel<1>, dim3(kN / 64, count), dim3(kTpb), 0, s, key, d, out, K, ks, addend, np, add_stride);
}

// arbitrary valid words (benchmarks)
__global__ __launch_bounds__(256) void fill_db1_random_kernel(uint64_t* db, uint32_t num_per, uint32_t dim0, uint64_t seed) {
    const uint64_t nwords = (uint64_t)kN * dim0 * num_per, stride = (uint64_t)gridDim.x * 256u;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < nwords; i += stride) {
        uint64_t x = seed + i + 0x9E3779B97F4A7C15ull;
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
        x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
        x ^= x >> 31;
        const uint32_t ii = (uint32_t)(i % num_per);
        const uint64_t rest = i / num_per;
        db1_put_word(db, (uint32_t)(rest / dim0), (uint32_t)(rest % dim0), ii, num_per, dim0, pack((uint32_t)(x & 0xffffffffull) % kP, (uint32_t)(x >> 32) % kB));
    }
}
void launch_fill_db1_random(uint64_t* db_dev, uint32_t num_per, uint32_t dim0, uint64_t seed, hipStream_t s) {
    hipLaunchKernelGGL(fill_db1_random_kernel, dim3(4096), dim3(256), 0, s, db_dev, num_per, dim0, seed);
}

}  // namespace spiral
