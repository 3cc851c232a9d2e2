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

// fast path (num_per >= 64): one wave per (z, block of 64 plaintext columns); lane = column; a 16-byte load brings
// two consecutive j; the two query records of the pair are wave-uniform (SGPRs)
constexpr uint32_t kSweep1Waves = 2;
__global__ __launch_bounds__(kSweep1Waves * 64) void sweep1_kernel(const uint64_t* __restrict__ db, const uint32_t* __restrict__ qs,
                                                                   uint64_t* __restrict__ acc, uint32_t num_per, uint32_t dim0) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * kSweep1Waves + (threadIdx.x >> 6));
    const uint32_t wpz = num_per >> 6, z = wave / wpz, ii = (wave - z * wpz) * 64u + lane, jp_n = dim0 >> 1;
    const u64x2* dbp = reinterpret_cast<const u64x2*>(db) + (size_t)wave * jp_n * 64u + lane;
    const uint4* q = reinterpret_cast<const uint4*>(qs) + (size_t)z * dim0;
    uint64_t a[4] = {0, 0, 0, 0};
    for (uint32_t j0 = 0; j0 < jp_n; j0 += 128) {  // 128 pairs = 256 terms per accumulator between reductions
        const uint32_t jend = min(j0 + 128u, jp_n);
#pragma unroll 8
        for (uint32_t jp = j0; jp < jend; jp++) {
            const u64x2 w = __builtin_nontemporal_load(dbp + (size_t)jp * 64u);
            mac4(a, q[2 * jp], w.x);
            mac4(a, q[2 * jp + 1], w.y);
        }
        reduce4(a);
    }
    store_acc1(acc, a, ii, z);
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
void launch_sweep1(const uint64_t* db, const uint32_t* qs1, uint64_t* acc, uint32_t num_per, uint32_t dim0, hipStream_t s) {
    if (num_per >= 64) {
        const uint32_t waves = kN * (num_per >> 6);
        hipLaunchKernelGGL(sweep1_kernel, dim3(waves / kSweep1Waves), dim3(kSweep1Waves * 64), 0, s, db, qs1, acc, num_per, dim0);
    } else {
        hipLaunchKernelGGL(sweep1_small_kernel, dim3((kN * num_per + 255) / 256), dim3(256), 0, s, db, qs1, acc, num_per, dim0);
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
    dev[db1_word_index(pk_pos(z), j, ii, num_per, dim0)] = ref[o];
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
    uint64_t lo = 0, hi = 0;
    for (uint32_t r = 0; r < out_n; r++) {
        const uint64_t* w = v_w + (((size_t)r * rows + row) * t_conv) * kN + z;
        const uint64_t* g = ginv + ((size_t)(r * out_n + c) * t_conv) * kN + z;
        for (uint32_t k = 0; k < t_conv; k++) {
            const uint64_t a = w[(size_t)k * kN], b = g[(size_t)k * kN];
            lo += (uint64_t)lo32(a) * lo32(b);
            hi += (uint64_t)hi32(a) * hi32(b);
        }
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

}  // namespace spiral
