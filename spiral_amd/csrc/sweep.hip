// First-dimension sweep: the HBM-bound kernel of the server path (replaces multiplyQueryByDatabase,
// reference src/spiral.cpp:628-999).
//
//   acc[ii][r][c].limb(z) = ( sum_{j,m} ct_j[r][m].limb(z) * DB[ii, j][m][c].limb(z) ) mod m_limb
//
// Per NTT slot z this is a (nic x JM) by (JM x 3) product per limb with nic = 2*num_per output columns
// (ii, c) and JM = 2*dim0 terms (j, m): 6 integer MADs per 8-byte database word, so the kernel is bound
// by streaming the database once from HBM.  MFMA does not apply (32x32->64-bit modular integer MACs).
//
// Device database layout (built at load time, any re-layout is internal; common.h db_word_index):
//     [z][column block of 64][j][lane][m], u64 = p-limb | b-limb << 32
// so that one wave reads 64 lanes x 16 B = 1 KiB per step (lane = column, both m of one j) and its dim0
// steps are strictly sequential addresses (a dim0 KiB contiguous stream per wave).  The query is stored as one 48-byte record per (z, j):
//     {p-limb rows 0..2 | b-limb rows 0..2} for m = 0, then the same for m = 1      (12 u32)
// which are wave-uniform (a wave works on one z) and are fetched through the scalar cache into SGPRs,
// so the vector memory pipe carries only the database stream.  Accumulation is v_mad_u64_u32 into six
// u64 accumulators per lane, reduced every 256 terms (256 * (2^28)^2 = 2^64, include/values.h:57).
#include "common.h"
#include "kernels.h"

namespace spiral {

typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void mac6(uint64_t (&a)[6], uint32_t p0, uint32_t p1, uint32_t p2, uint32_t b0, uint32_t b1, uint32_t b2, uint64_t w) {
    const uint32_t bl = lo32(w), bh = hi32(w);
    a[0] += (uint64_t)p0 * bl;
    a[1] += (uint64_t)p1 * bl;
    a[2] += (uint64_t)p2 * bl;
    a[3] += (uint64_t)b0 * bh;
    a[4] += (uint64_t)b1 * bh;
    a[5] += (uint64_t)b2 * bh;
}
// one j step: 12-dword query record (3 x uint4) against the two database words of lane ic
__device__ __forceinline__ void mac_j(uint64_t (&a)[6], const uint4* q, uint64_t w0, uint64_t w1) {
    const uint4 qa = q[0], qb = q[1], qc = q[2];
    mac6(a, qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, w0);
    mac6(a, qb.z, qb.w, qc.x, qc.y, qc.z, qc.w, w1);
}
__device__ __forceinline__ void reduce6(uint64_t (&a)[6]) {
#pragma unroll
    for (int r = 0; r < 3; r++) {
        a[r] = mod_p(a[r]);
        a[3 + r] = mod_b(a[3 + r]);
    }
}
// acc[perm(ii)][r][c][z], ic = ii*2 + c  ->  polynomial index 6*perm(ii) + 2*r + c.  perm groups the ciphertexts by
// ii mod G (G = 2^g_log ranks of a distributed fold: rank g then owns the contiguous chunk of cts ii = g + G*k, which
// is what one reduce-scatter hands it); G = 1 is the identity.
__device__ __forceinline__ void store_acc(uint64_t* acc, const uint64_t (&a)[6], uint32_t ic, uint32_t z, uint32_t num_per, uint32_t g_log) {
    const uint32_t i0 = ic >> 1, c = ic & 1u;
    const uint32_t ii = (i0 & ((1u << g_log) - 1u)) * (num_per >> g_log) + (i0 >> g_log);
#pragma unroll
    for (uint32_t r = 0; r < 3; r++) acc[((size_t)(6u * ii + 2u * r + c)) * kN + z] = pack((uint32_t)a[r], (uint32_t)a[3 + r]);
}

// fast path: nic >= 64.  One wave per (z, block of 64 output columns), 8 database loads in flight per wave
// (tools/sweep_tune.hip).  A workgroup is kSweepZ waves on CONSECUTIVE z of the same column block: a lane's three results
// belong to three different accumulator polynomials, 16 KiB apart; the waves of a workgroup trade results through LDS and
// write full 128-byte lines instead of 1.5 M scattered 8-byte words per launch.
// What the probes in tools/sweep_tune.hip say about this kernel (config 2, one MI355X): streaming the database alone takes
// 311-316 us (7.0 TB/s); the MACs, scalar query loads and reductions add nothing to that (315 us); writing the 12.6 MB of
// results costs 22-40 us whatever their pattern and whenever they are issued (quarter of the bytes: quarter of the cost) --
// writes interleaved into a saturated HBM read stream are expensive per burst -- 25 us as 128-byte lines (this kernel,
// 338-342 us), 30-35 us as 64-byte runs or scattered words.  Fewer, fatter waves (persistent, 2 or 4 tiles per wave) lose
// far more to the lower load concurrency (400-610 us).
#ifndef SPIRAL_SWEEP_Z
#define SPIRAL_SWEEP_Z 16
#endif
constexpr uint32_t kSweepZ = SPIRAL_SWEEP_Z;  // tools/build_variants.sh can override for A/B runs
constexpr uint32_t kSweepRow = 64 * 3 + 1;  // packed results per z in LDS, +1 word of padding against bank conflicts
__global__ __launch_bounds__(kSweepZ * 64) void sweep_kernel(const uint64_t* __restrict__ db, const uint32_t* __restrict__ qs, uint64_t* __restrict__ acc,
                                                             uint32_t nic, uint32_t dim0, uint32_t g_log) {
    __shared__ uint64_t sh[kSweepZ * kSweepRow];
    const uint32_t lane = threadIdx.x & 63u, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t wpz = nic >> 6;  // column blocks (= waves) per z
    const uint32_t zg = blockIdx.x / wpz, icb = blockIdx.x - zg * wpz;
    const uint32_t z = zg * kSweepZ + wv, tile = z * wpz + icb;
    const u64x2* dbp = reinterpret_cast<const u64x2*>(db) + (size_t)tile * dim0 * 64u + lane;  // block (z, icb)
    const uint4* q = reinterpret_cast<const uint4*>(qs) + (size_t)z * dim0 * 3u;  // wave-uniform, 3 x uint4 per j
    uint64_t a[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t j0 = 0; j0 < dim0; j0 += 128) {  // 128 j = 256 terms per accumulator between reductions
        const uint32_t jend = min(j0 + 128u, dim0);
#pragma unroll 8
        for (uint32_t j = j0; j < jend; j++) {
            const u64x2 w = __builtin_nontemporal_load(dbp + (size_t)j * 64u);
            mac_j(a, q + j * 3u, w.x, w.y);
        }
        reduce6(a);
    }
#pragma unroll
    for (uint32_t r = 0; r < 3; r++) sh[wv * kSweepRow + lane * 3u + r] = pack((uint32_t)a[r], (uint32_t)a[3 + r]);
    __syncthreads();
    // 192 (column, row) results x kSweepZ consecutive z: thread -> (result, z) with z fastest.
    // acc[perm(ii)][r][c][z], ic = ii*2 + c -> polynomial 6*perm(ii) + 2*r + c; perm groups the ciphertexts by ii mod G
    // (G = 2^g_log ranks of a distributed fold: rank g then owns the contiguous chunk ii = g + G*k, which is what one
    // reduce-scatter hands it); G = 1 is the identity.
    const uint32_t num_per = nic >> 1;
#pragma unroll
    for (uint32_t m = 0; m < 3; m++) {
        const uint32_t idx = threadIdx.x + kSweepZ * 64u * m, res = idx / kSweepZ, zz = idx - res * kSweepZ;
        const uint32_t col = res / 3u, r = res - col * 3u, ic = icb * 64u + col, i0 = ic >> 1, c = ic & 1u;
        const uint32_t ii = (i0 & ((1u << g_log) - 1u)) * (num_per >> g_log) + (i0 >> g_log);
        acc[((size_t)(6u * ii + 2u * r + c)) * kN + zg * kSweepZ + zz] = sh[zz * kSweepRow + res];
    }
}

// small-geometry path (nic < 64, test sizes only): one thread per (z, ic), no wave-uniform query
__global__ __launch_bounds__(256) void sweep_small_kernel(const uint64_t* __restrict__ db, const uint32_t* __restrict__ qs,
                                                          uint64_t* __restrict__ acc, uint32_t nic, uint32_t dim0, uint32_t g_log) {
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    const uint32_t z = g / nic, ic = g - z * nic;
    if (z >= kN) return;
    const ulonglong2* dbp = reinterpret_cast<const ulonglong2*>(db + db_word_index(z, 0, ic, 0, nic, dim0));
    const uint4* q = reinterpret_cast<const uint4*>(qs) + (size_t)z * dim0 * 3u;
    uint64_t a[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t j = 0; j < dim0; j++) {
        const ulonglong2 w = dbp[(size_t)j * nic];  // nic < 64: one block per z, block width = nic
        mac_j(a, q + j * 3u, w.x, w.y);
        if ((j & 127u) == 127u) reduce6(a);
    }
    reduce6(a);
    store_acc(acc, a, ic, z, nic >> 1, g_log);
}

void launch_sweep(const uint64_t* db, const uint32_t* qs, uint64_t* acc, uint32_t num_per, uint32_t jm_total, uint32_t g_log, hipStream_t s) {
    const uint32_t nic = 2 * num_per, dim0 = jm_total / 2;
    if (dim0 == 0) return;
    if (nic >= 64) {
        hipLaunchKernelGGL(sweep_kernel, dim3((kN / kSweepZ) * (nic >> 6)), dim3(kSweepZ * 64), 0, s, db, qs, acc, nic, dim0, g_log);
    } else {
        const uint32_t threads = kN * nic;
        hipLaunchKernelGGL(sweep_small_kernel, dim3((threads + 255) / 256), dim3(256), 0, s, db, qs, acc, nic, dim0, g_log);
    }
}

// reference layout (src/spiral.cpp:1139-1153): z*(num_per*2*dim0*2) + ii*(2*dim0*2) + c*(dim0*2) + j*2 + m
__global__ __launch_bounds__(256) void db_relayout_kernel(const uint64_t* __restrict__ ref, uint64_t* __restrict__ dev, uint32_t num_per, uint32_t dim0,
                                                          uint32_t j0, uint32_t dim0_shard, uint32_t nz) {
    const uint32_t nic = 2 * num_per;
    const size_t o = (size_t)blockIdx.x * 256u + threadIdx.x;  // output word index
    const size_t per_z = (size_t)dim0_shard * nic * 2u;
    const uint32_t z = (uint32_t)(o / per_z);
    if (z >= nz) return;
    size_t rem = o - (size_t)z * per_z;
    const uint32_t m = (uint32_t)(rem & 1u);
    rem >>= 1;
    const uint32_t ic = (uint32_t)(rem % nic), jl = (uint32_t)(rem / nic);
    const uint32_t ii = ic >> 1, c = ic & 1u, j = j0 + jl;
    dev[db_word_index(pk_pos(z), jl, ic, m, nic, dim0_shard)] = ref[(size_t)z * ((size_t)num_per * 2u * dim0 * 2u) + (size_t)ii * (2u * dim0 * 2u) + (size_t)c * (dim0 * 2u) + (size_t)j * 2u + m];
}
void launch_db_relayout(const uint64_t* db_ref, uint64_t* db_dev, uint32_t num_per, uint32_t dim0, uint32_t j0, uint32_t dim0_shard, uint32_t nz,
                        hipStream_t s) {
    const size_t words = (size_t)nz * dim0_shard * 2u * num_per * 2u;
    hipLaunchKernelGGL(db_relayout_kernel, dim3((uint32_t)((words + 255) / 256)), dim3(256), 0, s, db_ref, db_dev, num_per, dim0, j0, dim0_shard, nz);
}

// reference reorientCiphertexts layout (src/spiral.cpp:410-433): z*(dim0*2*4) + j*8 + m*4 + r
__global__ __launch_bounds__(256) void qs_from_reoriented_kernel(const uint64_t* __restrict__ re, uint32_t* __restrict__ qs, uint32_t jm_total) {
    const size_t g = (size_t)blockIdx.x * 256u + threadIdx.x;  // (z, jm), z in the reference's slot order
    if (g >= (size_t)kN * jm_total) return;
    const uint64_t* src = re + g * 4u;
    const uint32_t z = (uint32_t)(g / jm_total), jm = (uint32_t)(g - (size_t)z * jm_total);
    uint32_t* rec = qs + ((size_t)pk_pos(z) * jm_total + jm) * 6u;  // (z, j) record = 12 u32, m selects the half
#pragma unroll
    for (uint32_t r = 0; r < 3; r++) {
        rec[r] = lo32(src[r]) % kP;
        rec[3 + r] = hi32(src[r]) % kB;
    }
}
void launch_qs_from_reoriented(const uint64_t* reoriented, uint32_t* qs, uint32_t jm_total, hipStream_t s) {
    const size_t n = (size_t)kN * jm_total;
    hipLaunchKernelGGL(qs_from_reoriented_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, reoriented, qs, jm_total);
}

__global__ __launch_bounds__(256) void fill_db_random_kernel(uint64_t* db, uint64_t nwords, uint64_t seed) {
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < nwords; i += stride) {
        uint64_t x = seed + i + 0x9E3779B97F4A7C15ull;
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
        x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
        x ^= x >> 31;
        db[i] = pack((uint32_t)(x & 0xffffffffull) % kP, (uint32_t)(x >> 32) % kB);
    }
}
void launch_fill_db_random(uint64_t* db_dev, uint64_t nwords, uint64_t seed, hipStream_t s) {
    hipLaunchKernelGGL(fill_db_random_kernel, dim3(2048), dim3(256), 0, s, db_dev, nwords, seed);
}

}  // namespace spiral
