// First-dimension sweep: the HBM-bound kernel of the server path (replaces multiplyQueryByDatabase,
// reference src/spiral.cpp:628-999).
//
//   acc[ii][r][c].limb(z) = ( sum_{j,m} ct_j[r][m].limb(z) * DB[ii, j][m][c].limb(z) ) mod m_limb
//
// Per NTT slot z this is a (nic x JM) by (JM x 3) product per limb with nic = 2*num_per output columns
// (ii, c) and JM = 2*dim0 terms (j, m): 6 integer MADs per 8-byte database word, so the kernel is bound
// by streaming the database once from HBM and the vector ALU keeps up (6 x 32x32->64-bit MADs per 7 bytes).  Batches of queries against one pass
// over the database run on the matrix cores instead: sweep_mfma.hip.
//
// Device database layout (built at load time, any re-layout is internal; common.h): 64-lane tiles -- (z, block of 64
// columns), or 64/nic slots z x nic columns when there are fewer than 64 columns -- each one sequential stream; a word is
// stored in 7 bytes (two 28-bit residues), 8 j of a lane = 112 bytes = 7 x 16-byte loads, so one wave reads 64 lanes x
// 16 B = 1 KiB per instruction and 7/8 of the reference's database bytes per query.
// The query is stored as one 48-byte record per (z, j):
//     {p-limb rows 0..2 | b-limb rows 0..2} for m = 0, then the same for m = 1      (12 u32)
// With >= 64 columns they are wave-uniform (a wave works on one z) and are fetched through the scalar cache into SGPRs,
// so the vector memory pipe carries only the database stream; narrower geometries stage them in LDS per wave.  Accumulation is v_mad_u64_u32 into six
// u64 accumulators per lane, reduced every 256 terms (256 * (2^28)^2 = 2^64, include/values.h:57).
#include <cstdlib>
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
// Position of ciphertext i0 = g + G k in the accumulator buffer: [stage s][rank g][k' < Ls] with k = s Ls + k', Ls = 2^ls_log.
// One stage (ls_log = log2(num_per / G)) is the plain grouping by rank; with K = 2^k_log stages (ls_log smaller by k_log) the
// ciphertexts of stage s -- the contiguous columns [s num_per/K, (s+1) num_per/K) -- form one contiguous [G][Ls] block, which one
// reduce-scatter per stage turns into rank g's rows k of that stage (the sweep of stage s + 1 runs under it, server.cpp).
__device__ __forceinline__ void store_acc(uint64_t* acc, const uint64_t (&a)[6], uint32_t ic, uint32_t z, uint32_t g_log, uint32_t ls_log) {
    const uint32_t i0 = ic >> 1, c = ic & 1u;
    const uint32_t ii = acc_pos(i0, g_log, ls_log);
#pragma unroll
    for (uint32_t r = 0; r < 3; r++) acc[((size_t)(6u * ii + 2u * r + c)) * kN + z] = pack((uint32_t)a[r], (uint32_t)a[3 + r]);
}

// ---- fast path: packed database (dim0 % 8 == 0) ----------------------------------------------------------------------------
// MODE 0: one wave per (z, block of 64 output columns).  A workgroup is kSweepZ waves on CONSECUTIVE z of the same column block: a
// lane's three results belong to three different accumulator polynomials, 16 KiB apart; the waves of a workgroup trade
// results through LDS and write full 128-byte lines instead of 1.5 M scattered 8-byte words per launch.
// What the probes in tools/sweep_tune.hip say (config 2, one MI355X): streaming an 8-byte-per-word database alone takes
// 311-316 us (7.0 TB/s); the MACs, scalar query loads and reductions add nothing to that; writing the 12.6 MB of results
// costs 22-40 us whatever their pattern and whenever they are issued (writes interleaved into a saturated HBM read stream
// are expensive per burst), 25 us as 128-byte lines; fewer, fatter waves (persistent, 2 or 4 tiles per wave) lose far more
// to the lower load concurrency (400-610 us).  Packing the words in 7 bytes removes an eighth of the stream for ~2 extra
// VALU instructions per residue (funnel shift + mask), which the memory-bound loop absorbs: 343 -> 295 us.
#ifndef SPIRAL_SWEEP_Z
#define SPIRAL_SWEEP_Z 16
#endif
constexpr uint32_t kSweepZ = SPIRAL_SWEEP_Z;  // tools/build_variants.sh can override for A/B runs
constexpr uint32_t kSweepRow = 64 * 3 + 1;    // packed results per z in LDS, +1 word of padding against bank conflicts
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// j = JJ of the group: fields 4JJ .. 4JJ+3 = (p, b) of m = 0, (p, b) of m = 1, against the 12-dword query record of that j
template <int JJ>
__device__ __forceinline__ void mac_packed_j(uint64_t (&a)[6], const uint4* q, const uint32_t (&d)[28]) {
    const uint4 qa = q[0], qb = q[1], qc = q[2];
    const uint32_t p0 = field28<4 * JJ>(d), b0 = field28<4 * JJ + 1>(d), p1 = field28<4 * JJ + 2>(d), b1 = field28<4 * JJ + 3>(d);
    a[0] += (uint64_t)qa.x * p0;
    a[1] += (uint64_t)qa.y * p0;
    a[2] += (uint64_t)qa.z * p0;
    a[3] += (uint64_t)qa.w * b0;
    a[4] += (uint64_t)qb.x * b0;
    a[5] += (uint64_t)qb.y * b0;
    a[0] += (uint64_t)qb.z * p1;
    a[1] += (uint64_t)qb.w * p1;
    a[2] += (uint64_t)qc.x * p1;
    a[3] += (uint64_t)qc.y * b1;
    a[4] += (uint64_t)qc.z * b1;
    a[5] += (uint64_t)qc.w * b1;
}
// MODE 0 (nic >= 64): a wave is one slot z and 64 columns, the query records are wave-uniform (SGPR operands).
// MODE 1, 2 (nic = W < 64): a wave is P = 64/W consecutive slots x W columns, lane = (z % P) * W + column, and every lane
// needs the records of its own z.  MODE 1: each lane loads them through the vector pipe (the W lanes of a slot read the
// same address).  MODE 2 (P <= 8): the wave stages the P x 8 records of a group in LDS with two coalesced loads and the
// lanes read them from there, which takes three quarters of the requests off the vector memory pipe.
constexpr uint32_t kQStage = 8 * 24;  // uint4 per wave: up to 8 slots x (8 j x 3)
// NB > 1 (MODE 0 only): NB queries against ONE pass over the database -- every 112-byte group a lane fetches and unpacks is
// multiplied into NB accumulator sets, each against its own query's records (all wave-uniform, scalar loads).  The sweep has the
// VALU headroom (6 MADs per 7 bytes at 6.4 TB/s = 5.5 T MAD/s of ~31): for throughput at the database sizes where the sweep is
// most of a query, the stream is paid once per NB queries (spiral_gpu_server_first_dim_batch).  Single-query latency is NB = 1.
// NB = 2 is the fallback where the matrix-core sweep (sweep_mfma.hip: up to eight queries per pass at the cost of the stream) does not apply; with
// three and four queries per pass this kernel became VALU / LDS bound (350 / 440 us at config 2: round 5's LDS-record variant, removed).
struct SweepBatch {
    const uint32_t* qs[kSweepMaxBatch];
    uint64_t* acc[kSweepMaxBatch];
};
template <int MODE, int NB = 1, uint32_t Z = kSweepZ>
__global__ __launch_bounds__(Z * 64) void sweep_kernel(const uint64_t* __restrict__ db, SweepBatch bt, uint32_t nic, uint32_t dim0, uint32_t g_log, uint32_t ls_log,
                                                             uint32_t icb0, uint32_t n_icb) {
    static_assert(NB == 1 || MODE == 0, "batched sweeps use the wide geometry");
    const uint32_t* __restrict__ qs = bt.qs[0];
    uint64_t* __restrict__ acc = bt.acc[0];
    constexpr bool WIDE = MODE == 0;
    constexpr uint32_t kShWords = MODE == 2 ? (Z * kQStage * 2 > Z * kSweepRow ? Z * kQStage * 2 : Z * kSweepRow) : Z * kSweepRow;
    __shared__ __attribute__((aligned(16))) uint64_t sh[kShWords];  // results; MODE 2: first the record staging (aliased)
    const uint32_t lane = threadIdx.x & 63u, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t w = WIDE ? 64u : nic, pz = 64u / w, nblk = nic / w;  // columns and slots per tile, column blocks per slot group
    // Workgroups go to the 8 XCDs round-robin by blockIdx, and each XCD has its own L2.  The nblk workgroups that share a
    // z-group read the same query records, so consecutive WORK items (not consecutive block ids) are given to one XCD:
    // block b = 8q + x takes work x * (nblocks/8) + q.  (FETCH_SIZE showed the records being fetched once per XCD that
    // touched them: 4 x 25 MB instead of 25 MB at config 2.)
    uint32_t work = blockIdx.x;
    if ((gridDim.x & 7u) == 0) work = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    // (a launch may cover only the column blocks [icb0, icb0 + n_icb) of the wide geometry: one stage of a pipelined sweep)
    const uint32_t zg = work / n_icb, icb = icb0 + (work - zg * n_icb);  // WIDE: zg = group of Z slots; else one tile of pz slots
    const uint32_t groups = dim0 >> 3;
    // WIDE: the workgroup's waves take Z consecutive tiles whole.  !WIDE: there are only N/pz tiles, each a long
    // stream, so the waves of a workgroup split ONE tile's j range between them (load concurrency is what buys bandwidth)
    // and their partial sums are added through LDS.
    const uint32_t ztile = WIDE ? zg * Z + wv : zg, tile = ztile * nblk + icb;
    const uint32_t gper = WIDE ? groups : (groups + Z - 1u) / Z;
    const uint32_t gfirst = WIDE ? 0u : min(wv * gper, groups), glast = WIDE ? groups : min(gfirst + gper, groups);
    const uint32_t z = WIDE ? ztile : ztile * pz + lane / w;  // this lane's slot
    const u32x4* dbp = reinterpret_cast<const u32x4*>(db) + (size_t)tile * groups * 7u * 64u + lane;
    const uint4* q = reinterpret_cast<const uint4*>(qs) + (size_t)z * dim0 * 3u;  // 3 x uint4 per j; wave-uniform when WIDE
    uint64_t a[6] = {0, 0, 0, 0, 0, 0};
    uint64_t ax[NB > 1 ? NB - 1 : 1][6] = {};  // queries 1 .. NB-1 of a batch
    for (uint32_t g0 = gfirst; g0 < glast; g0 += 16) {  // 16 groups = 128 j = 256 terms per accumulator between reductions
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
            const uint4* qg = q + (size_t)g * 24u;
            if constexpr (MODE == 2) {
                uint4* qst = reinterpret_cast<uint4*>(sh) + wv * kQStage;
                const uint4* qsrc = reinterpret_cast<const uint4*>(qs);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();  // the previous group's reads of the staging area
                for (uint32_t t = lane; t < pz * 24u; t += 64u) {
                    const uint32_t zi = t / 24u, off = t - zi * 24u;
                    qst[t] = qsrc[((size_t)(ztile * pz + zi) * dim0 + (size_t)g * 8u) * 3u + off];
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                qg = qst + (lane / w) * 24u;
            }
            mac_packed_j<0>(a, qg, d);
            mac_packed_j<1>(a, qg + 3, d);
            mac_packed_j<2>(a, qg + 6, d);
            mac_packed_j<3>(a, qg + 9, d);
            mac_packed_j<4>(a, qg + 12, d);
            mac_packed_j<5>(a, qg + 15, d);
            mac_packed_j<6>(a, qg + 18, d);
            mac_packed_j<7>(a, qg + 21, d);
            if constexpr (NB > 1) {
#pragma unroll
                for (int b = 1; b < NB; b++) {
                    const uint4* qb = reinterpret_cast<const uint4*>(bt.qs[b]) + (size_t)z * dim0 * 3u + (size_t)g * 24u;
                    mac_packed_j<0>(ax[b - 1], qb, d);
                    mac_packed_j<1>(ax[b - 1], qb + 3, d);
                    mac_packed_j<2>(ax[b - 1], qb + 6, d);
                    mac_packed_j<3>(ax[b - 1], qb + 9, d);
                    mac_packed_j<4>(ax[b - 1], qb + 12, d);
                    mac_packed_j<5>(ax[b - 1], qb + 15, d);
                    mac_packed_j<6>(ax[b - 1], qb + 18, d);
                    mac_packed_j<7>(ax[b - 1], qb + 21, d);
                }
            }
        }
        reduce6(a);
        if constexpr (NB > 1) {
#pragma unroll
            for (int b = 1; b < NB; b++) reduce6(ax[b - 1]);
        }
    }
    if constexpr (MODE == 2) __syncthreads();  // every wave is done with its staging area before results overwrite it
#pragma unroll
    for (int b = 0; b < NB; b++) {  // one query's results at a time through the same LDS rows
    if constexpr (NB > 1) {
        if (b > 0) {
            __syncthreads();  // the previous query's rows have been read
            acc = bt.acc[b];
#pragma unroll
            for (int r = 0; r < 6; r++) a[r] = ax[b - 1][r];
        }
    }
#pragma unroll
    for (uint32_t r = 0; r < 3; r++) sh[wv * kSweepRow + lane * 3u + r] = pack((uint32_t)a[r], (uint32_t)a[3 + r]);
    __syncthreads();
    // w*3 (column, row) results x Z*pz consecutive z: thread -> (result, z) with z fastest.
    // acc[perm(ii)][r][c][z], ic = ii*2 + c -> polynomial 6*perm(ii) + 2*r + c; perm groups the ciphertexts by ii mod G
    // (G = 2^g_log ranks of a distributed fold: rank g then owns the contiguous chunk ii = g + G*k, which is what one
    // reduce-scatter hands it); G = 1 is the identity.
    if constexpr (WIDE) {
#pragma unroll
        for (uint32_t m = 0; m < 3; m++) {
            const uint32_t idx = threadIdx.x + Z * 64u * m, res = idx / Z, zz = idx - res * Z;
            const uint32_t col = res / 3u, r = res - col * 3u, ic = icb * 64u + col, i0 = ic >> 1, c = ic & 1u;
            const uint32_t ii = acc_pos(i0, g_log, ls_log);
            acc[((size_t)(6u * ii + 2u * r + c)) * kN + zg * Z + zz] = sh[zz * kSweepRow + res];
        }
    } else if (threadIdx.x < 192u) {  // 64 lanes x 3 results of this tile, each the sum of the Z waves' partials (< 16 * 2^28)
        const uint32_t sl = threadIdx.x / 3u, r = threadIdx.x - sl * 3u, zz = sl / w, col = sl - zz * w, i0 = col >> 1, c = col & 1u;
        uint64_t sp = 0, sb = 0;
#pragma unroll
        for (uint32_t v = 0; v < Z; v++) {
            const uint64_t x = sh[v * kSweepRow + threadIdx.x];
            sp += lo32(x);
            sb += hi32(x);
        }
        const uint32_t ii = acc_pos(i0, g_log, ls_log);
        acc[((size_t)(6u * ii + 2u * r + c)) * kN + zg * pz + zz] = pack(mod_p(sp), mod_b(sb));
    }
    }  // batch
}

// plain-layout path (dim0 < 8, test sizes only): one thread per (z, ic)
__global__ __launch_bounds__(256) void sweep_small_kernel(const uint64_t* __restrict__ db, const uint32_t* __restrict__ qs,
                                                          uint64_t* __restrict__ acc, uint32_t nic, uint32_t dim0, uint32_t g_log, uint32_t ls_log) {
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    const uint32_t z = g / nic, ic = g - z * nic;
    if (z >= kN) return;
    const uint4* q = reinterpret_cast<const uint4*>(qs) + (size_t)z * dim0 * 3u;
    uint64_t a[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t j = 0; j < dim0; j++) {
        const ulonglong2 w = *reinterpret_cast<const ulonglong2*>(db + db_word_index(z, j, ic, 0, nic, dim0));
        mac_j(a, q + j * 3u, w.x, w.y);
        if ((j & 127u) == 127u) reduce6(a);
    }
    reduce6(a);
    store_acc(acc, a, ic, z, g_log, ls_log);
}

static uint32_t log2u(uint32_t x) {
    uint32_t l = 0;
    while ((1u << l) < x) l++;
    return l;
}
bool sweep_batch_ok(uint32_t num_per, uint32_t jm_total) { return 2 * num_per >= 64 && db_packed(2 * num_per, jm_total / 2); }
void launch_sweep_batch(const uint64_t* db, const uint32_t* const* qs, uint64_t* const* acc, uint32_t n, uint32_t num_per, uint32_t jm_total, uint32_t g_log,
                        hipStream_t s) {
    const uint32_t nic = 2 * num_per, dim0 = jm_total / 2, ls_log = log2u(num_per) - g_log;
    SweepBatch bt{};
    for (uint32_t b = 0; b < n; b++) {
        bt.qs[b] = qs[b];
        bt.acc[b] = acc[b];
    }
    const dim3 grid((kN / kSweepZ) * (nic >> 6)), block(kSweepZ * 64);
    switch (n) {
        case 2: hipLaunchKernelGGL((sweep_kernel<0, 2>), grid, block, 0, s, db, bt, nic, dim0, g_log, ls_log, 0u, nic >> 6); break;
        default: abort();
    }
}
bool sweep_stages_ok(uint32_t num_per, uint32_t jm_total, uint32_t g_log, uint32_t k_log) {
    if (k_log == 0) return true;
    const uint32_t nic = 2 * num_per;  // whole 64-column blocks per stage, at least one ciphertext per rank and stage
    return (num_per & (num_per - 1)) == 0 && db_packed(nic, jm_total / 2) && (nic >> 6) >= (1u << k_log) && log2u(num_per) >= g_log + k_log;
}
void launch_sweep(const uint64_t* db, const uint32_t* qs, uint64_t* acc, uint32_t num_per, uint32_t jm_total, uint32_t g_log, hipStream_t s, uint32_t k_log,
                  int stage) {
    const uint32_t nic = 2 * num_per, dim0 = jm_total / 2;
    if (dim0 == 0) return;
    const uint32_t ls_log = log2u(num_per) - g_log - k_log;
    SweepBatch bt{};
    bt.qs[0] = qs;
    bt.acc[0] = acc;
    if (db_packed(nic, dim0)) {
        static const bool stage_recs = [] {
            const char* e = tuning_env("SPIRAL_SWEEP_STAGE");  // tuning only: 0 = narrow geometries load their records per lane
            return e ? atoi(e) != 0 : true;
        }();
        if (nic >= 64) {
            const uint32_t nblk = nic >> 6, per = nblk >> k_log;  // column blocks per stage
            const uint32_t icb0 = stage < 0 ? 0u : (uint32_t)stage * per, n_icb = stage < 0 ? nblk : per;
            // 64 columns (nu2 = 5; one stage of a pipelined sweep) are only 128 workgroups of 16 slots, half the chip: 8 slots per workgroup there
            // (177 -> 146 us = 6.4 -> 7.8 TB/s at nu1 = 9, nu2 = 5; no difference from 128 columns up)
            if ((kN / kSweepZ) * n_icb < 256u)
                hipLaunchKernelGGL((sweep_kernel<0, 1, 8>), dim3((kN / 8) * n_icb), dim3(8 * 64), 0, s, db, bt, nic, dim0, g_log, ls_log, icb0, n_icb);
            else
                hipLaunchKernelGGL(sweep_kernel<0>, dim3((kN / kSweepZ) * n_icb), dim3(kSweepZ * 64), 0, s, db, bt, nic, dim0, g_log, ls_log, icb0, n_icb);
        } else if (nic >= 8 && stage_recs)  // one workgroup per tile of 64/nic <= 8 slots, its waves split the j range; records staged in LDS
            hipLaunchKernelGGL(sweep_kernel<2>, dim3(kN / (64 / nic)), dim3(kSweepZ * 64), 0, s, db, bt, nic, dim0, g_log, ls_log, 0u, 1u);
        else
            hipLaunchKernelGGL(sweep_kernel<1>, dim3(kN / (64 / nic)), dim3(kSweepZ * 64), 0, s, db, bt, nic, dim0, g_log, ls_log, 0u, 1u);
    } else {
        const uint32_t threads = kN * nic;
        hipLaunchKernelGGL(sweep_small_kernel, dim3((threads + 255) / 256), dim3(256), 0, s, db, qs, acc, nic, dim0, g_log, ls_log);
    }
}

// reference layout (src/spiral.cpp:1139-1153): z*(num_per*2*dim0*2) + ii*(2*dim0*2) + c*(dim0*2) + j*2 + m.
// `ref` holds the nz slabs z0 .. z0+nz-1 (reference slot order); `dev` is the base of the shard's device database.
__global__ __launch_bounds__(256) void db_relayout_kernel(const uint64_t* __restrict__ ref, uint64_t* __restrict__ dev, uint32_t num_per, uint32_t dim0,
                                                          uint32_t j0, uint32_t dim0_shard, uint32_t z0, uint32_t nz) {
    const uint32_t nic = 2 * num_per;
    const size_t o = (size_t)blockIdx.x * 256u + threadIdx.x;  // word of the shard, slab-major
    const size_t per_z = (size_t)dim0_shard * nic * 2u;
    const uint32_t zl = (uint32_t)(o / per_z);
    if (zl >= nz) return;
    size_t rem = o - (size_t)zl * per_z;
    const uint32_t m = (uint32_t)(rem & 1u);
    rem >>= 1;
    const uint32_t ic = (uint32_t)(rem % nic), jl = (uint32_t)(rem / nic);
    const uint32_t ii = ic >> 1, c = ic & 1u, j = j0 + jl;
    const uint64_t v = ref[(size_t)zl * ((size_t)num_per * 2u * dim0 * 2u) + (size_t)ii * (2u * dim0 * 2u) + (size_t)c * (dim0 * 2u) + (size_t)j * 2u + m];
    db_put_word(dev, pk_pos(z0 + zl), jl, ic, m, nic, dim0_shard, pack(lo32(v) % kP, hi32(v) % kB));
}
void launch_db_relayout(const uint64_t* db_ref, uint64_t* db_dev, uint32_t num_per, uint32_t dim0, uint32_t j0, uint32_t dim0_shard, uint32_t z0,
                        uint32_t nz, hipStream_t s) {
    const size_t words = (size_t)nz * dim0_shard * 2u * num_per * 2u;
    hipLaunchKernelGGL(db_relayout_kernel, dim3((uint32_t)((words + 255) / 256)), dim3(256), 0, s, db_ref, db_dev, num_per, dim0, j0, dim0_shard, z0, nz);
}

// ---- database read-back (tests / inspection): the inverse maps of the two loaders above ----------------------------------
__global__ __launch_bounds__(256) void db_read_item_kernel(const uint64_t* __restrict__ dev, uint64_t* __restrict__ out, uint32_t num_per, uint32_t dim0_shard,
                                                           uint32_t jl, uint32_t ii, uint32_t limbs) {
    const uint32_t z = blockIdx.x * 256u + threadIdx.x, mc = blockIdx.y, m = mc >> 1, c = mc & 1u;  // polynomial (m, c) of the n0 x n2 plaintext
    const uint64_t v = limbs ? db_get_word_limbs(dev, pk_pos(z), jl, ii * 2u + c, m, 2u * num_per, dim0_shard)
                             : db_get_word(dev, pk_pos(z), jl, ii * 2u + c, m, 2u * num_per, dim0_shard);
    out[(size_t)mc * (2 * kN) + z] = lo32(v);
    out[(size_t)mc * (2 * kN) + kN + z] = hi32(v);
}
void launch_db_read_item(const uint64_t* db_dev, uint64_t* out_ref, uint32_t num_per, uint32_t dim0_shard, uint32_t j_local, uint32_t ii, hipStream_t s,
                         bool limbs) {
    hipLaunchKernelGGL(db_read_item_kernel, dim3(kN / 256, 4), dim3(256), 0, s, db_dev, out_ref, num_per, dim0_shard, j_local, ii, limbs ? 1u : 0u);
}
__global__ __launch_bounds__(256) void db_read_slots_kernel(const uint64_t* __restrict__ dev, uint64_t* __restrict__ out, uint32_t num_per, uint32_t dim0_shard,
                                                            uint32_t z0, uint32_t nz, uint32_t ii0, uint32_t n_ii, uint32_t limbs) {
    const size_t o = (size_t)blockIdx.x * 256u + threadIdx.x, per_z = (size_t)n_ii * 2u * dim0_shard * 2u;
    const uint32_t zl = (uint32_t)(o / per_z);
    if (zl >= nz) return;
    size_t rem = o - (size_t)zl * per_z;  // (ii - ii0)*(2*dim0*2) + c*(dim0*2) + j*2 + m
    const uint32_t m = (uint32_t)(rem & 1u);
    rem >>= 1;
    const uint32_t jl = (uint32_t)(rem % dim0_shard);
    rem /= dim0_shard;
    const uint32_t c = (uint32_t)(rem & 1u), ii = ii0 + (uint32_t)(rem >> 1);
    out[o] = limbs ? db_get_word_limbs(dev, pk_pos(z0 + zl), jl, ii * 2u + c, m, 2u * num_per, dim0_shard)
                   : db_get_word(dev, pk_pos(z0 + zl), jl, ii * 2u + c, m, 2u * num_per, dim0_shard);
}
void launch_db_read_slots(const uint64_t* db_dev, uint64_t* out, uint32_t num_per, uint32_t dim0_shard, uint32_t z0, uint32_t nz, uint32_t ii0, uint32_t n_ii,
                          hipStream_t s, bool limbs) {
    const size_t words = (size_t)nz * n_ii * 2u * dim0_shard * 2u;
    hipLaunchKernelGGL(db_read_slots_kernel, dim3((uint32_t)((words + 255) / 256)), dim3(256), 0, s, db_dev, out, num_per, dim0_shard, z0, nz, ii0, n_ii,
                       limbs ? 1u : 0u);
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

// arbitrary valid words (benchmarks): word number i of the shard, any order
__global__ __launch_bounds__(256) void fill_db_random_kernel(uint64_t* db, uint32_t nic, uint32_t dim0, uint64_t seed) {
    const uint64_t nwords = (uint64_t)kN * dim0 * nic * 2u, stride = (uint64_t)gridDim.x * 256u;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < nwords; i += stride) {
        uint64_t x = seed + i + 0x9E3779B97F4A7C15ull;
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
        x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
        x ^= x >> 31;
        const uint32_t m = (uint32_t)(i & 1u), ic = (uint32_t)((i >> 1) % nic);
        const uint64_t rest = (i >> 1) / nic;
        const uint32_t j = (uint32_t)(rest % dim0), z = (uint32_t)(rest / dim0);
        db_put_word(db, z, j, ic, m, nic, dim0, pack((uint32_t)(x & 0xffffffffull) % kP, (uint32_t)(x >> 32) % kB));
    }
}
void launch_fill_db_random(uint64_t* db_dev, uint32_t num_per, uint32_t dim0_shard, uint64_t seed, hipStream_t s) {
    hipLaunchKernelGGL(fill_db_random_kernel, dim3(4096), dim3(256), 0, s, db_dev, 2 * num_per, dim0_shard, seed);
}

}  // namespace spiral
