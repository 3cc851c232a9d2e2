// Tuning harness for the first-dimension sweep: variants of sweep_kernel on a synthetic 2 GiB database, timed
// interleaved with HIP events; every variant is checked against variant 0.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I spiral_amd/csrc tools/sweep_tune.hip -o tools/sweep_tune
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "common.h"
using namespace spiral;

typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void mac6(uint64_t (&a)[6], uint32_t p0, uint32_t p1, uint32_t p2, uint32_t b0, uint32_t b1, uint32_t b2, uint64_t w) {
    const uint32_t bl = lo32(w), bh = hi32(w);
    a[0] += (uint64_t)p0 * bl; a[1] += (uint64_t)p1 * bl; a[2] += (uint64_t)p2 * bl;
    a[3] += (uint64_t)b0 * bh; a[4] += (uint64_t)b1 * bh; a[5] += (uint64_t)b2 * bh;
}
__device__ __forceinline__ void mac_j(uint64_t (&a)[6], const uint4* q, uint64_t w0, uint64_t w1) {
    const uint4 qa = q[0], qb = q[1], qc = q[2];
    mac6(a, qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, w0);
    mac6(a, qb.z, qb.w, qc.x, qc.y, qc.z, qc.w, w1);
}
__device__ __forceinline__ void reduce6(uint64_t (&a)[6]) {
#pragma unroll
    for (int r = 0; r < 3; r++) { a[r] = mod_p(a[r]); a[3 + r] = mod_b(a[3 + r]); }
}
__device__ __forceinline__ void store_acc(uint64_t* acc, const uint64_t (&a)[6], uint32_t ic, uint32_t z) {
    const uint32_t ii = ic >> 1, c = ic & 1u;
#pragma unroll
    for (uint32_t r = 0; r < 3; r++) acc[((size_t)(6u * ii + 2u * r + c)) * kN + z] = pack((uint32_t)a[r], (uint32_t)a[3 + r]);
}

// one tile = (z, block of 64 columns); tiles numbered z*wpz + icb like the product kernel
template <int UNROLL, bool NT>
__device__ __forceinline__ void do_tile(const uint64_t* __restrict__ db, const uint32_t* __restrict__ qs, uint64_t* __restrict__ acc, uint32_t nic,
                                        uint32_t dim0, uint32_t tile, uint32_t lane) {
    const uint32_t wpz = nic >> 6;
    const uint32_t z = tile / wpz, ic = (tile - z * wpz) * 64u + lane;
    const u64x2* dbp = reinterpret_cast<const u64x2*>(db) + (size_t)tile * dim0 * 64u + lane;
    const uint4* q = reinterpret_cast<const uint4*>(qs) + (size_t)z * dim0 * 3u;
    uint64_t a[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t j0 = 0; j0 < dim0; j0 += 128) {
        const uint32_t jend = min(j0 + 128u, dim0);
#pragma unroll UNROLL
        for (uint32_t j = j0; j < jend; j++) {
            u64x2 w;
            if constexpr (NT) w = __builtin_nontemporal_load(dbp + (size_t)j * 64u);
            else w = dbp[(size_t)j * 64u];
            mac_j(a, q + j * 3u, w.x, w.y);
        }
        reduce6(a);
    }
    store_acc(acc, a, ic, z);
}

// rotated start: wave `tile` begins its j loop at rot(tile) instead of 0, so that concurrently running waves are at
// different offsets inside their 256 KiB streams (the sum is order independent)
template <int UNROLL, int MODE>
__device__ __forceinline__ void do_tile_rot(const uint64_t* __restrict__ db, const uint32_t* __restrict__ qs, uint64_t* __restrict__ acc, uint32_t nic,
                                            uint32_t dim0, uint32_t tile, uint32_t lane) {
    const uint32_t wpz = nic >> 6;
    const uint32_t z = tile / wpz, ic = (tile - z * wpz) * 64u + lane;
    const u64x2* dbp = reinterpret_cast<const u64x2*>(db) + (size_t)tile * dim0 * 64u + lane;
    const uint4* q = reinterpret_cast<const uint4*>(qs) + (size_t)z * dim0 * 3u;
    uint32_t rot;
    if (MODE == 0) rot = (tile * 8u) & (dim0 - 1u);            // consecutive tiles 8 KiB apart in phase
    else if (MODE == 1) rot = (tile * 40u) & (dim0 - 1u) & ~7u; // 40 KiB
    else rot = ((tile * 2654435761u) >> 16) & (dim0 - 1u) & ~7u; // hashed
    uint64_t a[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t j0 = 0; j0 < dim0; j0 += 128) {
#pragma unroll UNROLL
        for (uint32_t jj = j0; jj < j0 + 128u; jj++) {
            const uint32_t j = (jj + rot) & (dim0 - 1u);
            const u64x2 w = __builtin_nontemporal_load(dbp + (size_t)j * 64u);
            mac_j(a, q + j * 3u, w.x, w.y);
        }
        reduce6(a);
    }
    store_acc(acc, a, ic, z);
}
template <int UNROLL, int MODE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void sweep_rot(const uint64_t* __restrict__ db, const uint32_t* __restrict__ qs, uint64_t* __restrict__ acc,
                                                        uint32_t nic, uint32_t dim0) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t tile = __builtin_amdgcn_readfirstlane(blockIdx.x * WAVES + (threadIdx.x >> 6));
    do_tile_rot<UNROLL, MODE>(db, qs, acc, nic, dim0, tile, lane);
}
// sensitivity probes on the product kernel's structure: ABL 1 = p-limb MACs only, 2 = query operands from registers (no scalar loads),
// 3 = both
template <int UNROLL, int ABL, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void sweep_abl(const uint64_t* __restrict__ db, const uint32_t* __restrict__ qs, uint64_t* __restrict__ acc,
                                                        uint32_t nic, uint32_t dim0) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t tile = __builtin_amdgcn_readfirstlane(blockIdx.x * WAVES + (threadIdx.x >> 6));
    const uint32_t wpz = nic >> 6;
    const uint32_t z = tile / wpz, ic = (tile - z * wpz) * 64u + lane;
    const u64x2* dbp = reinterpret_cast<const u64x2*>(db) + (size_t)tile * dim0 * 64u + lane;
    const uint4* q = reinterpret_cast<const uint4*>(qs) + (size_t)z * dim0 * 3u;
    const uint4 c0 = q[0], c1 = q[1], c2 = q[2];
    uint64_t a[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t j0 = 0; j0 < dim0; j0 += 128) {
#pragma unroll UNROLL
        for (uint32_t j = j0; j < j0 + 128u; j++) {
            const u64x2 w = __builtin_nontemporal_load(dbp + (size_t)j * 64u);
            uint4 qa, qb, qc;
            if (ABL & 2) { qa = c0; qb = c1; qc = c2; } else { qa = q[j * 3u]; qb = q[j * 3u + 1]; qc = q[j * 3u + 2]; }
            if (ABL & 1) {
                a[0] += (uint64_t)qa.x * lo32(w.x); a[1] += (uint64_t)qa.y * lo32(w.x); a[2] += (uint64_t)qa.z * lo32(w.x);
                a[3] += (uint64_t)qb.z * lo32(w.y); a[4] += (uint64_t)qb.w * lo32(w.y); a[5] += (uint64_t)qc.x * lo32(w.y);
                a[0] ^= hi32(w.x) ^ hi32(w.y);
            } else {
                mac6(a, qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, w.x);
                mac6(a, qb.z, qb.w, qc.x, qc.y, qc.z, qc.w, w.y);
            }
        }
        reduce6(a);
    }
    store_acc(acc, a, ic, z);
}
// bisect between the read-only probe and the product: FEAT bit 0 = 6 MACs per 16 B with constant operands, bit 1 = 12 MACs,
// bit 2 = reduce every 128 j, bit 3 = store the result, bit 4 = xor-fold instead of MACs but into 6 accumulators
template <int UNROLL, int FEAT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void read_bisect(const uint64_t* __restrict__ db, const uint32_t* __restrict__ qs, uint64_t* __restrict__ acc,
                                                          uint32_t nic, uint32_t dim0) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t tile = __builtin_amdgcn_readfirstlane(blockIdx.x * WAVES + (threadIdx.x >> 6));
    const uint32_t wpz = nic >> 6;
    const uint32_t z = tile / wpz, ic = (tile - z * wpz) * 64u + lane;
    const u64x2* dbp = reinterpret_cast<const u64x2*>(db) + (size_t)tile * dim0 * 64u + lane;
    const uint4 c0 = reinterpret_cast<const uint4*>(qs)[z * 3u], c1 = reinterpret_cast<const uint4*>(qs)[z * 3u + 1];
    uint64_t a[6] = {0, 0, 0, 0, 0, 0};
    if (FEAT & 128) {  // the same 12.6 MB of wave-contiguous stores, but issued BEFORE the stream
#pragma unroll
        for (uint32_t r = 0; r < 3; r++) acc[((size_t)tile * 3u + r) * 64u + lane] = pack(c0.x + r, c0.y + lane);
    }
    if (FEAT & 256) {  // a quarter of the footprint at the end
        if ((tile & 3u) == 0)
#pragma unroll
            for (uint32_t r = 0; r < 3; r++) acc[((size_t)tile * 3u + r) * 64u + lane] = pack(c0.x + r, c0.y + lane);
    }
    for (uint32_t j0 = 0; j0 < dim0; j0 += 128) {
#pragma unroll UNROLL
        for (uint32_t j = j0; j < j0 + 128u; j++) {
            const u64x2 w = __builtin_nontemporal_load(dbp + (size_t)j * 64u);
            if (FEAT & 2) {
                mac6(a, c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, w.x);
                mac6(a, c1.z, c1.w, c0.x, c0.y, c0.z, c0.w, w.y);
            } else if (FEAT & 1) {
                mac6(a, c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, w.x ^ w.y);
            } else if (FEAT & 16) {
                a[0] ^= w.x; a[1] ^= w.y; a[2] += w.x; a[3] += w.y; a[4] ^= w.x >> 1; a[5] ^= w.y >> 1;
            } else {
                a[0] ^= w.x;
                a[1] ^= w.y;
            }
        }
        if (FEAT & 4) reduce6(a);
    }
    if (FEAT & 32) {  // store into a small region (every wave overwrites the same 3 x 512 B x 64 tiles)
#pragma unroll
        for (uint32_t r = 0; r < 3; r++) acc[((tile & 63u) * 3u + r) * 64u + lane] = pack((uint32_t)a[r], (uint32_t)a[3 + r]);
    } else if (FEAT & 64) {  // store wave-contiguous: 3 x 512-byte runs per wave, 12.6 MB in total, no sharing of lines between waves
#pragma unroll
        for (uint32_t r = 0; r < 3; r++) acc[((size_t)tile * 3u + r) * 64u + lane] = pack((uint32_t)a[r], (uint32_t)a[3 + r]);
    } else if (FEAT & 8) store_acc(acc, a, ic, z);
    else if ((a[0] ^ a[1] ^ a[2] ^ a[3] ^ a[4] ^ a[5]) == 0x1234567ull) acc[tile * 64 + lane] = a[0];
}
// ZW waves of a workgroup on consecutive z of one column block; results traded through LDS and written as ZW*8-byte runs
template <int UNROLL, int ZW, int NTS = 0>
__global__ __launch_bounds__(ZW * 64) void sweep_zl(const uint64_t* __restrict__ db, const uint32_t* __restrict__ qs, uint64_t* __restrict__ acc,
                                                    uint32_t nic, uint32_t dim0) {
    constexpr uint32_t ROW = 64 * 3 + 1;
    __shared__ uint64_t sh[ZW * ROW];
    const uint32_t lane = threadIdx.x & 63u, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t wpz = nic >> 6;
    const uint32_t zg = blockIdx.x / wpz, icb = blockIdx.x - zg * wpz;
    const uint32_t z = zg * ZW + wv, tile = z * wpz + icb;
    const u64x2* dbp = reinterpret_cast<const u64x2*>(db) + (size_t)tile * dim0 * 64u + lane;
    const uint4* q = reinterpret_cast<const uint4*>(qs) + (size_t)z * dim0 * 3u;
    uint64_t a[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t j0 = 0; j0 < dim0; j0 += 128) {
#pragma unroll UNROLL
        for (uint32_t j = j0; j < j0 + 128u; j++) {
            const u64x2 w = __builtin_nontemporal_load(dbp + (size_t)j * 64u);
            mac_j(a, q + j * 3u, w.x, w.y);
        }
        reduce6(a);
    }
#pragma unroll
    for (uint32_t r = 0; r < 3; r++) sh[wv * ROW + lane * 3u + r] = pack((uint32_t)a[r], (uint32_t)a[3 + r]);
    __syncthreads();
#pragma unroll
    for (uint32_t m = 0; m < 3; m++) {
        const uint32_t idx = threadIdx.x + ZW * 64u * m, res = idx / ZW, zz = idx - res * ZW;
        const uint32_t col = res / 3u, r = res - col * 3u, ic = icb * 64u + col, ii = ic >> 1, c = ic & 1u;
        uint64_t* dst = acc + ((size_t)(6u * ii + 2u * r + c)) * kN + zg * ZW + zz;
        if (NTS == 3) {
            const uint64_t val = sh[zz * ROW + res];
            asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1 nt" ::"v"(dst), "v"(val) : "memory");
        } else if (NTS == 4) {
            const uint64_t val = sh[zz * ROW + res];
            asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(dst), "v"(val) : "memory");
        } else if (NTS == 5) {
            const uint64_t val = sh[zz * ROW + res];
            asm volatile("global_store_dwordx2 %0, %1, off sc0 nt" ::"v"(dst), "v"(val) : "memory");
        } else if (NTS == 1) __builtin_nontemporal_store(sh[zz * ROW + res], dst);
        else if (NTS == 2) __hip_atomic_store(dst, sh[zz * ROW + res], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else *dst = sh[zz * ROW + res];
    }
}
// ---- packed database: 7 bytes per word (two 28-bit residues), 8 j = 16 words = 112 bytes = 7 x 16-byte loads per lane --------
// layout per tile: [j/8][k < 7][lane][16 B]
__global__ void pack_db(const uint64_t* __restrict__ db, uint32_t* __restrict__ pk, uint32_t dim0, uint32_t ntiles) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // (tile, group, lane)
    const uint32_t groups = dim0 / 8;
    if (g >= (size_t)ntiles * groups * 64) return;
    const uint32_t lane = g & 63u, grp = (g >> 6) % groups, tile = (uint32_t)((g >> 6) / groups);
    uint32_t out[28];
    for (int i = 0; i < 28; i++) out[i] = 0;
    for (uint32_t w = 0; w < 16; w++) {
        const uint32_t jj = w >> 1, m = w & 1u;
        const uint64_t v = db[((size_t)tile * dim0 + grp * 8 + jj) * 128u + lane * 2u + m];
        const uint64_t f = (uint64_t)lo32(v) | ((uint64_t)hi32(v) << 28);  // 56 bits
        const uint32_t bit = 56u * w, d = bit >> 5, sh = bit & 31u;
        out[d] |= (uint32_t)(f << sh);
        out[d + 1] |= (uint32_t)(sh ? (f >> (32 - sh)) : (f >> 32));
        if (sh > 8) out[d + 2] |= (uint32_t)(f >> (64 - sh));
    }
    for (uint32_t k = 0; k < 7; k++) {
        uint4* dst = reinterpret_cast<uint4*>(pk) + (((size_t)tile * groups + grp) * 7u + k) * 64u + lane;
        *dst = make_uint4(out[4 * k], out[4 * k + 1], out[4 * k + 2], out[4 * k + 3]);
    }
}
template <int T>
__device__ __forceinline__ uint32_t field28(const uint32_t (&D)[28]) {  // 28-bit field number T of the 112-byte group
    constexpr uint32_t bit = 28u * T, d = bit >> 5, sh = bit & 31u;
    if constexpr (sh <= 4) return (D[d] >> sh) & 0xFFFFFFFu;
    else return __builtin_amdgcn_alignbit(D[d + 1], D[d], sh) & 0xFFFFFFFu;
}
template <int JJ>
__device__ __forceinline__ void mac_packed_j(uint64_t (&a)[6], const uint4* q, const uint32_t (&D)[28]) {
    const uint4 qa = q[0], qb = q[1], qc = q[2];
    const uint32_t p0 = field28<4 * JJ>(D), b0 = field28<4 * JJ + 1>(D), p1 = field28<4 * JJ + 2>(D), b1 = field28<4 * JJ + 3>(D);
    a[0] += (uint64_t)qa.x * p0; a[1] += (uint64_t)qa.y * p0; a[2] += (uint64_t)qa.z * p0;
    a[3] += (uint64_t)qa.w * b0; a[4] += (uint64_t)qb.x * b0; a[5] += (uint64_t)qb.y * b0;
    a[0] += (uint64_t)qb.z * p1; a[1] += (uint64_t)qb.w * p1; a[2] += (uint64_t)qc.x * p1;
    a[3] += (uint64_t)qc.y * b1; a[4] += (uint64_t)qc.z * b1; a[5] += (uint64_t)qc.w * b1;
}
template <int ZW, int GU>
__global__ __launch_bounds__(ZW * 64) void sweep_packed(const uint32_t* __restrict__ pk, const uint32_t* __restrict__ qs, uint64_t* __restrict__ acc,
                                                        uint32_t nic, uint32_t dim0) {
    constexpr uint32_t ROW = 64 * 3 + 1;
    __shared__ uint64_t sh[ZW * ROW];
    const uint32_t lane = threadIdx.x & 63u, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t wpz = nic >> 6;
    const uint32_t zg = blockIdx.x / wpz, icb = blockIdx.x - zg * wpz;
    const uint32_t z = zg * ZW + wv, tile = z * wpz + icb, groups = dim0 / 8;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const u32x4* dbp = reinterpret_cast<const u32x4*>(pk) + (size_t)tile * groups * 7u * 64u + lane;
    const uint4* q = reinterpret_cast<const uint4*>(qs) + (size_t)z * dim0 * 3u;
    uint64_t a[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t g0 = 0; g0 < groups; g0 += 16) {  // 16 groups = 128 j = 256 terms between reductions
#pragma unroll GU
        for (uint32_t g = g0; g < g0 + 16u; g++) {
            uint32_t D[28];
#pragma unroll
            for (uint32_t k = 0; k < 7; k++) {
                const u32x4 v = __builtin_nontemporal_load(dbp + ((size_t)g * 7u + k) * 64u);
                D[4 * k] = v.x; D[4 * k + 1] = v.y; D[4 * k + 2] = v.z; D[4 * k + 3] = v.w;
            }
            const uint4* qg = q + (size_t)g * 24u;
            mac_packed_j<0>(a, qg, D); mac_packed_j<1>(a, qg + 3, D); mac_packed_j<2>(a, qg + 6, D); mac_packed_j<3>(a, qg + 9, D);
            mac_packed_j<4>(a, qg + 12, D); mac_packed_j<5>(a, qg + 15, D); mac_packed_j<6>(a, qg + 18, D); mac_packed_j<7>(a, qg + 21, D);
        }
        reduce6(a);
    }
#pragma unroll
    for (uint32_t r = 0; r < 3; r++) sh[wv * ROW + lane * 3u + r] = pack((uint32_t)a[r], (uint32_t)a[3 + r]);
    __syncthreads();
#pragma unroll
    for (uint32_t m = 0; m < 3; m++) {
        const uint32_t idx = threadIdx.x + ZW * 64u * m, res = idx / ZW, zz = idx - res * ZW;
        const uint32_t col = res / 3u, r = res - col * 3u, ic = icb * 64u + col, ii = ic >> 1, c = ic & 1u;
        acc[((size_t)(6u * ii + 2u * r + c)) * kN + zg * ZW + zz] = sh[zz * ROW + res];
    }
}
// read-only probes (no arithmetic beyond an xor, no query): the HBM read ceiling for a given access pattern.
// PATTERN 0: the sweep's (each wave streams its own contiguous dim0 KiB); 1: grid-interleaved (at step t wave w reads KiB t*nwaves + w)
template <int UNROLL, int PATTERN, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void read_probe(const uint64_t* __restrict__ db, uint64_t* __restrict__ acc, uint32_t dim0, uint32_t nwaves) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * WAVES + (threadIdx.x >> 6));
    const u64x2* base = reinterpret_cast<const u64x2*>(db) + lane;
    u64x2 x = {0, 0};
#pragma unroll UNROLL
    for (uint32_t j = 0; j < dim0; j++) {
        const size_t chunk = PATTERN == 0 ? (size_t)wave * dim0 + j : (size_t)j * nwaves + wave;
        const u64x2 w = __builtin_nontemporal_load(base + chunk * 64u);
        x ^= w;
    }
    if ((x.x ^ x.y) == 0x1234567ull) acc[wave * 64 + lane] = x.x;  // practically never
}

template <int UNROLL, bool NT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void sweep_v(const uint64_t* __restrict__ db, const uint32_t* __restrict__ qs, uint64_t* __restrict__ acc,
                                                      uint32_t nic, uint32_t dim0) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t tile = __builtin_amdgcn_readfirstlane(blockIdx.x * WAVES + (threadIdx.x >> 6));
    do_tile<UNROLL, NT>(db, qs, acc, nic, dim0, tile, lane);
}

// persistent: gridDim.x workgroups of WAVES waves stride over all tiles
template <int UNROLL, bool NT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void sweep_p(const uint64_t* __restrict__ db, const uint32_t* __restrict__ qs, uint64_t* __restrict__ acc,
                                                      uint32_t nic, uint32_t dim0, uint32_t ntiles) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w0 = __builtin_amdgcn_readfirstlane(blockIdx.x * WAVES + (threadIdx.x >> 6));
    for (uint32_t tile = w0; tile < ntiles; tile += gridDim.x * WAVES) do_tile<UNROLL, NT>(db, qs, acc, nic, dim0, tile, lane);
}

// two-pointer: each wave owns TWO tiles and alternates loads between them (more independent streams per wave)
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void sweep_2(const uint64_t* __restrict__ db, const uint32_t* __restrict__ qs, uint64_t* __restrict__ acc,
                                               uint32_t nic, uint32_t dim0) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
    // split the j range of one tile over two waves would need a combine; instead: half-length tiles (j split) with atomics is
    // avoided -- here each wave simply handles tile 2*wave and 2*wave+1 back to back
    do_tile<UNROLL, NT>(db, qs, acc, nic, dim0, 2 * wave, lane);
    do_tile<UNROLL, NT>(db, qs, acc, nic, dim0, 2 * wave + 1, lane);
}

__global__ void fill(uint64_t* p, size_t n, uint64_t seed) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint64_t x = seed + i + 0x9E3779B97F4A7C15ull;
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; x ^= x >> 31;
        p[i] = pack((uint32_t)(x & 0xffffffffull) % kP, (uint32_t)(x >> 32) % kB);
    }
}
__global__ void fillq(uint32_t* p, size_t n, uint64_t seed) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint64_t x = seed + i + 0x9E3779B97F4A7C15ull;
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; x ^= x >> 31;
        p[i] = (uint32_t)x % ((i % 6) < 3 ? kP : kB);
    }
}

static uint32_t* pkg;
struct Variant { const char* name; void (*launch)(const uint64_t*, const uint32_t*, uint64_t*, uint32_t, uint32_t); };

#define V(NAME, ...) {NAME, [](const uint64_t* db, const uint32_t* qs, uint64_t* acc, uint32_t nic, uint32_t dim0) { __VA_ARGS__; }}

int main(int argc, char** argv) {
    const uint32_t nu1 = argc > 1 ? atoi(argv[1]) : 8, nu2 = argc > 2 ? atoi(argv[2]) : 7;
    const uint32_t dim0 = 1u << nu1, num_per = 1u << nu2, nic = 2 * num_per, ntiles = kN * (nic / 64);
    const size_t dbw = (size_t)kN * dim0 * nic * 2, qw = (size_t)kN * dim0 * 12, accw = (size_t)num_per * 6 * kN;
    uint64_t *db, *acc, *ref; uint32_t* qs;
    hipMalloc(&db, dbw * 8); hipMalloc(&qs, qw * 4); hipMalloc(&acc, accw * 8); hipMalloc(&ref, accw * 8);
    fill<<<2048, 256>>>(db, dbw, 1); fillq<<<2048, 256>>>(qs, qw, 2);
    uint32_t* pk;
    hipMalloc(&pk, dbw * 7);
    pkg = pk;
    {
        const size_t n = (size_t)ntiles * (dim0 / 8) * 64;
        pack_db<<<(unsigned)((n + 255) / 256), 256>>>(db, pk, dim0, ntiles);
    }
    hipDeviceSynchronize();
    const double bytes = (double)dbw * 8 + (double)dim0 * 6 * kN * 8 + (double)num_per * 12 * kN * 8;
    std::vector<Variant> vs = {
        V("u4 nt w4 (v1 product)", hipLaunchKernelGGL((sweep_v<4, true, 4>), dim3(kN * (nic / 64) / 4), dim3(256), 0, 0, db, qs, acc, nic, dim0)),
        V("u8 nt w4", hipLaunchKernelGGL((sweep_v<8, true, 4>), dim3(kN * (nic / 64) / 4), dim3(256), 0, 0, db, qs, acc, nic, dim0)),
        V("u4 nt w1", hipLaunchKernelGGL((sweep_v<4, true, 1>), dim3(kN * (nic / 64)), dim3(64), 0, 0, db, qs, acc, nic, dim0)),
        V("u6 nt w1", hipLaunchKernelGGL((sweep_v<6, true, 1>), dim3(kN * (nic / 64)), dim3(64), 0, 0, db, qs, acc, nic, dim0)),
        V("u3 nt w1", hipLaunchKernelGGL((sweep_v<3, true, 1>), dim3(kN * (nic / 64)), dim3(64), 0, 0, db, qs, acc, nic, dim0)),
        V("u4 nt w2", hipLaunchKernelGGL((sweep_v<4, true, 2>), dim3(kN * (nic / 64) / 2), dim3(128), 0, 0, db, qs, acc, nic, dim0)),
        V("u8 nt w2", hipLaunchKernelGGL((sweep_v<8, true, 2>), dim3(kN * (nic / 64) / 2), dim3(128), 0, 0, db, qs, acc, nic, dim0)),
        V("u6 nt w2", hipLaunchKernelGGL((sweep_v<6, true, 2>), dim3(kN * (nic / 64) / 2), dim3(128), 0, 0, db, qs, acc, nic, dim0)),
        V("u6 nt w4", hipLaunchKernelGGL((sweep_v<6, true, 4>), dim3(kN * (nic / 64) / 4), dim3(256), 0, 0, db, qs, acc, nic, dim0)),
        V("u12 nt w4", hipLaunchKernelGGL((sweep_v<12, true, 4>), dim3(kN * (nic / 64) / 4), dim3(256), 0, 0, db, qs, acc, nic, dim0)),
        V("rot8K  u8 w2", hipLaunchKernelGGL((sweep_rot<8, 0, 2>), dim3(kN * (nic / 64) / 2), dim3(128), 0, 0, db, qs, acc, nic, dim0)),
        V("rot40K u8 w2", hipLaunchKernelGGL((sweep_rot<8, 1, 2>), dim3(kN * (nic / 64) / 2), dim3(128), 0, 0, db, qs, acc, nic, dim0)),
        V("rotHash u8 w2", hipLaunchKernelGGL((sweep_rot<8, 2, 2>), dim3(kN * (nic / 64) / 2), dim3(128), 0, 0, db, qs, acc, nic, dim0)),
        V("R-abl half MACs u8 w2", hipLaunchKernelGGL((sweep_abl<8, 1, 2>), dim3(kN * (nic / 64) / 2), dim3(128), 0, 0, db, qs, acc, nic, dim0)),
        V("R-abl no s_load u8 w2", hipLaunchKernelGGL((sweep_abl<8, 2, 2>), dim3(kN * (nic / 64) / 2), dim3(128), 0, 0, db, qs, acc, nic, dim0)),
        V("R-abl both u8 w2", hipLaunchKernelGGL((sweep_abl<8, 3, 2>), dim3(kN * (nic / 64) / 2), dim3(128), 0, 0, db, qs, acc, nic, dim0)),
        V("u16 nt w2", hipLaunchKernelGGL((sweep_v<16, true, 2>), dim3(kN * (nic / 64) / 2), dim3(128), 0, 0, db, qs, acc, nic, dim0)),
        V("u16 nt w1", hipLaunchKernelGGL((sweep_v<16, true, 1>), dim3(kN * (nic / 64)), dim3(64), 0, 0, db, qs, acc, nic, dim0)),
        V("u8 nt w1", hipLaunchKernelGGL((sweep_v<8, true, 1>), dim3(kN * (nic / 64)), dim3(64), 0, 0, db, qs, acc, nic, dim0)),
        V("zl8 u8", hipLaunchKernelGGL((sweep_zl<8, 8>), dim3(kN / 8 * (nic / 64)), dim3(512), 0, 0, db, qs, acc, nic, dim0)),
        V("zl16 u8", hipLaunchKernelGGL((sweep_zl<8, 16>), dim3(kN / 16 * (nic / 64)), dim3(1024), 0, 0, db, qs, acc, nic, dim0)),
        V("zl4 u8", hipLaunchKernelGGL((sweep_zl<8, 4>), dim3(kN / 4 * (nic / 64)), dim3(256), 0, 0, db, qs, acc, nic, dim0)),
        V("zl8 u6", hipLaunchKernelGGL((sweep_zl<6, 8>), dim3(kN / 8 * (nic / 64)), dim3(512), 0, 0, db, qs, acc, nic, dim0)),
        V("zl8 u16", hipLaunchKernelGGL((sweep_zl<16, 8>), dim3(kN / 8 * (nic / 64)), dim3(512), 0, 0, db, qs, acc, nic, dim0)),
        V("zl8 u8 nt-store", hipLaunchKernelGGL((sweep_zl<8, 8, 1>), dim3(kN / 8 * (nic / 64)), dim3(512), 0, 0, db, qs, acc, nic, dim0)),
        V("zl16 u8 nt-store", hipLaunchKernelGGL((sweep_zl<8, 16, 1>), dim3(kN / 16 * (nic / 64)), dim3(1024), 0, 0, db, qs, acc, nic, dim0)),
        V("zl16 u8 asm sc0sc1nt", hipLaunchKernelGGL((sweep_zl<8, 16, 3>), dim3(kN / 16 * (nic / 64)), dim3(1024), 0, 0, db, qs, acc, nic, dim0)),
        V("zl16 u8 asm sc1", hipLaunchKernelGGL((sweep_zl<8, 16, 4>), dim3(kN / 16 * (nic / 64)), dim3(1024), 0, 0, db, qs, acc, nic, dim0)),
        V("zl16 u8 asm sc0nt", hipLaunchKernelGGL((sweep_zl<8, 16, 5>), dim3(kN / 16 * (nic / 64)), dim3(1024), 0, 0, db, qs, acc, nic, dim0)),
        V("zl8 u8 sys-store", hipLaunchKernelGGL((sweep_zl<8, 8, 2>), dim3(kN / 8 * (nic / 64)), dim3(512), 0, 0, db, qs, acc, nic, dim0)),
        V("zl16 u8 sys-store", hipLaunchKernelGGL((sweep_zl<8, 16, 2>), dim3(kN / 16 * (nic / 64)), dim3(1024), 0, 0, db, qs, acc, nic, dim0)),
        V("persist 4096w u8 w2", hipLaunchKernelGGL((sweep_p<8, true, 2>), dim3(2048), dim3(128), 0, 0, db, qs, acc, nic, dim0, kN * (nic / 64))),
        V("persist 2048w u8 w2", hipLaunchKernelGGL((sweep_p<8, true, 2>), dim3(1024), dim3(128), 0, 0, db, qs, acc, nic, dim0, kN * (nic / 64))),
        V("persist 4096w u16 w2", hipLaunchKernelGGL((sweep_p<16, true, 2>), dim3(2048), dim3(128), 0, 0, db, qs, acc, nic, dim0, kN * (nic / 64))),
        V("persist 2048w u16 w2", hipLaunchKernelGGL((sweep_p<16, true, 2>), dim3(1024), dim3(128), 0, 0, db, qs, acc, nic, dim0, kN * (nic / 64))),
        V("persist 4096w u8 w1", hipLaunchKernelGGL((sweep_p<8, true, 1>), dim3(4096), dim3(64), 0, 0, db, qs, acc, nic, dim0, kN * (nic / 64))),
        V("persist 6144w u8 w2", hipLaunchKernelGGL((sweep_p<8, true, 2>), dim3(3072), dim3(128), 0, 0, db, qs, acc, nic, dim0, kN * (nic / 64))),
        V("16384 waves half-j? n/a", hipLaunchKernelGGL((sweep_p<8, true, 2>), dim3(1536), dim3(128), 0, 0, db, qs, acc, nic, dim0, kN * (nic / 64))),
        V("PACKED zl16 gu1", hipLaunchKernelGGL((sweep_packed<16, 1>), dim3(kN / 16 * (nic / 64)), dim3(1024), 0, 0, pkg, qs, acc, nic, dim0)),
        V("PACKED zl16 gu2", hipLaunchKernelGGL((sweep_packed<16, 2>), dim3(kN / 16 * (nic / 64)), dim3(1024), 0, 0, pkg, qs, acc, nic, dim0)),
        V("PACKED zl8 gu1", hipLaunchKernelGGL((sweep_packed<8, 1>), dim3(kN / 8 * (nic / 64)), dim3(512), 0, 0, pkg, qs, acc, nic, dim0)),
        V("R-bis plain", hipLaunchKernelGGL((read_bisect<8, 0, 2>), dim3(kN * (nic / 64) / 2), dim3(128), 0, 0, db, qs, acc, nic, dim0)),
        V("R-bis 6acc-xor", hipLaunchKernelGGL((read_bisect<8, 16, 2>), dim3(kN * (nic / 64) / 2), dim3(128), 0, 0, db, qs, acc, nic, dim0)),
        V("R-bis 6mac", hipLaunchKernelGGL((read_bisect<8, 1, 2>), dim3(kN * (nic / 64) / 2), dim3(128), 0, 0, db, qs, acc, nic, dim0)),
        V("R-bis 12mac", hipLaunchKernelGGL((read_bisect<8, 2, 2>), dim3(kN * (nic / 64) / 2), dim3(128), 0, 0, db, qs, acc, nic, dim0)),
        V("R-bis 12mac+reduce", hipLaunchKernelGGL((read_bisect<8, 6, 2>), dim3(kN * (nic / 64) / 2), dim3(128), 0, 0, db, qs, acc, nic, dim0)),
        V("R-bis 12mac+reduce+store", hipLaunchKernelGGL((read_bisect<8, 14, 2>), dim3(kN * (nic / 64) / 2), dim3(128), 0, 0, db, qs, acc, nic, dim0)),
        V("R-bis 12mac+reduce+SMALLstore", hipLaunchKernelGGL((read_bisect<8, 6 + 32, 2>), dim3(kN * (nic / 64) / 2), dim3(128), 0, 0, db, qs, acc, nic, dim0)),
        V("R-bis 12mac+reduce+WAVEstore", hipLaunchKernelGGL((read_bisect<8, 6 + 64, 2>), dim3(kN * (nic / 64) / 2), dim3(128), 0, 0, db, qs, acc, nic, dim0)),
        V("R-bis 12mac+reduce+EARLYstore", hipLaunchKernelGGL((read_bisect<8, 6 + 128, 2>), dim3(kN * (nic / 64) / 2), dim3(128), 0, 0, db, qs, acc, nic, dim0)),
        V("R-bis 12mac+reduce+EARLYquarter", hipLaunchKernelGGL((read_bisect<8, 6 + 256, 2>), dim3(kN * (nic / 64) / 2), dim3(128), 0, 0, db, qs, acc, nic, dim0)),
        V("R-bis plain+store", hipLaunchKernelGGL((read_bisect<8, 8, 2>), dim3(kN * (nic / 64) / 2), dim3(128), 0, 0, db, qs, acc, nic, dim0)),
        V("R-bis plain+reduce", hipLaunchKernelGGL((read_bisect<8, 4, 2>), dim3(kN * (nic / 64) / 2), dim3(128), 0, 0, db, qs, acc, nic, dim0)),
        V("READ-ONLY own-stream u8 w2", hipLaunchKernelGGL((read_probe<8, 0, 2>), dim3(kN * (nic / 64) / 2), dim3(128), 0, 0, db, acc, dim0, kN * (nic / 64))),
        V("READ-ONLY own-stream u16 w2", hipLaunchKernelGGL((read_probe<16, 0, 2>), dim3(kN * (nic / 64) / 2), dim3(128), 0, 0, db, acc, dim0, kN * (nic / 64))),
        V("READ-ONLY interleaved u8 w2", hipLaunchKernelGGL((read_probe<8, 1, 2>), dim3(kN * (nic / 64) / 2), dim3(128), 0, 0, db, acc, dim0, kN * (nic / 64))),
        V("READ-ONLY interleaved u8 w4", hipLaunchKernelGGL((read_probe<8, 1, 4>), dim3(kN * (nic / 64) / 4), dim3(256), 0, 0, db, acc, dim0, kN * (nic / 64))),
        V("READ-ONLY interleaved u16 w4", hipLaunchKernelGGL((read_probe<16, 1, 4>), dim3(kN * (nic / 64) / 4), dim3(256), 0, 0, db, acc, dim0, kN * (nic / 64))),
        V("persist 8192w u4 nt w1", hipLaunchKernelGGL((sweep_p<4, true, 1>), dim3(8192), dim3(64), 0, 0, db, qs, acc, nic, dim0, kN * (nic / 64))),
    };
    (void)ntiles;
    std::vector<std::vector<float>> times(vs.size());
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<uint64_t> href(accw), hacc(accw);
    for (int round = 0; round < 6; round++) {
        for (size_t v = 0; v < vs.size(); v++) {
            hipMemset(acc, 0, accw * 8);
            hipEventRecord(e0);
            vs[v].launch(db, qs, acc, nic, dim0);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (round > 0) times[v].push_back(ms);
            if (round == 0) {
                hipMemcpy(v == 0 ? href.data() : hacc.data(), acc, accw * 8, hipMemcpyDeviceToHost);
                if (v > 0 && vs[v].name[0] != 'R' && href != hacc) printf("!! variant %s differs from variant 0\n", vs[v].name);
            }
        }
    }
    for (size_t v = 0; v < vs.size(); v++) {
        std::sort(times[v].begin(), times[v].end());
        float med = times[v][times[v].size() / 2], mn = times[v][0];
        printf("%-28s median %7.1f us (min %7.1f)  %6.1f GB/s  %4.1f%% of 8 TB/s\n", vs[v].name, med * 1e3, mn * 1e3, bytes / (med * 1e-3) / 1e9,
               bytes / (med * 1e-3) / 8e12 * 100);
    }
    return 0;
}
