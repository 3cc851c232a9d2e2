// Shared constants and device arithmetic for the Spiral server-answer kernels (gfx950).
//
// Internal data formats (device side):
//   PK  "packed NTT form": one u64 per NTT slot, low 32 bits = residue mod p, high 32 bits = residue
//       mod b (the reference packs the same way for its hot loops, src/spiral.cpp:345-433).  A
//       polynomial is 2048 u64 = 16 KiB.  Slot order is the reference's (bit-reversed) order.
//   RAW one u64 per coefficient, value in [0, Q] (reference raw form, include/poly.h:24-64).
// Residues are < 2^28, so every product fits u64 (< 2^56) and lazy butterflies fit u32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace spiral {

constexpr uint32_t kN = 2048;
constexpr uint32_t kLogN = 11;
constexpr uint32_t kP = 268369921u;  // include/values.h:13
constexpr uint32_t kB = 249561089u;  // include/values.h:21
constexpr uint64_t kQ = (uint64_t)kP * kB;
constexpr uint32_t kPinvB = 97389680u;  // p^-1 mod b, include/values.h:24
constexpr uint32_t kN0 = 2, kN1 = 3, kN2 = 2;

__device__ __forceinline__ uint32_t lo32(uint64_t x) { return (uint32_t)x; }
__device__ __forceinline__ uint32_t hi32(uint64_t x) { return (uint32_t)(x >> 32); }
__device__ __forceinline__ uint64_t pack(uint32_t lo, uint32_t hi) { return (uint64_t)lo | ((uint64_t)hi << 32); }

// exact reductions of a u64 (include/poly.h:137-153 computes the same function with Barrett)
__device__ __forceinline__ uint32_t mod_p(uint64_t x) { return (uint32_t)(x % kP); }
__device__ __forceinline__ uint32_t mod_b(uint64_t x) { return (uint32_t)(x % kB); }
// reduce a u32 known to be < 4m / < 2m
__device__ __forceinline__ uint32_t csub(uint32_t x, uint32_t m) { return x >= m ? x - m : x; }

// CRT lift of (x mod p, y mod b) to the canonical value in [0, Q): Garner form of
// src/poly.cpp:344-353 (same unique result, no 128-bit arithmetic)
__device__ __forceinline__ uint64_t crt_compose(uint32_t x, uint32_t y) {
    uint32_t xb = csub(x, kB);                       // x < p < 2b
    uint32_t d = y >= xb ? y - xb : y + kB - xb;     // (y - x) mod b
    uint32_t k = mod_b((uint64_t)d * kPinvB);
    return (uint64_t)x + (uint64_t)kP * k;
}

// Physical position of NTT slot s inside a PK polynomial.  The transforms hold slots 8*tid .. 8*tid+7 in thread tid,
// so PK buffers are stored "thread-transposed" in pairs: slots 8*tid + 2q and 8*tid + 2q + 1 live at q*512 + 2*tid and
// q*512 + 2*tid + 1.  A thread then moves its 8 slots with four 16-byte accesses and a wave reads / writes 1 KiB of
// contiguous bytes per instruction instead of 64 scattered 64-byte lines (the strided form made the forward transforms
// store-bound).  Everything pointwise is oblivious to the order; only the boundary kernels that meet the reference's
// slot order (ref<->PK conversion, database / query relayout) apply the map.
__host__ __device__ inline uint32_t pk_pos(uint32_t s) { return (((s & 7u) >> 1) << 9) | ((s >> 3) << 1) | (s & 1u); }
// the same for register k of thread tid (slot 8*tid + k)
__host__ __device__ inline uint32_t pk_pos_tk(uint32_t tid, uint32_t k) { return ((k >> 1) << 9) + 2u * tid + (k & 1u); }

// Device database layout (internal; built at load time).  The nic = 2*num_per output columns ic = ii*2 + c are
// grouped in blocks of 64 -- one wave of the sweep owns one (z, block) tile and streams it front to back.
//
// Packed layout (dim0 % 8 == 0, every real geometry): a database word is two 28-bit residues, so it is stored in 7 bytes,
// not 8: the sweep is bound by streaming the database and 1/8 of the reference's bytes are zero bits.  A lane's 16 words
// of 8 consecutive j (j-major, m minor) form one 112-byte little-endian bit string, word w at bits [56w, 56w + 56) =
// p-residue | b-residue << 28, fetched as 7 x 16 bytes:
//     tile -> group j/8 -> chunk k < 7 -> lane -> 16 bytes
// so every load instruction of a wave covers 1 KiB of consecutive addresses and a tile is one sequential stream of
// dim0/8 * 7 KiB.  A tile is 64 lanes wide: W = min(64, nic) columns x P = 64/W consecutive z.  With nic >= 64 a tile is
// (z, block of 64 columns); with fewer columns (small nu2: the streaming parameter sets) a wave takes P slots z at once,
// lane = (z % P) * W + ic.
// Plain layout (dim0 < 8, tiny test geometries only): word(z, j, ic, m) at (((z*nblk + ic/W)*dim0 + j)*W + ic%W)*2 + m.
__host__ __device__ inline uint32_t db_block_width(uint32_t nic) { return nic < 64u ? nic : 64u; }
__host__ __device__ inline size_t db_word_index(uint32_t z, uint32_t j, uint32_t ic, uint32_t m, uint32_t nic, uint32_t dim0) {
    const uint32_t w = db_block_width(nic), nblk = nic / w;
    return ((((size_t)z * nblk + ic / w) * dim0 + j) * w + ic % w) * 2u + m;
}
__host__ __device__ inline bool db_packed(uint32_t nic, uint32_t dim0) { return (dim0 & 7u) == 0u && nic >= 2u && (nic & (nic - 1u)) == 0u; }
// u64 words to allocate for the device database of one shard
__host__ __device__ inline size_t db_device_words(uint32_t nic, uint32_t dim0) {
    const size_t words = (size_t)kN * dim0 * nic * 2u;
    return db_packed(nic, dim0) ? words / 8u * 7u : words;
}
// tile and lane of (z, ic): W columns x P slots per tile
__host__ __device__ inline void db_tile_lane(uint32_t z, uint32_t ic, uint32_t nic, uint32_t& tile, uint32_t& lane) {
    const uint32_t w = db_block_width(nic), pz = 64u / w, nblk = nic / w;
    tile = (z / pz) * nblk + ic / w;
    lane = (z % pz) * w + ic % w;
}
// byte address of byte `by` < 7 of word (z, j, ic, m) in the packed layout
__host__ __device__ inline size_t db_packed_byte(uint32_t z, uint32_t j, uint32_t ic, uint32_t m, uint32_t by, uint32_t nic, uint32_t dim0) {
    uint32_t tile, lane;
    db_tile_lane(z, ic, nic, tile, lane);
    const uint32_t b = 7u * ((j & 7u) * 2u + m) + by;
    return ((((size_t)tile * (dim0 >> 3) + (j >> 3)) * 7u + (b >> 4)) * 64u + lane) * 16u + (b & 15u);
}
// store one database word (p-residue | b-residue << 32, both < 2^28) in whichever layout the geometry uses
__device__ __forceinline__ void db_put_word(uint64_t* db, uint32_t z, uint32_t j, uint32_t ic, uint32_t m, uint32_t nic, uint32_t dim0, uint64_t v) {
    if (db_packed(nic, dim0)) {
        const uint64_t f = (uint64_t)lo32(v) | ((uint64_t)hi32(v) << 28);
        uint8_t* bytes = reinterpret_cast<uint8_t*>(db);
#pragma unroll
        for (uint32_t by = 0; by < 7; by++) bytes[db_packed_byte(z, j, ic, m, by, nic, dim0)] = (uint8_t)(f >> (8u * by));
    } else {
        db[db_word_index(z, j, ic, m, nic, dim0)] = v;
    }
}

// read one database word back (the inverse of db_put_word): tests and the read_db_* entry points
__device__ __forceinline__ uint64_t db_get_word(const uint64_t* db, uint32_t z, uint32_t j, uint32_t ic, uint32_t m, uint32_t nic, uint32_t dim0) {
    if (db_packed(nic, dim0)) {
        const uint8_t* bytes = reinterpret_cast<const uint8_t*>(db);
        uint64_t f = 0;
#pragma unroll
        for (uint32_t by = 0; by < 7; by++) f |= (uint64_t)bytes[db_packed_byte(z, j, ic, m, by, nic, dim0)] << (8u * by);
        return (f & 0xFFFFFFFull) | ((f >> 28) << 32);
    }
    return db[db_word_index(z, j, ic, m, nic, dim0)];
}

// The same word from the limb-plane form of the image (sweep_mfma.hip: [z][16 columns][prime][piece of 128 terms t = 2 (j & 63) + m][7 x 1 KiB],
// 1 KiB = limb i < 3 of chunk c = t >> 6 as signed bytes, the seventh the 4-bit top limbs of both chunks; lane = ((t >> 4) & 3) * 16 + (ic & 15), byte t & 15).
// A residue a is stored as a'' = a or a - m in [-0x808080, 2^28 - 0x808080): a'' + 0x808080 = (s0 + 128) | (s1 + 128) << 8 | (s2 + 128) << 16 | u3 << 24.
__device__ __forceinline__ uint64_t db_get_word_limbs(const uint64_t* db, uint32_t z, uint32_t j, uint32_t ic, uint32_t m, uint32_t nic, uint32_t dim0) {
    const uint8_t* bytes = reinterpret_cast<const uint8_t*>(db);
    const uint32_t nk2 = dim0 >> 6, t = 2u * (j & 63u) + m, c = t >> 6, lane = ((t >> 4) & 3u) * 16u + (ic & 15u), e = t & 15u;
    uint32_t res[2];
#pragma unroll
    for (uint32_t pr = 0; pr < 2; pr++) {
        const size_t piece = (((size_t)z * (nic >> 4) + (ic >> 4)) * 2u + pr) * nk2 + (j >> 6);  // 7 KiB each
        const uint8_t* b = bytes + piece * 7168u + (size_t)lane * 16u + e;
        uint32_t w = 0;
#pragma unroll
        for (uint32_t i = 0; i < 3; i++) w |= (uint32_t)(b[(2u * i + c) * 1024u] ^ 0x80u) << (8u * i);
        w |= ((uint32_t)(b[6u * 1024u] >> (4u * c)) & 0xFu) << 24;
        const int32_t a = (int32_t)w - 0x808080;
        res[pr] = a < 0 ? (uint32_t)(a + (int32_t)(pr ? kB : kP)) : (uint32_t)a;
    }
    return pack(res[0], res[1]);
}

// Plaintext coefficient number ci of a bit-packed item stream (raw database ingest): coefficients are coeff_bits wide,
// little-endian bit order as the reference's read_arbitrary_bits (src/core.cpp:20-30); coeff_bits == 64: plain u64 words
// (the reference's raw MatPoly).  The staging buffer carries 16 bytes of padding, so the 12-byte window may over-read.
__device__ __forceinline__ uint64_t packed_coeff(const uint8_t* items, uint64_t ci, uint32_t coeff_bits) {
    if (coeff_bits == 64) return reinterpret_cast<const uint64_t*>(items)[ci];
    const uint64_t bit = ci * coeff_bits;
    const uint32_t* w = reinterpret_cast<const uint32_t*>(items) + (bit >> 5);
    const uint32_t sh = (uint32_t)(bit & 31u);
    const uint64_t lo = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
    uint64_t v = lo >> sh;
    if (sh + coeff_bits > 64u) v |= (uint64_t)w[2] << (64u - sh);
    return v & ((1ull << coeff_bits) - 1ull);
}

// response modulus switch of one coefficient (src/poly.cpp:578-601):
// round(centre(a) * out_mod / inp_mod) mod out_mod with the reference's round-half-away-from-zero and
// truncating division; the 128-bit quotient is a double estimate corrected exactly.
__device__ __forceinline__ uint64_t rescale_dev(uint64_t a, uint64_t inp_mod, uint64_t out_mod) {
    a %= inp_mod;
    const bool neg = a >= inp_mod / 2;
    const uint64_t mag = neg ? inp_mod - a : a;
    const unsigned __int128 x = (unsigned __int128)mag * out_mod + inp_mod / 2;
    const double xd = (double)(uint64_t)(x >> 64) * 18446744073709551616.0 + (double)(uint64_t)x;
    uint64_t q = (uint64_t)(xd / (double)inp_mod);
    __int128 r = (__int128)x - (__int128)((unsigned __int128)q * inp_mod);
    while (r < 0) {
        q--;
        r += inp_mod;
    }
    while (r >= (__int128)inp_mod) {
        q++;
        r -= inp_mod;
    }
    uint64_t res = q % out_mod;
    return (neg && res != 0) ? out_mod - res : res;
}

// 28-bit field T of a 112-byte group of the packed database (sweep.hip, sweep_mfma.hip, pack.hip)
template <int T>
__device__ __forceinline__ uint32_t field28(const uint32_t (&d)[28]) {
    constexpr uint32_t bit = 28u * T, w = bit >> 5, sh = bit & 31u;
    if constexpr (sh <= 4)
        return (d[w] >> sh) & 0xFFFFFFFu;
    else
        return __builtin_amdgcn_alignbit(d[w + 1], d[w], sh) & 0xFFFFFFFu;
}

// position of ciphertext i0 in the sweep's accumulator buffer (sweep.hip: [stage][rank][ct])
__device__ __forceinline__ uint32_t acc_pos(uint32_t i0, uint32_t g_log, uint32_t ls_log) {
    const uint32_t g = i0 & ((1u << g_log) - 1u), k = i0 >> g_log;
    return ((((k >> ls_log) << g_log) | g) << ls_log) | (k & ((1u << ls_log) - 1u));
}

// include/util.h:34-38
__host__ __device__ constexpr uint32_t get_bits_per(uint32_t dim) { return dim == 56 ? 1u : 56u / dim + 1u; }

}  // namespace spiral
