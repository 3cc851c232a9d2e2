// Loader arithmetic of the batched transforms (ntt.hip): gadget digits of raw coefficients in the forms the reference's hot path uses --
// unsigned digits (gadget_invert, src/util.cpp:114-144), the balanced digits of split_and_crt (src/spiral.cpp:270-330) in three tiers
// (carry-free differences, 32-bit carry compare, the generic 64-bit walk), the automorphism gather and the seeded plaintext generator.
// Everything here is per-coefficient integer code with wave-uniform parameters; the transforms themselves are in ntt_device.h.
#pragma once
#include <type_traits>

#include "common.h"
#include "ntt_device.h"

namespace spiral {

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// raw coefficient `idx` of the automorphed polynomial a(x^t): gather form of src/poly.cpp:240-261.
// e = idx * t^-1 mod 2N; source index e mod N, negated (as Q - a, so 0 -> Q) when e >= N.
__device__ __forceinline__ uint64_t load_raw(const uint64_t* src, uint32_t idx, uint32_t tinv) {
    if (tinv == 0) return src[idx];
    uint32_t e = (idx * tinv) & (2 * kN - 1);
    uint64_t v = src[e & (kN - 1)];
    return (e & kN) ? kQ - v : v;
}

__device__ __forceinline__ uint64_t digit_of(uint64_t v, uint32_t k, uint32_t bits, uint64_t mask) {
    uint32_t sh = k * bits;
    if (bits <= 27u) {  // the value's two words and a wave-uniform offset: one v_alignbit instead of a 64-bit shift
        const bool up = sh >= 32u;
        const uint32_t d = __builtin_amdgcn_alignbit(up ? 0u : hi32(v), up ? hi32(v) : lo32(v), sh & 31u) & (uint32_t)mask;
        return sh >= 64 ? 0u : d;
    }
    return sh >= 64 ? 0ull : ((v >> sh) & mask);  // a shift count >= 64 is UB in src/util.cpp:136; defined as 0
}

// unsigned digits k of a thread's 8 coefficients, digit width <= 27 bits: which word the digit starts in is wave-uniform, so the choice is
// made once around the loop (inside it the compiler turns it into two selects per value)
__device__ __forceinline__ void udigits8(const uint64_t* raw, uint32_t k, uint32_t bits, uint32_t* d) {
    const uint32_t sh = k * bits, mask = (1u << bits) - 1u;
    if (sh >= 64u) {
#pragma unroll
        for (int r = 0; r < 8; r++) d[r] = 0;
    } else if (sh >= 32u) {
#pragma unroll
        for (int r = 0; r < 8; r++) d[r] = (hi32(raw[r]) >> (sh - 32u)) & mask;
    } else {
#pragma unroll
        for (int r = 0; r < 8; r++) d[r] = __builtin_amdgcn_alignbit(hi32(raw[r]), lo32(raw[r]), sh) & mask;
    }
}
// b / d for a wave-uniform job index b and a small divisor (digit counts <= 56, b < 2^26) as one scalar multiply: inv = 2^32 / d + 1
// (launch_ntt_forward fills it in); the plain division is a float reciprocal sequence on the vector ALU
__device__ __forceinline__ uint32_t udiv_small(uint32_t b, uint32_t d, uint32_t inv) { return d == 1u ? b : __umulhi(b, inv); }

// a gadget digit is < 2^bits <= 2^32 (bits = 32 only through the to_ntt_no_reduce seam, whose contract is
// values < 2^29); the forward transform wants its inputs below 2m.
// SMALL: the digit width is at most 27 bits -- every published parameter set (bits = floor(56/t) + 1 <= 15 for t >= 4; 29 only
// for t = 2) -- so a digit, and a balanced piece <= 2^bits, is already below 2^28 < 2m and needs no reduction.  The loaders
// branch on that once per workgroup (it is a launch parameter) instead of carrying a compare + remainder per coefficient.
template <bool SMALL>
__device__ __forceinline__ uint32_t digit_residue(uint32_t d, uint32_t m) {
    if constexpr (SMALL) return d;
    return d < (1u << 28) ? d : d % m;
}
constexpr uint32_t kSmallDigitBits = 27;
#define DIGIT_WIDTH_DISPATCH(bits, body) \
    do {                                 \
        if ((bits) <= kSmallDigitBits)   \
            body(std::true_type{});      \
        else                             \
            body(std::false_type{});     \
    } while (0)

// balanced digit k of v under split_and_crt's two carry chains (src/spiral.cpp:283-292, 313-322): digits 0..ell/2-1 and
// ell/2..ell-1 each propagate a carry (piece > 2^bits/2 borrows 2^bits from the next digit), the first chain's last digit
// never borrows.  The reference walks the chain; the carry into position j of a chain is a pure function of the chain's
// low j digits L:  carry_1 = [d_0 > B/2],  carry_{j+1} = [d_j > B/2] or [d_j == B/2 and carry_j]  ==  [L_{j+1} > T_{j+1}]
// with T_j = (B/2)(1 + B + ... + B^(j-1)), so one mask-and-compare replaces the walk.
struct SDigit {
    uint32_t sh_chain, sh_digit;  // bit offsets of the chain start and of digit k
    uint64_t low_mask, thresh_in; // L = (v >> sh_chain) & low_mask ; carry-in = L > thresh_in (j > 0)
    uint64_t mask, base, thresh;
    bool has_in, may;
};
__device__ __forceinline__ SDigit sdigit_setup(uint32_t k, uint32_t bits, uint32_t ell) {
    SDigit d;
    const uint32_t half = ell >> 1, start = k < half ? 0u : half, j = k - start;
    d.base = 1ull << bits;
    d.mask = d.base - 1;
    d.thresh = d.base >> 1;
    d.sh_chain = start * bits;
    d.sh_digit = k * bits;
    d.has_in = j > 0;
    d.low_mask = (bits * j >= 64) ? ~0ull : ((1ull << (bits * j)) - 1);
    uint64_t t = 0;
    for (uint32_t i = 0; i < j; i++) t = (t << bits) + d.thresh;
    d.thresh_in = t;
    d.may = k < half ? (k + 1 < half) : true;
    return d;
}
// returned as residues (mod p, mod b); a borrowed digit is piece + Q - 2^bits == piece - 2^bits (mod m)
template <bool SMALL>
__device__ __forceinline__ void sdigit_of(uint64_t v, const SDigit& d, uint32_t& rp, uint32_t& rb) {
    const uint64_t dig = d.sh_digit >= 64 ? 0ull : ((v >> d.sh_digit) & d.mask);  // shift counts >= 64: 0, as digit_of
    const uint64_t low = d.sh_chain >= 64 ? 0ull : ((v >> d.sh_chain) & d.low_mask);
    if constexpr (SMALL) {  // everything fits 32 bits and both outcomes are cheap: select instead of branching
        const uint32_t piece = (uint32_t)dig + ((d.has_in && low > d.thresh_in) ? 1u : 0u);
        const uint32_t x = (uint32_t)d.base - piece;  // in [0, 2^bits/2) when borrowed
        const bool borrow = piece > (uint32_t)d.thresh && d.may;
        rp = borrow ? kP - x : piece;
        rb = borrow ? kB - x : piece;
    } else {
        const uint64_t piece = dig + ((d.has_in && low > d.thresh_in) ? 1u : 0u);
        if (piece > d.thresh && d.may) {
            const uint32_t x = (uint32_t)(d.base - piece);  // in [0, 2^bits/2)
            rp = kP - x;
            rb = kB - x;
        } else {
            rp = digit_residue<false>((uint32_t)piece, kP);
            rb = digit_residue<false>((uint32_t)piece, kB);
        }
    }
}

// The same digit as a signed integer (piece, or piece - 2^bits when it borrows) in 32-bit arithmetic: the value is below 2^56 and
// every offset is a launch constant, so the digit and the chain's low part are v_alignbit / shift extractions from the value's two
// words, 9 instructions per value (the 64-bit shifts, masks and compares of the generic form are the most expensive part of a
// digit loader).  "No carry in" and "never borrows" are encoded in the constants (an empty mask against a full threshold), and
// which word the digit and the chain start in (DHI, CHI) is a wave-uniform choice made once around the 8-value loop.
// ok: digits of at most 27 bits whose chain prefix fits one word (j * bits <= 32) -- every published parameter set; the loaders fall
// back to sdigit_of<false> otherwise.
struct SDig32 {
    uint32_t d_sh, d_mask;            // digit: shift within the word pair (or within the high word), mask (0 beyond bit 63)
    uint32_t c_sh, c_mask, c_thresh;  // chain prefix (the low j digits) and its carry threshold T_j
    uint32_t base, thresh;
    bool d_hi, c_hi, ok;              // the digit / the chain starts in the high word
};
__device__ __forceinline__ SDig32 sdig32_setup(uint32_t k, uint32_t bits, uint32_t ell) {
    SDig32 d;
    const uint32_t half = ell >> 1, start = k < half ? 0u : half, j = k - start;
    const uint32_t o = k * bits, oc = start * bits, w = j * bits;
    d.ok = bits <= kSmallDigitBits && w <= 32u;
    d.base = 1u << (bits & 31u);
    const bool may = k < half ? (k + 1 < half) : true;
    d.thresh = may ? d.base >> 1 : ~0u;
    d.d_hi = o >= 32u;
    d.d_sh = o & 31u;
    d.d_mask = o >= 64u ? 0u : d.base - 1u;
    d.c_hi = oc >= 32u;
    d.c_sh = oc & 31u;
    d.c_mask = (j == 0 || oc >= 64u) ? 0u : (w >= 32u ? ~0u : (1u << w) - 1u);
    uint32_t t = 0;
    for (uint32_t i = 0; i < j; i++) t = (t << (bits & 31u)) + (d.base >> 1);
    d.c_thresh = j == 0 ? ~0u : t;
    return d;
}
template <bool DHI, bool CHI>
__device__ __forceinline__ int32_t sdig32(uint64_t v, const SDig32& d) {
    const uint32_t lo = lo32(v), hi = hi32(v);
    uint32_t piece = (DHI ? hi >> d.d_sh : __builtin_amdgcn_alignbit(hi, lo, d.d_sh)) & d.d_mask;
    const uint32_t low = (CHI ? hi >> d.c_sh : __builtin_amdgcn_alignbit(hi, lo, d.c_sh)) & d.c_mask;
    piece += low > d.c_thresh ? 1u : 0u;
    return (int32_t)(piece - (piece > d.thresh ? d.base : 0u));
}
// residues of a signed digit / digit difference s, |s| < 2^28: a negative s wraps to a huge u32 and s + m back into [0, m)
__device__ __forceinline__ void signed_residues(int32_t s, uint32_t& rp, uint32_t& rb) {
    const uint32_t d = (uint32_t)s;
    rp = min(d, d + kP);
    rb = min(d, d + kB);
}
// The DIFFERENCE of two balanced digits needs no carry logic at all.  Within a chain whose digits all may borrow, d_j = u_j - (B/2 - 1) where
// u_j is the plain base-B digit j of x + bias, x the chain's own bits and bias = sum_j (B/2 - 1) B^j: adding B/2 - 1 to a piece makes it
// overflow into the next digit exactly when the piece (with its carry-in) exceeds B/2, which is the reference's borrow rule
// (src/spiral.cpp:283-292, 313-322; the digits of a number in [-B/2 + 1, B/2] are unique), and the constant cancels in G^-1(H)_k - G^-1(L)_k.
// The first chain's last digit never borrows: it is its plain digit plus the carry out of the biased digits below it.  Per value that is an
// add and a bit-field extract instead of sdig32's two extracts, compare, add, compare and select.  ok: both chains fit 32 bits
// (every even gadget dimension up to 12 among others); the other dimensions keep sdig32.
struct SFast {
    bool ok, chain1, last0, x_hi;
    uint32_t x_sh, bias, d_sh, bits, low_mask;
};
__device__ __forceinline__ SFast sfast_setup(uint32_t k, uint32_t bits, uint32_t ell) {
    SFast f;
    const uint32_t n0 = ell >> 1, n1 = ell - n0, oc = n0 * bits;
    f.ok = bits <= kSmallDigitBits && n0 >= 1 && n0 * bits <= 32u && n1 * bits <= 32u && ell * bits >= 57u;
    f.chain1 = k >= n0;
    const uint32_t j = f.chain1 ? k - n0 : k;
    f.last0 = !f.chain1 && k + 1 == n0;
    const uint32_t nb = f.chain1 ? n1 : n0 - 1u;  // the chain's borrowing digits
    uint32_t b = 0;
    for (uint32_t i = 0; i < nb; i++) b += ((1u << (bits & 31u)) / 2u - 1u) << ((i * bits) & 31u);
    f.bias = b;
    f.x_hi = oc >= 32u;
    f.x_sh = oc & 31u;
    f.d_sh = (j * bits) & 31u;
    f.bits = bits;
    f.low_mask = (1u << f.d_sh) - 1u;
    return f;
}
// x: the chain's bits (the value's low word for the first chain; the value shifted down to the second chain's start)
template <bool LAST0>
__device__ __forceinline__ uint32_t sfast_word(uint32_t x, const SFast& f) {
    if constexpr (LAST0) return __builtin_amdgcn_ubfe(x, f.d_sh, f.bits) + (((x & f.low_mask) + f.bias) >> f.d_sh);
    return __builtin_amdgcn_ubfe(x + f.bias, f.d_sh, f.bits);  // v_bfe_u32: d_sh + bits <= 32
}
template <bool CHAIN1, bool XHI, bool LAST0>
__device__ __forceinline__ uint32_t sfast_digit(uint64_t v, const SFast& f) {
    return sfast_word<LAST0>(CHAIN1 ? (XHI ? hi32(v) >> f.x_sh : __builtin_amdgcn_alignbit(hi32(v), lo32(v), f.x_sh)) : lo32(v), f);
}
// digit differences G^-1(h)_k - G^-1(l)_k of a thread's 8 pairs of lifted coefficients as residues (the pair form of a fold
// round); H(r), L(r) fetch the values (registers or LDS)
template <class FH, class FL>
__device__ __forceinline__ void sdigit_diff8(FH H, FL L, uint32_t k, uint32_t bits, uint32_t ell, uint32_t* lo, uint32_t* hi) {
    const SFast f = sfast_setup(k, bits, ell);
    if (f.ok) {
        if (f.last0) {
#pragma unroll
            for (int r = 0; r < 8; r++) signed_residues((int32_t)(sfast_digit<false, false, true>(H(r), f) - sfast_digit<false, false, true>(L(r), f)), lo[r], hi[r]);
        } else if (!f.chain1) {
#pragma unroll
            for (int r = 0; r < 8; r++) signed_residues((int32_t)(sfast_digit<false, false, false>(H(r), f) - sfast_digit<false, false, false>(L(r), f)), lo[r], hi[r]);
        } else if (f.x_hi) {
#pragma unroll
            for (int r = 0; r < 8; r++) signed_residues((int32_t)(sfast_digit<true, true, false>(H(r), f) - sfast_digit<true, true, false>(L(r), f)), lo[r], hi[r]);
        } else {
#pragma unroll
            for (int r = 0; r < 8; r++) signed_residues((int32_t)(sfast_digit<true, false, false>(H(r), f) - sfast_digit<true, false, false>(L(r), f)), lo[r], hi[r]);
        }
        return;
    }
    const SDig32 d = sdig32_setup(k, bits, ell);
    if (d.ok) {
        if (!d.d_hi) {
#pragma unroll
            for (int r = 0; r < 8; r++) signed_residues(sdig32<false, false>(H(r), d) - sdig32<false, false>(L(r), d), lo[r], hi[r]);
        } else if (!d.c_hi) {
#pragma unroll
            for (int r = 0; r < 8; r++) signed_residues(sdig32<true, false>(H(r), d) - sdig32<true, false>(L(r), d), lo[r], hi[r]);
        } else {
#pragma unroll
            for (int r = 0; r < 8; r++) signed_residues(sdig32<true, true>(H(r), d) - sdig32<true, true>(L(r), d), lo[r], hi[r]);
        }
    } else {  // wide digits or long chains: sdigit_of leaves residues below 2^28, which may exceed b
        const SDigit sd = sdigit_setup(k, bits, ell);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            uint32_t ap, ab, bp, bb;
            sdigit_of<false>(H(r), sd, ap, ab);
            sdigit_of<false>(L(r), sd, bp, bb);
            const uint32_t dp = csub_min(ap, kP) - csub_min(bp, kP), db = csub_min(ab, kB) - csub_min(bb, kB);
            lo[r] = min(dp, dp + kP);
            hi[r] = min(db, db + kB);
        }
    }
}

}  // namespace spiral
