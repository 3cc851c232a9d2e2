// Batched NTT / INTT kernels with fused pre- and post-operations, gfx950.
// One 256-thread workgroup per output polynomial.  See ntt_device.h for the transform itself.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "digits_device.h"
#include "kernels.h"
#include "ntt_device.h"

namespace spiral {

// a thread's 8 slots of a PK polynomial as residues; reduce: the fields may be lazy sums (< 2^32: the output of a reduce over
// ranks), otherwise they are canonical already
__device__ __forceinline__ void pk_load8_red(const uint64_t* poly, bool reduce, uint32_t tid, uint32_t* lo, uint32_t* hi) {
    uint64_t x[8];
    pk_load8(poly, tid, x);
    pk_unpack8(x, lo, hi);
    if (reduce) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            lo[r] %= kP;
            hi[r] %= kB;
        }
    }
}

template <uint32_t LOAD, uint32_t STORE, class L>  // L: Lanes (a batch: gridDim.z lanes) or NoLanes (one query: no lane arguments at all, kernels.h)
__global__ __launch_bounds__(256, 8) void ntt_forward_kernel(Tables t, FwdParamsT<L> p) {
    __shared__ uint64_t sh[kLdsWords];
    const uint32_t tid = threadIdx.x;
    uint32_t b = blockIdx.x;
    if constexpr (STORE == ST_PK) {  // query lane blockIdx.z (kernels.h Lanes); the database loaders never take lanes
        const int64_t lane = p.lanes.here();
        lane_shift(p.src, lane);
        lane_shift(p.dst, lane);
    }
    // the ell digit jobs of a polynomial pair read the same 32 KiB of lifted coefficients: consecutive JOBS (not block ids, which are dealt
    // round-robin to the 8 XCDs) go to one XCD, so that its L2 serves the re-reads (-3.5 us on the fold of config 2; the one-source digit
    // launches measured +-0 with the same map in round 2 and +5 us on expand + convert now: they keep the plain order)
    if constexpr (LOAD == LD_SDIFF || LOAD == LD_PDIFF)
        if ((gridDim.x & 7u) == 0) b = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    uint32_t s = udiv_small(b, p.n_digits, p.inv_n_digits), k = b - s * p.n_digits;
    uint32_t lo[8], hi[8];

    if constexpr (LOAD == LD_DBGEN) {
        // block b <-> (item, polynomial mc) ; coefficient index within the item = mc*N + z
        const uint64_t item = p.item_base + (b >> 2);
        const uint32_t mc = b & 3u;
        const uint64_t half_p = p.p_db >> 1;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            uint32_t idx = ix_a(tid, r);
            uint64_t v = p.items ? packed_coeff(p.items, (item - p.items_first) * (4ull * kN) + (uint64_t)mc * kN + idx, p.coeff_bits)
                                 : splitmix64(p.seed ^ (item * (4ull * kN) + (uint64_t)mc * kN + idx)) % p.p_db;
            if (v >= p.p_db) {  // only an ingested coefficient can be
                *p.err = 1u;
                v %= p.p_db;
            }
            if (v >= half_p) {  // v - p_db + Q  ==  -(p_db - v) mod m
                uint64_t d = p.p_db - v;
                lo[r] = kP - mod_p(d);
                hi[r] = kB - mod_b(d);
            } else {
                lo[r] = mod_p(v);
                hi[r] = mod_b(v);
            }
        }
    } else if constexpr (LOAD == LD_DBGEN1) {
        const uint64_t item = p.item_base + b;
        const uint64_t half_p = p.p_db >> 1;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            uint32_t idx = ix_a(tid, r);
            uint64_t v = p.items ? packed_coeff(p.items, (item - p.items_first) * (uint64_t)kN + idx, p.coeff_bits)
                                 : splitmix64(p.seed ^ (((uint64_t)p.trial * p.total_n + item) * kN + idx)) % p.p_db;
            if (v >= p.p_db) {
                *p.err = 1u;
                v %= p.p_db;
            }
            if (v >= half_p) {
                uint64_t d = p.p_db - v;
                lo[r] = kP - mod_p(d);
                hi[r] = kB - mod_b(d);
            } else {
                lo[r] = mod_p(v);
                hi[r] = mod_b(v);
            }
        }
    } else if constexpr (LOAD == LD_PDIGIT) {
        uint32_t sp;  // source polynomial
        if (p.pmode == PM_GSW) {
            sp = s;
        } else if (p.pmode == PM_FOLD) {
            const uint32_t per_t = 4u * p.fold_np, t = s / per_t, rem = s - t * per_t;
            sp = (t * p.pk_num_per + (rem >> 1)) * 2u + (rem & 1u);
        } else {
            sp = s * p.pk_num_per * 2u;
        }
        const uint64_t* src = p.src + (size_t)sp * kN;
        const uint64_t mask = (1ull << p.bits) - 1;
        uint64_t raw[8];  // all eight requested before the digit-width branch (inside it, the compiler waits for the first alone)
#pragma unroll
        for (int r = 0; r < 8; r++) raw[r] = src[ix_a(tid, r)];
        if (p.bits <= kSmallDigitBits) {
            udigits8(raw, k, p.bits, lo);
#pragma unroll
            for (int r = 0; r < 8; r++) hi[r] = lo[r];
        } else {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                uint32_t d = (uint32_t)digit_of(raw[r], k, p.bits, mask);
                lo[r] = digit_residue<false>(d, kP);
                hi[r] = digit_residue<false>(d, kB);
            }
        }
    } else if constexpr (LOAD == LD_EXPAND) {
        // job b -> (active ct a, digit k of automorph(c)[0])
        const uint32_t je = p.cnt_e * p.t_e;
        uint32_t a, tdim;
        if (b < je) {
            tdim = p.t_e;
            a = udiv_small(b, tdim, p.inv_te);
            k = b - a * tdim;
        } else {
            tdim = p.t_o;
            const uint32_t bb = b - je;
            a = udiv_small(bb, tdim, p.inv_to);
            k = bb - a * tdim;
            a += p.cnt_e;
        }
        const uint64_t* src = p.src + (size_t)a * 2u * kN;
        uint64_t raw[8];  // requested before anything else is computed (see LD_PDIGIT)
#pragma unroll
        for (int r = 0; r < 8; r++) raw[r] = src[ix_a(tid, r)];
        const uint32_t bits = get_bits_per(tdim);
        if (bits <= kSmallDigitBits) {  // (the source is already automorphed by the inverse pass)
            udigits8(raw, k, bits, lo);
#pragma unroll
            for (int r = 0; r < 8; r++) hi[r] = lo[r];
        } else {
            const uint64_t mask = (1ull << bits) - 1;
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const uint32_t d = (uint32_t)digit_of(raw[r], k, bits, mask);
                lo[r] = digit_residue<false>(d, kP);
                hi[r] = digit_residue<false>(d, kB);
            }
        }
        s = b;
    } else if constexpr (LOAD == LD_SDIFF) {
        // Pair form of a fold round (src/spiral.cpp:1349-1383 through an identity).  The reference folds L = C[i] and H = C[np + i] into
        // Q_neg G^-1(L) + Q G^-1(H) with Q_neg = G2 - Q slot by slot (:2361-2379); in exact arithmetic mod p and mod b that is
        // G2 G^-1(L) + Q (G^-1(H) - G^-1(L)), and G2 G^-1(L) recomposes L whenever split_and_crt's balanced digits (:270-330) sum back to
        // the value (kernels.h fold_pair_exact).  Hence out[i] = L + Q NTT(G^-1(H) - G^-1(L)): half the forward transforms, half the operand.
        // Source s = (pair i, r, c) over [np][3][2] of the lifted ciphertexts, L = raw[i][r][c], H = raw[np + i][r][c]; the digit
        // difference G^-1(H)_k - G^-1(L)_k (an integer in (-1.5 B, 1.5 B)) as residues
        const uint32_t i = s / 6u, rc = s - i * 6u;
        const uint64_t* sl = p.src + ((size_t)i * 6u + rc) * kN;
        const uint64_t* sh_ = p.src + ((size_t)(p.fold_np + i) * 6u + rc) * kN;
        const SFast f = sfast_setup(k, p.bits, p.ell);
        if (f.ok && (!f.chain1 || f.x_hi)) {
            // the digit lives in ONE 32-bit word of the value (the first chain in the low word; the second chain, when it starts at or beyond
            // bit 32, in the high word): half the loads and half the registers of the general form
            const uint32_t* wl = reinterpret_cast<const uint32_t*>(sl) + (f.chain1 ? 1u : 0u);
            const uint32_t* wh = reinterpret_cast<const uint32_t*>(sh_) + (f.chain1 ? 1u : 0u);
            uint32_t xl[8], xh[8];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                xl[r] = wl[2u * ix_a(tid, r)];
                xh[r] = wh[2u * ix_a(tid, r)];
            }
            const uint32_t xs = f.chain1 ? f.x_sh : 0u;
            if (f.last0) {
#pragma unroll
                for (int r = 0; r < 8; r++) signed_residues((int32_t)(sfast_word<true>(xh[r], f) - sfast_word<true>(xl[r], f)), lo[r], hi[r]);
            } else if (xs == 0u) {  // (the chain starts at the word's bit 0: the first chain always, the second when it starts at bit 32)
#pragma unroll
                for (int r = 0; r < 8; r++) signed_residues((int32_t)(sfast_word<false>(xh[r], f) - sfast_word<false>(xl[r], f)), lo[r], hi[r]);
            } else {
#pragma unroll
                for (int r = 0; r < 8; r++) signed_residues((int32_t)(sfast_word<false>(xh[r] >> xs, f) - sfast_word<false>(xl[r] >> xs, f)), lo[r], hi[r]);
            }
        } else {
            uint64_t rl[8], rh[8];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                rl[r] = sl[ix_a(tid, r)];
                rh[r] = sh_[ix_a(tid, r)];
            }
            sdigit_diff8([&](int r) { return rh[r]; }, [&](int r) { return rl[r]; }, k, p.bits, p.ell, lo, hi);
        }
    } else if constexpr (LOAD == LD_PDIFF) {
        // SpiralPack fold round in pair form: source s = (trial t, pair i, row) over [nt][np][2]; lifted ciphertexts [t][2 np][2] at p.src;
        // unsigned digits (gadget_invert, src/util.cpp:114), whose base-2^bits expansion always recomposes the value
        const uint32_t ti = s >> 1, row = s & 1u, t = ti / p.fold_np, i = ti - t * p.fold_np;
        const uint64_t* sl = p.src + ((size_t)(t * 2u * p.fold_np + i) * 2u + row) * kN;
        const uint64_t* sh_ = p.src + ((size_t)(t * 2u * p.fold_np + p.fold_np + i) * 2u + row) * kN;
        uint64_t rl[8], rh[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            rl[r] = sl[ix_a(tid, r)];
            rh[r] = sh_[ix_a(tid, r)];
        }
        const uint64_t mask = (1ull << p.bits) - 1;
        if (p.bits <= kSmallDigitBits) {
#pragma unroll
            for (int r = 0; r < 8; r++)
                signed_residues((int32_t)(uint32_t)digit_of(rh[r], k, p.bits, mask) - (int32_t)(uint32_t)digit_of(rl[r], k, p.bits, mask), lo[r], hi[r]);
        } else {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const uint32_t dh = (uint32_t)digit_of(rh[r], k, p.bits, mask), dl = (uint32_t)digit_of(rl[r], k, p.bits, mask);
                const uint32_t dp = dh % kP - dl % kP, db = dh % kB - dl % kB;
                lo[r] = min(dp, dp + kP);
                hi[r] = min(db, db + kB);
            }
        }
    } else if constexpr (LOAD == LD_LIMBS) {
        const uint64_t* src = p.src + (size_t)p.src_map(s) * (2 * kN);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            uint32_t idx = ix_a(tid, r);
            lo[r] = csub((uint32_t)src[idx], 2 * kP);  // the reference accepts lazy inputs < 4m here
            hi[r] = csub((uint32_t)src[kN + idx], 2 * kB);
        }
    } else {
        const uint64_t* src = p.src + (size_t)p.src_map(s) * kN;
        const uint64_t mask = (1ull << p.bits) - 1;
        uint64_t raw[8];  // requested before anything else is computed (see LD_PDIGIT)
#pragma unroll
        for (int r = 0; r < 8; r++) raw[r] = load_raw(src, ix_a(tid, r), p.tinv);
        SDigit sd{};
        if constexpr (LOAD == LD_SDIGIT) sd = sdigit_setup(k, p.bits, p.ell);
        auto body = [&](auto small) {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                uint64_t v = raw[r];
                if constexpr (LOAD == LD_RAW) {
                    lo[r] = mod_p(v);
                    hi[r] = mod_b(v);
                } else if constexpr (LOAD == LD_DIGIT) {
                    uint32_t d = (uint32_t)digit_of(v, k, p.bits, mask);
                    lo[r] = digit_residue<decltype(small)::value>(d, kP);
                    hi[r] = digit_residue<decltype(small)::value>(d, kB);
                } else {
                    sdigit_of<decltype(small)::value>(v, sd, lo[r], hi[r]);
                }
            }
        };
        if constexpr (LOAD == LD_RAW) {
            body(std::false_type{});
        } else if constexpr (LOAD == LD_DIGIT) {
            if (p.bits <= kSmallDigitBits) {  // the word choice hoisted out of the loop (udigits8)
                udigits8(raw, k, p.bits, lo);
#pragma unroll
                for (int r = 0; r < 8; r++) hi[r] = lo[r];
            } else {
                body(std::false_type{});
            }
        } else {
            DIGIT_WIDTH_DISPATCH(p.bits, body);
        }
    }

    ntt_forward_block<false>(lo, hi, sh, t.fwd, tid);
    // the expansion's digit transforms feed only expand_mac_round_* (at most 56 + 56 products per accumulator): left in [0, 2m)
    bool lazy = LOAD == LD_EXPAND;
    if constexpr (STORE == ST_PK && LOAD != LD_EXPAND) lazy = p.lazy_out != 0;
    if (!lazy) canonicalize8(lo, hi);

#ifdef NTT_ABLATE_STORE
    if (lo[0] != 0x12345u) return;
#endif
    if constexpr (STORE == ST_PK) {
        uint32_t di;
        if constexpr (LOAD == LD_PDIGIT) {
            const uint32_t nd = p.n_digits;
            if (p.pmode == PM_GSW) {
                di = (s >> 1) * (2u * nd) + 2u * k + (s & 1u);
            } else if (p.pmode == PM_FOLD) {
                const uint32_t per_t = 4u * p.fold_np, t = s / per_t, rem = s - t * per_t, ip = rem >> 1, row = rem & 1u;
                const uint32_t hh = ip / p.fold_np, i = ip - hh * p.fold_np;
                di = ((t * p.fold_np + i) * 2u + hh) * (2u * nd) + 2u * k + row;
            } else {
                di = s * nd + k;
            }
        } else if constexpr (LOAD == LD_PDIFF) {
            di = (s >> 1) * (2u * p.n_digits) + 2u * k + (s & 1u);  // operand layout D'[t][i][row + 2k]
        } else if constexpr (LOAD == LD_SDIFF) {
            const uint32_t i = s / 6u, rc = s - i * 6u;  // operand layout D'[i][r + 3k][c]
            di = (i * 3u * p.ell + (rc >> 1) + 3u * k) * 2u + (rc & 1u);
        } else if constexpr (LOAD == LD_SDIGIT) {
            // source s = (ct i', r, c) over [2*np][3][2]; operand layout D[i' % np][(i' / np)*m2 + r + 3k][c]
            const uint32_t ct = s / 6u, rc = s - ct * 6u, r = rc >> 1, c = rc & 1u;
            const uint32_t m2 = 3u * p.ell, hi_half = ct / p.fold_np, i = ct - hi_half * p.fold_np;
            di = ((i * 2u + hi_half) * m2 + r + 3u * k) * 2u + c;
        } else {
            di = p.dst_map(b);
        }
        uint64_t v[8];
        pk_pack8(lo, hi, v);
#ifdef FWD_NT_STORE
        pk_store8<true>(p.dst + (size_t)di * kN, tid, v);
#else
        pk_store8(p.dst + (size_t)di * kN, tid, v);
#endif
    } else if constexpr (STORE == ST_REF) {
        uint64_t* dst = p.dst + (size_t)p.dst_map(b) * (2 * kN) + 8u * tid;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            dst[r] = lo[r];
            dst[kN + r] = hi[r];
        }
    } else if constexpr (STORE == ST_DB1) {
        const uint64_t item = p.item_base + b;
        const uint32_t ii = (uint32_t)(item % p.num_per), j = (uint32_t)(item / p.num_per);
#pragma unroll
        for (int r = 0; r < 8; r++) db1_put_word(p.dst, pk_pos_tk(tid, r), j, ii, p.num_per, p.dim0_shard, pack(lo[r], hi[r]));  // z = pk_pos(slot)
    } else {  // ST_DB: scatter into the sweep layout
        const uint64_t item = p.item_base + (b >> 2);
        const uint32_t mc = b & 3u, m = mc >> 1, c = mc & 1u;
        const uint32_t ii = (uint32_t)(item % p.num_per), j = (uint32_t)(item / p.num_per);
        const uint32_t ic = ii * 2u + c, nic = 2u * p.num_per;
#pragma unroll
        for (int r = 0; r < 8; r++) db_put_word(p.dst, pk_pos_tk(tid, r), j - p.j0, ic, m, nic, p.dim0_shard, pack(lo[r], hi[r]));  // z = pk_pos(slot)
    }
}

// Two gadget digits of one source polynomial per workgroup (LD_DIGIT / LD_EXPAND / LD_SDIFF / LD_PDIFF, ST_PK): the source is read once for both and
// the two forward transforms share every twiddle fetch (ntt_forward_block2).  Job b2 = (source, digit pair kk): digits 2kk and
// 2kk + 1 (the second absent when the digit count is odd); destinations and results exactly those of ntt_forward_kernel.
// LD_SDIFF (round 6): the fold's digit-difference transforms, two digits of one polynomial PAIR per workgroup -- the wide rounds of a batch
// are thousands of such transforms per launch, where the shared twiddle fetch is worth what it is for the expansion's digits.
template <uint32_t LOAD, class L>
__global__ __launch_bounds__(256) void ntt_forward2_kernel(Tables t, FwdParamsT<L> p) {
    __shared__ uint64_t sh[2][kLdsWords];
    const uint32_t tid = threadIdx.x;
    uint32_t b2 = blockIdx.x;
    {
        const int64_t lane = p.lanes.here();
        lane_shift(p.src, lane);
        lane_shift(p.dst, lane);
    }
    if constexpr (LOAD == LD_SDIFF) {
        // the digit-pair jobs of a polynomial pair re-read the same 32 KiB: consecutive jobs on one XCD (as ntt_forward_kernel does)
        if ((gridDim.x & 7u) == 0) b2 = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
        const uint32_t tdim = p.n_digits, jp = (tdim + 1u) >> 1, s = b2 / jp, kk = b2 - s * jp, k0 = 2u * kk, k1 = k0 + 1u;
        const bool two = k1 < tdim;
        const uint32_t i = s / 6u, rc = s - i * 6u;  // source s = (pair i, r, c): L = raw[i][r][c], H = raw[np + i][r][c]
        const uint64_t* sl = p.src + ((size_t)i * 6u + rc) * kN;
        const uint64_t* sh_ = p.src + ((size_t)(p.fold_np + i) * 6u + rc) * kN;
        uint64_t rl[8], rh[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            rl[r] = sl[ix_a(tid, r)];
            rh[r] = sh_[ix_a(tid, r)];
        }
        uint32_t lo0[8], hi0[8], lo1[8], hi1[8];
        sdigit_diff8([&](int r) { return rh[r]; }, [&](int r) { return rl[r]; }, k0, p.bits, p.ell, lo0, hi0);
        sdigit_diff8([&](int r) { return rh[r]; }, [&](int r) { return rl[r]; }, two ? k1 : k0, p.bits, p.ell, lo1, hi1);
        ntt_forward_block2<false>(lo0, hi0, lo1, hi1, sh[0], sh[1], t.fwd, tid);
        if (!p.lazy_out) {
            canonicalize8(lo0, hi0);
            canonicalize8(lo1, hi1);
        }
        const uint32_t d0 = (i * 3u * p.ell + (rc >> 1) + 3u * k0) * 2u + (rc & 1u);  // operand layout D'[i][r + 3k][c]
        uint64_t v[8];
        pk_pack8(lo0, hi0, v);
        pk_store8(p.dst + (size_t)d0 * kN, tid, v);
        if (two) {
            pk_pack8(lo1, hi1, v);
            pk_store8(p.dst + (size_t)(d0 + 6u) * kN, tid, v);  // digit k + 1: three rows x two columns further
        }
        return;
    }
    if constexpr (LOAD == LD_PDIFF) {
        // SpiralPack's fold in pair form (foldCiphertextsDim1): two unsigned digit differences of one polynomial pair; source s = (trial t, pair i, row)
        if ((gridDim.x & 7u) == 0) b2 = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
        const uint32_t tdim = p.n_digits, jp = (tdim + 1u) >> 1, s = b2 / jp, kk = b2 - s * jp, k0 = 2u * kk, k1 = k0 + 1u;
        const bool two = k1 < tdim;
        const uint32_t ti = s >> 1, row = s & 1u, tr = ti / p.fold_np, i = ti - tr * p.fold_np;
        const uint64_t* sl = p.src + ((size_t)(tr * 2u * p.fold_np + i) * 2u + row) * kN;
        const uint64_t* sh_ = p.src + ((size_t)(tr * 2u * p.fold_np + p.fold_np + i) * 2u + row) * kN;
        uint64_t rl[8], rh[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            rl[r] = sl[ix_a(tid, r)];
            rh[r] = sh_[ix_a(tid, r)];
        }
        const uint64_t mask = (1ull << p.bits) - 1;
        uint32_t lo0[8], hi0[8], lo1[8], hi1[8];
        auto diff = [&](uint32_t k, uint32_t* lo, uint32_t* hi) {
            if (p.bits <= kSmallDigitBits) {
#pragma unroll
                for (int r = 0; r < 8; r++)
                    signed_residues((int32_t)(uint32_t)digit_of(rh[r], k, p.bits, mask) - (int32_t)(uint32_t)digit_of(rl[r], k, p.bits, mask), lo[r], hi[r]);
            } else {
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const uint32_t dh = (uint32_t)digit_of(rh[r], k, p.bits, mask), dl = (uint32_t)digit_of(rl[r], k, p.bits, mask);
                    const uint32_t dp = dh % kP - dl % kP, db = dh % kB - dl % kB;
                    lo[r] = min(dp, dp + kP);
                    hi[r] = min(db, db + kB);
                }
            }
        };
        diff(k0, lo0, hi0);
        diff(two ? k1 : k0, lo1, hi1);
        ntt_forward_block2<false>(lo0, hi0, lo1, hi1, sh[0], sh[1], t.fwd, tid);
        if (!p.lazy_out) {
            canonicalize8(lo0, hi0);
            canonicalize8(lo1, hi1);
        }
        const uint32_t d0 = (s >> 1) * (2u * p.n_digits) + 2u * k0 + (s & 1u);  // operand layout D'[t][i][row + 2k]
        uint64_t v[8];
        pk_pack8(lo0, hi0, v);
        pk_store8(p.dst + (size_t)d0 * kN, tid, v);
        if (two) {
            pk_pack8(lo1, hi1, v);
            pk_store8(p.dst + (size_t)(d0 + 2u) * kN, tid, v);
        }
        return;
    }
    uint32_t tdim, kk, ob;  // digits per source, pair index, job id of digit 0 of this source in ntt_forward_kernel's numbering
    const uint64_t* src;
    uint32_t bits;
    if constexpr (LOAD == LD_EXPAND) {
        const uint32_t jpe = (p.t_e + 1u) >> 1, jpo = (p.t_o + 1u) >> 1, je2 = p.cnt_e * jpe;
        uint32_t a;
        if (b2 < je2) {
            tdim = p.t_e;
            a = b2 / jpe;
            kk = b2 - a * jpe;
            ob = a * tdim;
        } else {
            tdim = p.t_o;
            const uint32_t bb = b2 - je2;
            a = bb / jpo;
            kk = bb - a * jpo;
            ob = p.cnt_e * p.t_e + a * tdim;
            a += p.cnt_e;
        }
        src = p.src + (size_t)a * 2u * kN;
        bits = get_bits_per(tdim);
    } else {
        tdim = p.n_digits;
        const uint32_t jp = (tdim + 1u) >> 1, s = b2 / jp;
        kk = b2 - s * jp;
        ob = s * tdim;
        src = p.src + (size_t)p.src_map(s) * kN;
        bits = p.bits;
    }
    uint64_t raw[8];
#pragma unroll
    for (int r = 0; r < 8; r++) raw[r] = src[ix_a(tid, r)];
    const uint32_t k0 = 2u * kk, k1 = k0 + 1u;
    const bool two = k1 < tdim;
    const uint64_t mask = (1ull << bits) - 1;
    uint32_t lo0[8], hi0[8], lo1[8], hi1[8];
    auto body = [&](auto small) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t d0 = (uint32_t)digit_of(raw[r], k0, bits, mask), d1 = (uint32_t)digit_of(raw[r], k1, bits, mask);
            lo0[r] = digit_residue<decltype(small)::value>(d0, kP);
            hi0[r] = digit_residue<decltype(small)::value>(d0, kB);
            lo1[r] = digit_residue<decltype(small)::value>(d1, kP);
            hi1[r] = digit_residue<decltype(small)::value>(d1, kB);
        }
    };
    DIGIT_WIDTH_DISPATCH(bits, body);
    ntt_forward_block2<false>(lo0, hi0, lo1, hi1, sh[0], sh[1], t.fwd, tid);
    const bool lazy = LOAD == LD_EXPAND || p.lazy_out != 0;
    if (!lazy) {
        canonicalize8(lo0, hi0);
        canonicalize8(lo1, hi1);
    }
    uint64_t v[8];
    pk_pack8(lo0, hi0, v);
    pk_store8(p.dst + (size_t)p.dst_map(ob + k0) * kN, tid, v);
    if (two) {
        pk_pack8(lo1, hi1, v);
        pk_store8(p.dst + (size_t)p.dst_map(ob + k1) * kN, tid, v);
    }
}

// (8 workgroups per CU: the second launch bound keeps the kernel at 64 VGPRs, where the compiler's own choice was 65)
template <uint32_t STORE, bool EXPAND, class L>
__global__ __launch_bounds__(256, 8) void ntt_inverse_kernel(Tables t, InvParamsT<L> p) {
    __shared__ uint64_t sh[kLdsWords];
    const uint32_t tid = threadIdx.x, b = blockIdx.x;
    {
        const int64_t lane = p.lanes.here();
        lane_shift(p.src, lane);
        lane_shift(p.dst, lane);
        if constexpr (EXPAND) {
            lane_shift(p.cv, lane);
            lane_shift(p.query, lane);
        }
    }
    uint32_t lo[8], hi[8];
    if constexpr (EXPAND) {
        const uint32_t a = b >> 1, row = b & 1u;
        const uint32_t i = p.act.index(a, p.cnt_e);
#ifdef EXP_ABL_NO_ROW1
        if (row == 1) return;
#endif
#ifdef EXP_ABL_NO_CREATE
        const bool created = false;
#else
        // cv[i] = neg1 * cv[i - num_in] (src/spiral.cpp:1709) is created here in round 0 only; later rounds find it already
        // written by the previous round's MAC, which has the new cv[i - num_in] in registers
        const bool created = p.create_here && i >= p.num_in;
#endif
        // round 0 may take the query ciphertext from its own buffer; cv[0] is then written here as well
        const bool from_query = p.create_here && p.query != nullptr;
        const uint64_t* src = (from_query ? p.query : p.cv) + ((size_t)(created ? i - p.num_in : i) * 2u + row) * kN;
        uint64_t* dstc = p.cv + ((size_t)i * 2u + row) * kN;
        uint64_t v[8];
        if (created || row == 0 || from_query) pk_load8(src, tid, v);
        if (from_query && !created) pk_store8(dstc, tid, v);
        if (created) {
            uint64_t w[8], ws[8];
            pk_load8(p.neg1, tid, w);
            pk_load8(p.neg1s, tid, ws);
#pragma unroll
            for (int r = 0; r < 8; r++) {
                lo[r] = csub(shoup(lo32(v[r]), lo32(w[r]), lo32(ws[r]), kP), kP);
                hi[r] = csub(shoup(hi32(v[r]), hi32(w[r]), hi32(ws[r]), kB), kB);
                v[r] = pack(lo[r], hi[r]);
            }
            pk_store8(dstc, tid, v);
        } else if (row == 0) {
            pk_unpack8(v, lo, hi);
        }
        if (row == 1) {
            // Row 1 only needs NTT(automorph(c_1)), and in the transform domain the automorphism x -> x^t is a slot
            // permutation: slot s evaluates at psi^(2 brev(s) + 1), so out[s] = in[s'] with 2 brev(s') + 1 =
            // (2 brev(s) + 1) t mod 2N.  The residues are the same canonical values the reference gets by going
            // through the coefficient domain (src/spiral.cpp:1713-1720), without the two transforms.
            uint64_t o[8];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const uint32_t e = ((2u * (__brev(8u * tid + r) >> 21) + 1u) * p.auto_t) & (2u * kN - 1u);
                const uint32_t pp = pk_pos(__brev(e >> 1) >> 21);
                uint64_t x = src[pp];
                if (created) {
                    const uint64_t w = p.neg1[pp], ws = p.neg1s[pp];
                    x = pack(csub(shoup(lo32(x), lo32(w), lo32(ws), kP), kP), csub(shoup(hi32(x), hi32(w), hi32(ws), kB), kB));
                }
                o[r] = x;
            }
            pk_store8(p.dst + (size_t)b * kN, tid, o);
            return;
        }
    } else if (p.src_ref) {
        const uint64_t* src = p.src + (size_t)p.src_map(b) * (2 * kN) + 8u * tid;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            lo[r] = (uint32_t)src[r];
            hi[r] = (uint32_t)src[kN + r];
        }
    } else {
        const uint32_t sp = (p.split && b >= p.split) ? p.src_map2(b - p.split) : p.src_map(b);
        pk_load8_red(p.src + (size_t)sp * kN, false, tid, lo, hi);
    }
    if (p.pre_reduce) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            lo[r] %= kP;
            hi[r] %= kB;
        }
    }
    constexpr bool kLift = EXPAND || STORE == IST_CRT;  // the CRT lift takes the transform's lazy outputs (crt_compose_lazy)
    ntt_inverse_block<!kLift>(lo, hi, sh, t.inv, tid);
    if constexpr (EXPAND) {
        // store the automorphed polynomial a(x^t) (scatter form of src/poly.cpp:240-261: coefficient i goes to i*t mod N,
        // negated as Q - a when i*t mod 2N >= N), so that the t_exp .. t_exp_right digit transforms that follow read it
        // in order instead of each gathering it
        // (through LDS, so that the global stores are four contiguous 16-byte pieces per thread instead of 8 scattered words)
        __syncthreads();  // the transform's last LDS reads
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t e = (ix_a(tid, r) * p.auto_t) & (2u * kN - 1u);
            const uint64_t v = crt_compose_lazy(csub_min(lo[r], kP), hi[r]);
            sh[lds_ix(e & (kN - 1u))] = (e & kN) ? kQ - v : v;
        }
        __syncthreads();
        pk_u64x2* dst = reinterpret_cast<pk_u64x2*>(p.dst + (size_t)b * kN) + tid;
#pragma unroll
        for (int q = 0; q < 4; q++) dst[q * 256] = pk_u64x2{sh[lds_ix(512u * q + 2u * tid)], sh[lds_ix(512u * q + 2u * tid + 1u)]};
    } else if constexpr (STORE == IST_CRT) {
        uint64_t* dst = p.dst + (size_t)p.dst_map(b) * kN;
#pragma unroll
        for (int r = 0; r < 8; r++) dst[ix_a(tid, r)] = crt_compose_lazy(csub_min(lo[r], kP), hi[r]);
    } else {
        uint64_t* dst = p.dst + (size_t)p.dst_map(b) * (2 * kN);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            dst[ix_a(tid, r)] = lo[r];
            dst[kN + ix_a(tid, r)] = hi[r];
        }
    }
}

// Fold chain: the inverse transform + CRT lift of a PK polynomial (a first-dimension accumulator, or the previous fold
// round's product) followed in registers by the balanced digits and forward transforms the next fold round consumes
// (nttInvAndCrtLiftCiphertexts / from_ntt then split_and_crt, src/spiral.cpp:1349-1410, 270-330).  The inverse leaves
// coefficient tid + 256k in register k, which is exactly what the forward transform wants, so nothing is exchanged and
// the raw polynomial never goes to memory.
// A block takes source polynomial s and digits [k0, k0 + dpb): one inverse transform, then dpb forward transforms.
// dpb = ell: one workgroup per polynomial, no redundant work (rounds that fill the chip); dpb = 1: one workgroup per
// (polynomial, digit), every digit job repeating the inverse transform (the last rounds, which are latency-bound); the host
// picks dpb per round so that a round is about as many blocks as the chip holds.
// (register budget: 103 VGPRs = 4 workgroups per CU.  Forcing 5 or 6 through the launch bound spills 24 / 84 bytes per thread and
// measured 0 / +30 us on the fold, profiles/r03_variants.txt; holding one twiddle row set instead of two does not lower the count.
// Since round 4 this two-product form is the fallback of the pair form (LD_SDIFF): SPIRAL_FOLD_PAIR=0, gadget dimensions whose digits do not recompose.)
template <class L>
__global__ __launch_bounds__(256) void fold_chain_kernel(Tables t, FoldChainParamsT<L> p) {
    __shared__ uint64_t sh[kLdsWords];
    const uint32_t tid = threadIdx.x, b = blockIdx.x;
    {
        const int64_t lane = p.lanes.here();
        lane_shift(p.src, lane);
        lane_shift(p.dst, lane);
    }
    const uint32_t cpp = (p.ell + p.dpb - 1u) / p.dpb;  // chunks per polynomial
    const uint32_t s = b / cpp, k0 = (b - s * cpp) * p.dpb, k1 = min(k0 + p.dpb, p.ell);
    uint32_t lo[8], hi[8];
    // SpiralPack (p.pack): s = (trial t, ct i' < 2np, row) over 2 x 1 ciphertexts, trial stride p.src_stride cts in the source
    const uint32_t per_t = 4u * p.fold_np, pt = s / per_t, prem = s - pt * per_t, pip = prem >> 1, prow = prem & 1u;
    {
        const size_t sp = p.pack ? ((size_t)pt * p.src_stride + pip) * 2u + prow : (size_t)s;
        pk_load8_red(p.src + sp * kN, p.pre_reduce != 0, tid, lo, hi);
    }
    ntt_inverse_block<false>(lo, hi, sh, t.inv, tid);
    uint64_t v[8];
    crt_lift8(lo, hi, v);
    // source s = (ct i', r, c) over [2*np][3][2]; operand layout D[i' % np][(i' / np)*m2 + r + 3k][c]  (as LD_SDIGIT)
    const uint32_t ct = s / 6u, rc = s - ct * 6u, row = rc >> 1, c = rc & 1u;
    const uint32_t m2 = 3u * p.ell, hi_half = ct / p.fold_np, i = ct - hi_half * p.fold_np;
    const uint32_t phh = pip / p.fold_np, pi = pip - phh * p.fold_np;
    const uint64_t mask = (1ull << p.bits) - 1;
    for (uint32_t k = k0; k < k1; k++) {
        auto digits = [&](auto small) {
            if (p.pack) {  // unsigned digits (src/testing.cpp:596-624), operand layout D[t][i' % np][(i' / np)*2ell + row + 2k]
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const uint32_t d = (uint32_t)digit_of(v[r], k, p.bits, mask);
                    lo[r] = digit_residue<decltype(small)::value>(d, kP);
                    hi[r] = digit_residue<decltype(small)::value>(d, kB);
                }
            } else {
                const SDigit sd = sdigit_setup(k, p.bits, p.ell);
#pragma unroll
                for (int r = 0; r < 8; r++) sdigit_of<decltype(small)::value>(v[r], sd, lo[r], hi[r]);
            }
        };
        DIGIT_WIDTH_DISPATCH(p.bits, digits);
        if (k > k0) __syncthreads();  // the previous transform's last LDS reads
        ntt_forward_block<false>(lo, hi, sh, t.fwd, tid);
        if (!p.lazy_out) canonicalize8(lo, hi);
        const size_t di = p.pack ? (size_t)(((pt * p.fold_np + pi) * 2u + phh) * (2u * p.ell) + 2u * k + prow)
                                 : (size_t)(((i * 2u + hi_half) * m2 + row + 3u * k) * 2u + c);
        uint64_t x[8];
        pk_pack8(lo, hi, x);
        pk_store8(p.dst + di * kN, tid, x);
    }
}

__global__ __launch_bounds__(256) void ref_to_pk_kernel(const uint64_t* ref, uint64_t* pk, IndexMap pk_map) {
    const size_t poly = blockIdx.y;
    const uint32_t z = blockIdx.x * 256u + threadIdx.x;
    const uint64_t* r = ref + poly * (2 * kN);
    pk[(size_t)pk_map((uint32_t)poly) * kN + pk_pos(z)] = pack((uint32_t)(r[z] % kP), (uint32_t)(r[kN + z] % kB));
}
__global__ __launch_bounds__(256) void pk_to_ref_kernel(const uint64_t* pk, uint64_t* ref, IndexMap pk_map) {
    const size_t poly = blockIdx.y;
    const uint32_t z = blockIdx.x * 256u + threadIdx.x;
    uint64_t v = pk[(size_t)pk_map((uint32_t)poly) * kN + pk_pos(z)];
    ref[poly * (2 * kN) + z] = lo32(v);
    ref[poly * (2 * kN) + kN + z] = hi32(v);
}

// one query: the instantiation without lane arguments (kernels.h NoLanes); a batch: gridDim.z = the lanes
#define LAUNCH_LANES(KERNEL, GRIDX, TB, P, ...)                                                                                          \
    do {                                                                                                                                 \
        if ((P).lanes.n > 1)                                                                                                             \
            hipLaunchKernelGGL((KERNEL<__VA_ARGS__, Lanes>), dim3((GRIDX), 1, (P).lanes.n), dim3(256), 0, s, TB, P);                     \
        else                                                                                                                             \
            hipLaunchKernelGGL((KERNEL<__VA_ARGS__, NoLanes>), dim3((GRIDX), 1, 1), dim3(256), 0, s, TB, no_lanes(P));                   \
    } while (0)
#define FWD_CASE(L, S)                                                   \
    if (load == L && store == S) {                                       \
        LAUNCH_LANES(ntt_forward_kernel, nblocks, tb, p, L, S);          \
        return;                                                          \
    }

void launch_ntt_forward(const DeviceTables& t, const FwdParams& p_in, uint32_t load, uint32_t store, uint32_t nblocks, hipStream_t s) {
    if (nblocks == 0) return;
    FwdParams p = p_in;
    auto inv = [](uint32_t d) { return d > 1u ? (uint32_t)((1ull << 32) / d + 1ull) : 0u; };  // udiv_small
    p.inv_n_digits = inv(p.n_digits);
    p.inv_te = inv(p.t_e);
    p.inv_to = inv(p.t_o);
    Tables tb{t.fwd, t.inv};
    // Two digits per workgroup on one twiddle fetch (ntt_forward2_kernel) pay in steady state only -- 14-24 % from 16 k transforms up, nothing
    // at one or two generations of resident workgroups (profiles/r04_twiddle_sharing.txt): used from option fwd2_min = 8192 transforms per launch (all
    // query lanes together).  Option "fwd2" = 0 / 1 forces it off / on (read per call: tests, A/B).  Same results either way.
    const int fwd2_opt = options().fwd2;
    const bool fwd2 = (fwd2_opt >= 0 ? fwd2_opt != 0 : (uint64_t)nblocks * p.lanes.n >= options().fwd2_min) && p.tinv == 0;  // (the two-digit loader has no automorphism gather)
    // udiv_small is exact for job indices below 2^32 / d: far above any launch of the server, checked here because the seams take caller sizes
    if ((uint64_t)nblocks * std::max({p.n_digits, p.t_e, p.t_o, 1u}) >= (1ull << 32)) {
        fprintf(stderr, "launch_ntt_forward: %u jobs exceed the range of the job-index division\n", nblocks);
        abort();
    }
    if (fwd2 && store == ST_PK && load == LD_DIGIT && p.n_digits >= 2) {
        const uint32_t nsrc = nblocks / p.n_digits;
        LAUNCH_LANES(ntt_forward2_kernel, nsrc * ((p.n_digits + 1u) / 2u), tb, p, LD_DIGIT);
        return;
    }
    if (fwd2 && store == ST_PK && load == LD_SDIFF && p.n_digits >= 2) {
        const uint32_t nsrc = nblocks / p.n_digits;
        LAUNCH_LANES(ntt_forward2_kernel, nsrc * ((p.n_digits + 1u) / 2u), tb, p, LD_SDIFF);
        return;
    }
    if (fwd2 && store == ST_PK && load == LD_PDIFF && p.n_digits >= 2) {
        const uint32_t nsrc = nblocks / p.n_digits;
        LAUNCH_LANES(ntt_forward2_kernel, nsrc * ((p.n_digits + 1u) / 2u), tb, p, LD_PDIFF);
        return;
    }
    if (fwd2 && store == ST_PK && load == LD_EXPAND) {
        const uint32_t cnt_o = p.t_o ? (nblocks - p.cnt_e * p.t_e) / p.t_o : 0u;
        LAUNCH_LANES(ntt_forward2_kernel, p.cnt_e * ((p.t_e + 1u) / 2u) + cnt_o * ((p.t_o + 1u) / 2u), tb, p, LD_EXPAND);
        return;
    }
    FWD_CASE(LD_RAW, ST_PK)
    FWD_CASE(LD_RAW, ST_REF)
    FWD_CASE(LD_DIGIT, ST_PK)
    FWD_CASE(LD_SDIGIT, ST_PK)
    FWD_CASE(LD_SDIFF, ST_PK)
    FWD_CASE(LD_PDIFF, ST_PK)
    FWD_CASE(LD_LIMBS, ST_REF)
    FWD_CASE(LD_LIMBS, ST_PK)
    FWD_CASE(LD_DBGEN, ST_DB)
    FWD_CASE(LD_EXPAND, ST_PK)
    FWD_CASE(LD_PDIGIT, ST_PK)
    FWD_CASE(LD_DBGEN1, ST_DB1)
    abort();
}

void launch_ntt_inverse_expand(const DeviceTables& t, const InvParams& p, uint32_t nblocks, hipStream_t s) {
    if (nblocks == 0) return;
    Tables tb{t.fwd, t.inv};
    LAUNCH_LANES(ntt_inverse_kernel, nblocks, tb, p, IST_CRT, true);
}

void launch_ntt_inverse(const DeviceTables& t, const InvParams& p, uint32_t store, uint32_t nblocks, hipStream_t s) {
    if (nblocks == 0) return;
    Tables tb{t.fwd, t.inv};
    if (store == IST_CRT)
        LAUNCH_LANES(ntt_inverse_kernel, nblocks, tb, p, IST_CRT, false);
    else
        hipLaunchKernelGGL((ntt_inverse_kernel<IST_LIMBS, false, NoLanes>), dim3(nblocks), dim3(256), 0, s, tb, no_lanes(p));
}

void launch_fold_chain(const DeviceTables& t, const FoldChainParams& p, uint32_t n_src, hipStream_t s) {
    if (n_src == 0) return;
    Tables tb{t.fwd, t.inv};
    if (p.lanes.n > 1)
        hipLaunchKernelGGL(fold_chain_kernel<Lanes>, dim3(n_src * ((p.ell + p.dpb - 1u) / p.dpb), 1, p.lanes.n), dim3(256), 0, s, tb, p);
    else
        hipLaunchKernelGGL(fold_chain_kernel<NoLanes>, dim3(n_src * ((p.ell + p.dpb - 1u) / p.dpb), 1, 1), dim3(256), 0, s, tb, no_lanes(p));
}

void launch_ref_to_pk(const uint64_t* ref, uint64_t* pk, uint32_t npolys, IndexMap pk_map, hipStream_t s) {
    if (npolys) hipLaunchKernelGGL(ref_to_pk_kernel, dim3(kN / 256, npolys), dim3(256), 0, s, ref, pk, pk_map);
}
void launch_pk_to_ref(const uint64_t* pk, uint64_t* ref, uint32_t npolys, IndexMap pk_map, hipStream_t s) {
    if (npolys) hipLaunchKernelGGL(pk_to_ref_kernel, dim3(kN / 256, npolys), dim3(256), 0, s, pk, ref, pk_map);
}

}  // namespace spiral
