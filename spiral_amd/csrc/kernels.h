// Host-callable launchers for the HIP kernels (internal interface between server.cpp and *.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "common.h"

namespace spiral {

// block/poly index map: idx(b) = (b / inner) * outer_stride + (b % inner) + off
struct IndexMap {
    uint32_t inner, outer_stride, off;
    __host__ __device__ uint32_t operator()(uint32_t b) const { return (b / inner) * outer_stride + (b % inner) + off; }
};
inline IndexMap identity_map() { return IndexMap{1u, 1u, 0u}; }

// Environment switches.  The shipped library reads exactly three, all documented in README.md: SPIRAL_FOLD_PAIR, SPIRAL_SWEEP_MFMA and
// SPIRAL_DB_STAGE_BYTES (initial values of the options of spiral_gpu_set_option, include/spiral_gpu.h).  The thresholds that only a tuning
// session moves (tools/build_variants.sh builds with -DSPIRAL_TUNING) are read through tuning_env, which is a constant nullptr otherwise.
#ifdef SPIRAL_TUNING
inline const char* tuning_env(const char* name) { return getenv(name); }
#else
inline const char* tuning_env(const char*) { return nullptr; }
#endif
// Process-wide options (spiral_gpu_set_option).  fwd2: -1 = the two-digits-per-workgroup transform kernel from kFwd2Min transforms per
// launch (ntt.hip), 0 / 1 = never / always.  Same results either way; tests force each form.
struct Options {
    int fold_pair = 1, fold_chain = 1, fwd2 = -1;
    uint32_t fold_blocks = 768, sweep_mfma_min = 2, fwd2_min = 8192;
    size_t db_stage_bytes = (size_t)64 << 20;
    int one_image = 1;  // a server that batches on the matrix cores keeps ONLY the limb-plane image of its database (server.cpp)
};
Options& options();  // server.cpp; the three documented environment variables are read once, on first use

// Query lanes of one launch (server.cpp run_query_batch).  The launch-bound stages of a query -- expansion, conversion, lift, folding: ~50
// dependent launches that cost ~5 us each whatever they carry -- take a QUERY dimension: gridDim.z = n <= kMaxLanes queries, each with its own
// keys, ciphertexts and scratch.  Every per-query buffer of a server lives in one arena with the same internal layout (srv_alloc), so lane q's
// buffer is lane 0's pointer + off[q] words: a kernel shifts every non-table pointer of its parameters by off[blockIdx.z] and is otherwise
// unchanged (n = 1, off = 0: the single-query launch).  The reference answers one query per process_crtd_query (src/spiral.cpp:2337-2406).
#ifndef SPIRAL_MAX_LANES
#define SPIRAL_MAX_LANES 8
#endif
constexpr uint32_t kMaxLanes = SPIRAL_MAX_LANES;  // (= the queries one pass of the matrix-core sweep takes, sweep_mfma.hip)
struct Lanes {
    uint32_t n = 1;
    int64_t off[kMaxLanes] = {};  // u64 words from lane 0's arena to lane q's
#ifdef __HIPCC__
    // (a select chain on constant indices: indexing the by-value kernel argument with blockIdx.z would make the compiler keep the whole
    // parameter struct in scratch memory -- 232 bytes per lane and transforms twice as slow, measured)
    __device__ __forceinline__ int64_t here() const {
        const uint32_t z = blockIdx.z;
        int64_t o = 0;
#pragma unroll
        for (uint32_t q = 1; q < kMaxLanes; q++) o = z == q ? off[q] : o;
        return o;
    }
#endif
};
#ifdef __HIPCC__
template <class T>
__device__ __forceinline__ void lane_shift(T*& p, int64_t words) {  // optional (null) pointers stay null
    if (p) p = reinterpret_cast<T*>(reinterpret_cast<uintptr_t>(p) + (intptr_t)words * 8);
}
#endif
// ONE query's launches carry no lane machinery at all.  A query is ~50 dependent launch-bound launches, and each pays for every instruction of its
// prologue: with the eight offsets in every launch (eight scalar loads and a select chain before the first address is known) one query took 748 us,
// without them 725 (round 6, same box, alternating builds; 64 bytes of UNUSED kernel arguments cost nothing measurable: it is the dependent prologue).  So every kernel that takes lanes is instantiated twice -- `Lanes` for batches, `NoLanes` (no
// offsets, here() == 0, the shifts fold away) for n == 1 -- and every parameter struct is `XCore` (the fields) + `XT<L>` (the fields and an L); the
// host fills an `X = XT<Lanes>` as before and the launcher picks the instantiation (no_lanes() copies the fields).
struct NoLanes {
    uint32_t n = 1;
#ifdef __HIPCC__
    __device__ __forceinline__ int64_t here() const { return 0; }
#endif
};
template <class P>
inline typename P::NoLanesT no_lanes(const P& p) {
    typename P::NoLanesT q{};
    static_cast<typename P::Core&>(q) = static_cast<const typename P::Core&>(p);
    return q;
}

struct DeviceTables {
    uint4* fwd = nullptr;      // [2048] forward twiddles {W_p, W'_p, W_b, W'_b}
    uint4* inv = nullptr;      // [2048] inverse twiddles psi^-i (row 1 times N^-1, row 0 = N^-1: the stages are unscaled, tables.cpp)
    uint64_t* neg1 = nullptr;  // [11][2048] PK: NTT(-x^(N-2^r))   (src/spiral.cpp:171-190)
    uint64_t* neg1s = nullptr; // the same words' Shoup companions floor(w * 2^32 / m), PK
};
// builds the tables on `device` (host computes psi powers, tables.cpp); idempotent per device
int tables_get(int device, DeviceTables* out);
// host copy of the reference-ordered rows (for the C-ABI's spiral_gpu_get_tables)
void tables_host_rows(uint64_t* out8x2048);

// Lazy digit transforms.  A forward transform of gadget digits may leave its outputs in [0, 2m) instead of [0, m) (FwdParams::lazy_out,
// FoldChainParams::lazy_out, always for LD_EXPAND) ONLY when every reader is a u64 multiply-accumulate (poly.hip Acc2, pack.hip)
// against canonical words summing at most `terms` products per accumulator: terms * (2m - 1) * (m - 1) must stay below 2^64.
// Kernels that add or compare such words (add_pk, csub, canonical stores) need canonical operands and must not be fed lazy ones.
// This is the one place the bound is decided; the hosts call it with the number of products their MAC kernel sums.
constexpr bool lazy_ok(uint32_t terms) { return (unsigned __int128)terms * (2ull * kP - 1ull) * (kP - 1ull) < ((unsigned __int128)1 << 64); }
static_assert(lazy_ok(56) && lazy_ok(2 * 56) && lazy_ok(128) && !lazy_ok(129), "u64 accumulators hold up to 128 lazy products");

// ---- forward NTT ---------------------------------------------------------------------------------
enum FwdLoad : uint32_t {
    LD_RAW = 0,     // raw u64 coefficient, reduced mod p / mod b        (to_ntt, src/poly.cpp:311)
    LD_DIGIT = 1,   // unsigned gadget digit k of a raw coefficient      (gadget_invert + to_ntt_no_reduce)
    LD_SDIGIT = 2,  // balanced digit with the fold's carry rules        (split_and_crt, src/spiral.cpp:270)
    LD_LIMBS = 3,   // reference NTT layout [2][N] u64 taken as per-limb coefficient arrays (ntt_forward)
    LD_DBGEN = 4,   // seeded plaintext coefficient, centred lift        (load_db, src/spiral.cpp:1116-1127)
    LD_PDIGIT = 6,  // SpiralPack: unsigned reduced digit k (gadget_invert + to_ntt, src/testing.cpp:130-131, 226-227, 612-617)
                    // with the source / destination maps of FwdParams::pmode
    LD_DBGEN1 = 7,  // SpiralPack database: 1 x 1 plaintext of (trial, item), centred lift (src/testing.cpp:845-869)
    LD_SDIFF = 8,   // fold round in pair form from lifted ciphertexts: difference of the balanced digits k of raw[np + i] and raw[i]
    LD_PDIFF = 9,   // SpiralPack fold round in pair form from lifted ciphertexts [trial][2 np][2]: difference of the unsigned digits k of
                    // ct np + i and ct i, destination D'[trial][i][row + 2k]  (foldCiphertextsDim1 through the same identity as LD_SDIFF)
    LD_EXPAND = 5,  // one expansion round: digits of automorph(c)[0] and the reduced automorph(c)[1] of every
                    // active ciphertext, both parities, in one launch      (src/spiral.cpp:1711-1720)
};
enum FwdStore : uint32_t {
    ST_PK = 0,      // packed slot words
    ST_REF = 1,     // reference layout [2][N] u64
    ST_DB = 2,      // scatter into the device DB layout (see sweep)
    ST_DB1 = 3,     // scatter into the SpiralPack device DB layout (pack.hip)
};
enum PackMap : uint32_t {
    PM_GSW = 0,   // s = (input ct, row); dst = chat[ct][row + 2k]                     (regevToSimpleGsw)
    PM_FOLD = 1,  // s = (trial, ct i' < 2np', row); dst = D[trial][i' % np'][(i' / np') * 2ell + row + 2k]
    PM_PACK = 2,  // s = trial; source = row 0 of the trial's folded ct; dst = ginv[trial][k]   (pack)
};
struct FwdParamsCore {
    const uint64_t* src;
    uint64_t* dst;
    IndexMap src_map;   // s -> source polynomial index
    IndexMap dst_map;   // b -> destination polynomial index (ST_PK / ST_REF)
    uint32_t n_digits;  // b -> (s = b / n_digits, k = b % n_digits)
    uint32_t inv_n_digits, inv_te, inv_to;  // 2^32 / d + 1 for the three job-index divisors (filled in by launch_ntt_forward)
    uint32_t bits;      // digit width
    uint32_t ell;       // LD_SDIGIT: digits per value (t_GSW)
    uint32_t tinv;      // automorphism gather x -> x^t folded into the load: t^-1 mod 2N, 0 = none
    uint32_t lazy_out;  // ST_PK: 1 = leave the outputs in [0, 2m) instead of [0, m): allowed when the only readers are u64
                        // multiply-accumulate kernels summing fewer than 128 products (digit operands); LD_EXPAND always does
    uint32_t fold_np;   // LD_SDIGIT: num_per' (destination is the fold operand layout)
    // LD_PDIGIT: map selector; fold_np = np' and num_per (ct stride of a trial in the raw buffer) for PM_FOLD / PM_PACK
    uint32_t pmode, pk_num_per;
    // LD_DBGEN1: trial and total item count
    uint32_t trial;
    uint64_t total_n;
    // LD_EXPAND: active ct a < cnt_e is even (t_e digits), the rest odd (t_o digits); jobs per ct = t
    uint32_t cnt_e, t_e, t_o;
    // LD_DBGEN / LD_DBGEN1: plaintext coefficients come from the seeded generator, or (items != null) from a staged
    // stream of bit-packed items whose first item is items_first (common.h packed_coeff); err: set to 1 when a
    // coefficient is not below p_db (the reference asserts, src/spiral.cpp:1117)
    const uint8_t* items;
    uint32_t coeff_bits;
    uint64_t items_first;
    uint32_t* err;
    // LD_DBGEN / ST_DB
    uint64_t seed, p_db;
    uint64_t item_base;                 // first item handled by this launch
    uint32_t num_per, dim0_shard, j0;   // DB geometry of this shard
};
template <class L>
struct FwdParamsT : FwdParamsCore {
    using Core = FwdParamsCore;
    using NoLanesT = FwdParamsT<NoLanes>;
    L lanes;                        // src, dst per query lane
};
using FwdParams = FwdParamsT<Lanes>;
void launch_ntt_forward(const DeviceTables& t, const FwdParams& p, uint32_t load, uint32_t store, uint32_t nblocks, hipStream_t s);

// Which ciphertexts of an expansion round a launch works on (src/spiral.cpp:1700-1702 enumerates i < 2^(r+1), odd i only up to
// `stopround`).  Active ciphertext a < cnt_e is the even one i = 2 (a + e_off); the others are odd, i = 2 ((a - cnt_e) *
// (o_stride_m1 + 1) + o_off) + 1.  All zero = every ciphertext of the round (one GPU).  A rank of a G-GPU answer expands only
// what it needs (server.cpp expand shard): the even subtree above its own first-dimension range (a contiguous block of a),
// and of the odd ciphertexts -- the GSW bits -- every G-th.
struct ExpandActive {
    uint32_t e_off, o_stride_m1, o_off;
    __host__ __device__ uint32_t index(uint32_t a, uint32_t cnt_e) const {
        return a < cnt_e ? 2u * (a + e_off) : 2u * ((a - cnt_e) * (o_stride_m1 + 1u) + o_off) + 1u;
    }
};

// ---- inverse NTT ---------------------------------------------------------------------------------
enum InvStore : uint32_t {
    IST_CRT = 0,    // CRT-lifted raw coefficient in [0, Q)   (from_ntt, src/poly.cpp:357)
    IST_LIMBS = 1,  // reference layout [2][N] u64 residues   (ntt_inverse)
};
struct InvParamsCore {
    const uint64_t* src;
    uint64_t* dst;
    IndexMap src_map, dst_map;
    uint32_t pre_reduce;  // 1: fields are lazy sums (< 2^32), reduce mod m first
    uint32_t src_ref;     // 1: source is reference layout [2][N] u64 instead of PK
    uint32_t split;       // > 0: blocks b >= split take their source from src_map2(b - split) (two source regions, one launch)
    IndexMap src_map2;
    // expansion round (launch_ntt_inverse_expand): block b = (active ct a, row); a < cnt_e -> i = 2a, else
    // i = 2(a - cnt_e) + 1; a ct with i >= num_in is first created as neg1 * cv[i - num_in] (src/spiral.cpp:1709)
    // Row 0 is transformed to dst[2a]; row 1 is not: its automorphed image, a slot permutation, goes to dst[2a + 1] in PK.
    uint64_t* cv;
    const uint64_t* neg1;
    const uint64_t* neg1s;  // Shoup companions of neg1
    uint32_t num_in, cnt_e;
    ExpandActive act;
    uint32_t auto_t;  // the round's automorphism x -> x^t
    uint32_t create_here;  // 1: cts with i >= num_in do not exist yet (round 0); 0: the previous round's MAC wrote them
    const uint64_t* query;  // create_here only, optional: cv[0] is read from here (and written to cv) instead of from cv
};
template <class L>
struct InvParamsT : InvParamsCore {
    using Core = InvParamsCore;
    using NoLanesT = InvParamsT<NoLanes>;
    L lanes;            // src, dst, cv, query per query lane (neg1 / neg1s are shared tables)
};
using InvParams = InvParamsT<Lanes>;
void launch_ntt_inverse(const DeviceTables& t, const InvParams& p, uint32_t store, uint32_t nblocks, hipStream_t s);
void launch_ntt_inverse_expand(const DeviceTables& t, const InvParams& p, uint32_t nblocks, hipStream_t s);

// fold chain (ntt.hip): PK polynomials [2*np][3][2] -> inverse transform, CRT lift, balanced digits, forward transforms
// into the fold operand layout D (as LD_SDIGIT); a workgroup handles one polynomial and dpb consecutive digits
struct FoldChainParamsCore {
    const uint64_t* src;
    uint64_t* dst;
    uint32_t ell, bits, fold_np;
    uint32_t pre_reduce;  // 1: fields are lazy sums (< 2^32), reduce mod m first
    uint32_t dpb;         // digits per block, 1 .. ell
    uint32_t lazy_out;    // 1: digit transforms left in [0, 2m) (the product kernel sums 6 * ell < 128 terms of < 2^57)
    // SpiralPack fold (foldCiphertextsDim1): sources are [trial][2*np][2] 2 x 1 ciphertexts with a trial stride of src_stride
    // ciphertexts, unsigned digits, operand layout as LD_PDIGIT / PM_FOLD
    uint32_t pack, src_stride;
};
template <class L>
struct FoldChainParamsT : FoldChainParamsCore {
    using Core = FoldChainParamsCore;
    using NoLanesT = FoldChainParamsT<NoLanes>;
    L lanes;
};
using FoldChainParams = FoldChainParamsT<Lanes>;
void launch_fold_chain(const DeviceTables& t, const FoldChainParams& p, uint32_t n_src, hipStream_t s);

// The fold round in pair form (ntt.hip LD_SDIFF + launch_fold_mac with an addend): out[i] = C[i] + Q * NTT(G^-1(C[np + i]) - G^-1(C[i])).
// split_and_crt's digits recompose the value exactly -- no borrow is lost at the top of the second carry chain -- when the
// digits cover at least 57 bits (values are below Q < 2^56) and every shift is a defined one
constexpr bool fold_pair_exact(uint32_t ell) { return ell >= 2 && ell * get_bits_per(ell) >= 57 && (ell - 1) * get_bits_per(ell) < 64; }

// ---- layout conversion at the C-ABI boundary -------------------------------------------------------
// reference polynomial b <-> packed polynomial pk_map(b)
void launch_ref_to_pk(const uint64_t* ref, uint64_t* pk, uint32_t npolys, IndexMap pk_map, hipStream_t s);
void launch_pk_to_ref(const uint64_t* pk, uint64_t* ref, uint32_t npolys, IndexMap pk_map, hipStream_t s);

// ---- pointwise polynomial kernels (poly.hip) -------------------------------------------------------
// out[b][r][c] = sum_m A[r][m] * B[b][m][c]  (+ addend), all PK; generic MatPoly multiply (src/poly.cpp:34)
struct MatmulParams {
    const uint64_t* a;  // [rs][ms] polys, stride a_batch polys between batches (0 = shared)
    const uint64_t* b;  // [ms][cs] polys
    uint64_t* out;      // [rs][cs] polys
    uint32_t rs, ms, cs;
    uint32_t a_batch, b_batch, out_batch;  // strides in polynomials
};
void launch_matmul(const MatmulParams& p, uint32_t batch, hipStream_t s);
// fold product: out[i][3][2] = key[3][K] * d[i][K][2], K = 2*m2 (src/spiral.cpp:1361-1383)
// key_stride: polynomials between the key's rows (K when the rows are K long; 2*m2 with K = m2 to take one half of [Q_neg | Q]);
// addend: optional [np][3][2] PK polynomials added to the products (fields may be any u32: lazy sums are fine)
void launch_fold_mac(const uint64_t* key, const uint64_t* d, uint64_t* out, uint32_t K, uint32_t np, hipStream_t s, uint32_t key_stride = 0,
                     const uint64_t* addend = nullptr, const Lanes& lanes = Lanes{});
// the reference's two-product fold round Q_neg D_L + Q D_H from the matrices q[3][m2] alone (Q_neg = G2 - Q is derived: poly.hip fold_mac_two_kernel);
// d: [np][2 halves][m2][2] (the LD_SDIGIT / fold_chain operand layout)
void launch_fold_mac_two(const uint64_t* q, const uint64_t* d, uint64_t* out, uint32_t m2, uint32_t ell, uint32_t bits, uint32_t np, hipStream_t s,
                         const Lanes& lanes = Lanes{});
// out = (a + b) mod m ; out = single * a (src/poly.cpp:138,190)
void launch_add(const uint64_t* a, const uint64_t* b, uint64_t* out, uint32_t npolys, hipStream_t s);
void launch_mul_by_const(const uint64_t* single, const uint64_t* a, uint64_t* out, uint32_t npolys, hipStream_t s);
// raw-domain automorphism / negation (src/poly.cpp:240,269)
void launch_automorph(const uint64_t* in, uint64_t* out, uint32_t npolys, uint32_t t, hipStream_t s);
void launch_invert(const uint64_t* in, uint64_t* out, uint32_t npolys, hipStream_t s);
// unsigned gadget digits in the raw domain (src/util.cpp:114)
void launch_gadget_invert(const uint64_t* in, uint64_t* out, uint32_t mx, uint32_t rdim, uint32_t cols, hipStream_t s);
// response modulus switch (src/poly.cpp:578-601, src/spiral.cpp:1441-1447)
void launch_rescale(const uint64_t* in, uint64_t* out, uint32_t n, uint64_t inp_mod, uint64_t out_mod, hipStream_t s);
void launch_response_wire(const uint64_t* in, uint64_t* out, uint32_t n0, uint32_t w0, uint32_t n1, uint32_t w1, hipStream_t s);
void launch_rescale2(const uint64_t* in, uint64_t* out, uint32_t n0, uint32_t n, uint64_t inp_mod, uint64_t out_mod0, uint64_t out_mod1, hipStream_t s,
                     const Lanes& lanes = Lanes{});

// ---- expansion / conversion / fold specials ----------------------------------------------------------
// the same for a whole round in one launch: active ct a < cnt_e even (W_left, t_e digits) else odd (W_right, t_o);
// g holds t digit polynomials per ct in LD_EXPAND job order; a1[2a + 1] is NTT(automorph(c_1)) of active ct a
struct ExpandMacParamsCore {
    uint64_t* cv;
    const uint64_t* w_e;
    const uint64_t* w_o;
    const uint64_t* g;
    const uint64_t* a1;
    uint32_t cnt_e, cnt_o, t_e, t_o;
    ExpandActive act;
    // next round's new ciphertexts cv[i + next_num_in] = neg1 * cv[i] (src/spiral.cpp:1709), written while the updated cv[i]
    // is in registers: every even i, and odd i when (i - 1)/2 + next_num_in/2 < next_cnt_o (the whole round's count of odd
    // ciphertexts, not a shard's).  neg1n == null: none.
    const uint64_t* neg1n;
    const uint64_t* neg1ns;  // Shoup companions
    uint32_t next_num_in, next_cnt_o;
};
template <class L>
struct ExpandMacParamsT : ExpandMacParamsCore {
    using Core = ExpandMacParamsCore;
    using NoLanesT = ExpandMacParamsT<NoLanes>;
    L lanes;  // cv, w_e, w_o, g, a1 per query lane (neg1n / neg1ns are shared tables)
};
using ExpandMacParams = ExpandMacParamsT<Lanes>;
void launch_expand_mac_round(const ExpandMacParams& p, hipStream_t s);
// scalToMat product: out[a][r][c] = sum_k W[r][2k+c] * G[a][k] + pad(cv[pos(a)][1])   (src/spiral.cpp:1850-1885)
// qs != null: also (or instead, out may be null) write the sweep's query records (see sweep)
struct Scal2MatParamsCore {
    const uint64_t* w;    // [3][2*t_conv] PK
    const uint64_t* g;    // [count][t_conv] PK digits of cv row 0
    const uint64_t* cv;   // expanded cts
    IndexMap cv_pos;      // a -> ct index in cv
    uint64_t* out;        // [count][3][2] PK or null
    uint32_t* qs;         // sweep query records [N][jm_total/2][12] u32 or null
    uint32_t t_conv, count, jm_total, j_base;
};
template <class L>
struct Scal2MatParamsT : Scal2MatParamsCore {
    using Core = Scal2MatParamsCore;
    using NoLanesT = Scal2MatParamsT<NoLanes>;
    L lanes;  // every pointer per query lane
};
using Scal2MatParams = Scal2MatParamsT<Lanes>;
void launch_scal2mat(const Scal2MatParams& p, hipStream_t s);
// regevToGSW assembly for one dimension: gsw[r][3i] = sum_k V[r][k] * chat[i][k], gsw[r][3i+1+c] = scalToMat(cv_i)[r][c]
struct GswParamsCore {
    const uint64_t* w;     // [3][2*t_conv]
    const uint64_t* v;     // [3][2*t_conv]
    const uint64_t* chat;  // [dims*ell][2*t_conv] PK digits (cv row 0 digits, then row 1 digits)
    const uint64_t* cv;
    IndexMap cv_pos;       // i (over dims*ell) -> ct index
    uint64_t* gsw;         // optional: [dims][3][3*ell] PK, dimension d stored at index (dims-1-d)  (src/spiral.cpp:2324)
    uint64_t* key;         // optional: the fold key of the same matrices, written in the same pass: key[d][r][0..m2) = G2 - gsw (= Q_neg, src/spiral.cpp:2361-2379), key[d][r][m2..2*m2) = gsw
    uint32_t t_conv, ell, dims;
};
template <class L>
struct GswParamsT : GswParamsCore {
    using Core = GswParamsCore;
    using NoLanesT = GswParamsT<NoLanes>;
    L lanes;  // every pointer per query lane
};
using GswParams = GswParamsT<Lanes>;
void launch_regev_to_gsw(const GswParams& p, hipStream_t s);
// both conversion products in one launch when the record-writing ScalToMat applies (else the two launches)
void launch_convert_products(const Scal2MatParams& sp, const GswParams& gp, hipStream_t s);
// the same key from the reference's reoriented (z, r, m) packed matrices (reorient_Q, src/spiral.cpp:388)
void launch_fold_key_from_reoriented(const uint64_t* q_re, const uint64_t* qneg_re, uint64_t* key, uint32_t m2, hipStream_t s);

// ---- first-dimension sweep (sweep.hip) -----------------------------------------------------------------
// device DB layout: common.h (packed 7-byte words for real geometries, plain for tiny ones), ic = ii*2 + c, nic = 2*num_per.
// acc[ii][r][c][z] PK (fields < m).
// g_log: log2 of the number of ranks of a distributed fold (accumulators grouped by ii mod 2^g_log), 0 = natural order
// k_log: log2 of the number of STAGES of a pipelined sweep (0 = one): the accumulators are laid out [stage][rank][ct] so that every
// stage's block can be reduce-scattered on its own (sweep.hip acc_pos); stage >= 0 launches only that stage's column blocks
// (wide packed geometries, sweep_stages_ok), stage < 0 all of them
void launch_sweep(const uint64_t* db, const uint32_t* qs, uint64_t* acc, uint32_t num_per, uint32_t jm_total, uint32_t g_log, hipStream_t s, uint32_t k_log = 0,
                  int stage = -1);
bool sweep_stages_ok(uint32_t num_per, uint32_t jm_total, uint32_t g_log, uint32_t k_log);
// n = 2 .. kSweepMaxBatch queries against one pass over the database (records qs[b] -> accumulators acc[b]); only where
// sweep_batch_ok (the packed layout with at least 64 output columns: every published geometry but the smallest streaming ones)
constexpr uint32_t kSweepMaxBatch = 2;
bool sweep_batch_ok(uint32_t num_per, uint32_t jm_total);
void launch_sweep_batch(const uint64_t* db, const uint32_t* const* qs, uint64_t* const* acc, uint32_t n, uint32_t num_per, uint32_t jm_total, uint32_t g_log,
                        hipStream_t s);
// the same on the matrix cores for n = 1 .. kMaxLanes queries per pass (sweep_mfma.hip): needs the "limb plane" image of the database, built
// from the packed one by launch_db_limb_planes (as many words); where sweep_mfma_ok (>= 64 ciphertexts per slot, first dimension a power of two in
// [64, 2048]).  Returns the launch's error (the > 64 KiB LDS opt-in is per device).  k_log: the accumulators' stage layout, as launch_sweep
bool sweep_mfma_ok(uint32_t num_per, uint32_t jm_total);
// nz slots z (both pointers at the first of them; a slot's region is db_device_words / kN words in either form, so a server converts its image IN PLACE
// a few slots at a time through a staging buffer); launch_db_limb_unplanes is the inverse map (limb planes -> packed), bit-exact both ways
void launch_db_limb_planes(const uint64_t* db_packed_img, uint64_t* db_limbs, uint32_t num_per, uint32_t jm_total, hipStream_t s, uint32_t nz = kN);
void launch_db_limb_unplanes(const uint64_t* db_limbs, uint64_t* db_packed_img, uint32_t num_per, uint32_t jm_total, hipStream_t s, uint32_t nz = kN);
hipError_t launch_sweep_mfma(const uint64_t* db_limbs, const uint32_t* const* qs, uint64_t* const* acc, uint32_t n, uint32_t num_per, uint32_t jm_total, uint32_t g_log,
                             hipStream_t s, uint32_t k_log = 0);
// reference DB layout (src/spiral.cpp:1139-1153) -> device layout, for the j-range [j0, j0 + dim0_shard): db_ref holds the nz
// consecutive z slabs z0 .. z0+nz-1, db_dev is the base of the shard's device database
void launch_db_relayout(const uint64_t* db_ref, uint64_t* db_dev, uint32_t num_per, uint32_t dim0, uint32_t j0, uint32_t dim0_shard, uint32_t z0,
                        uint32_t nz, hipStream_t s);
// reference reorientCiphertexts layout (z, j, m, r_pad4) u64 -> sweep query records
void launch_qs_from_reoriented(const uint64_t* reoriented, uint32_t* qs, uint32_t jm_total, hipStream_t s);
void launch_fill_db_random(uint64_t* db_dev, uint32_t num_per, uint32_t dim0_shard, uint64_t seed, hipStream_t s);
// read the device database back in the reference's layouts (tests, spiral_gpu_server_read_db_*): one plaintext item
// (j local to the shard) as n0 x n2 reference NTT-form polynomials, or nz slabs z0.. of load_db's layout restricted to
// the shard's j-range (z in the reference's slot order)
// limbs: the image is in limb-plane form (sweep_mfma.hip) instead of the packed one
void launch_db_read_item(const uint64_t* db_dev, uint64_t* out_ref, uint32_t num_per, uint32_t dim0_shard, uint32_t j_local, uint32_t ii, hipStream_t s,
                         bool limbs = false);
// (restricted to the n_ii plaintext columns ii0 .. ii0 + n_ii - 1: the same layout with num_per = n_ii)
void launch_db_read_slots(const uint64_t* db_dev, uint64_t* out, uint32_t num_per, uint32_t dim0_shard, uint32_t z0, uint32_t nz, uint32_t ii0, uint32_t n_ii,
                          hipStream_t s, bool limbs = false);
void launch_fill_db1_random(uint64_t* db_dev, uint32_t num_per, uint32_t dim0, uint64_t seed, hipStream_t s);
// the GSW-bit ciphertexts (odd slots 2 i + 1 of cv, i < n_bits) a rank of a G-rank answer expanded itself (i = a G + rank) <->
// its block of the all-gather buffer [rank][a < n_max][2 polynomials]; pack: cv -> this rank's block, unpack: all blocks -> cv
void launch_gsw_bits_pack(const uint64_t* cv, uint64_t* block, uint32_t rank, uint32_t n_ranks, uint32_t n_bits, hipStream_t s);
void launch_gsw_bits_unpack(uint64_t* cv, const uint64_t* gathered, uint32_t n_ranks, uint32_t n_bits, hipStream_t s);

// ---- SpiralPack (pack.hip; reference src/testing.cpp) -----------------------------------------------------------
// device DB layout, 1 x 1 plaintexts.  Packed (dim0 % 16 == 0; as the base path's, common.h): a word is two 28-bit
// residues in 7 bytes; a lane's 16 words of 16 consecutive j are one 112-byte string fetched as 7 x 16 bytes:
//     tile -> group j/16 -> chunk k < 7 -> lane -> 16 bytes,
// a tile being W = min(64, num_per) columns x P = 64/W consecutive slots z, lane = (z % P) * W + ii % W.
// Plain (dim0 < 16, tiny geometries): word(z, j, ii) at (((z*nblk + ii/W)*(dim0/2) + j/2)*W + ii%W)*2 + (j&1)
__host__ __device__ inline size_t db1_word_index(uint32_t z, uint32_t j, uint32_t ii, uint32_t num_per, uint32_t dim0) {
    const uint32_t w = num_per < 64u ? num_per : 64u, nblk = num_per / w;
    return ((((size_t)z * nblk + ii / w) * (dim0 / 2) + (j >> 1)) * w + ii % w) * 2u + (j & 1u);
}
__host__ __device__ inline bool db1_packed(uint32_t num_per, uint32_t dim0) { return (dim0 & 15u) == 0u && num_per >= 1u && (num_per & (num_per - 1u)) == 0u; }
__host__ __device__ inline size_t db1_device_words(uint32_t num_per, uint32_t dim0) {  // u64 words per trial
    const size_t words = (size_t)kN * dim0 * num_per;
    return db1_packed(num_per, dim0) ? words / 8u * 7u : words;
}
__host__ __device__ inline size_t db1_packed_byte(uint32_t z, uint32_t j, uint32_t ii, uint32_t by, uint32_t num_per, uint32_t dim0) {
    const uint32_t w = num_per < 64u ? num_per : 64u, pz = 64u / w, nblk = num_per / w;
    const uint32_t tile = (z / pz) * nblk + ii / w, lane = (z % pz) * w + ii % w, b = 7u * (j & 15u) + by;
    return ((((size_t)tile * (dim0 >> 4) + (j >> 4)) * 7u + (b >> 4)) * 64u + lane) * 16u + (b & 15u);
}
#ifdef __HIPCC__
__device__ __forceinline__ void db1_put_word(uint64_t* db, uint32_t z, uint32_t j, uint32_t ii, uint32_t num_per, uint32_t dim0, uint64_t v) {
    if (db1_packed(num_per, dim0)) {
        const uint64_t f = (uint64_t)(uint32_t)v | ((uint64_t)(uint32_t)(v >> 32) << 28);
        uint8_t* bytes = reinterpret_cast<uint8_t*>(db);
#pragma unroll
        for (uint32_t by = 0; by < 7; by++) bytes[db1_packed_byte(z, j, ii, by, num_per, dim0)] = (uint8_t)(f >> (8u * by));
    } else {
        db[db1_word_index(z, j, ii, num_per, dim0)] = v;
    }
}
#endif
// fastMultiplyQueryByDatabaseDim1 (src/testing.cpp:364): acc[ii][r][z] PK; qs1 records [z][j] = {p r0, p r1, b r0, b r1}
// `trials` databases db + t*db_stride swept in one launch into acc + t*acc_stride (strides in u64 words)
void launch_sweep1(const uint64_t* db, const uint32_t* qs1, uint64_t* acc, uint32_t num_per, uint32_t dim0, uint32_t trials, size_t db_stride,
                   size_t acc_stride, hipStream_t s);
// query records from the expanded cts: first-dimension ct j is cv[j * idx_factor] (reorientCiphertextsDim1, :342)
void launch_qs1_from_cv(const uint64_t* cv, uint32_t* qs1, uint32_t dim0, uint32_t idx_factor, hipStream_t s);
void launch_qs1_from_reoriented(const uint64_t* re, uint32_t* qs1, uint32_t dim0, hipStream_t s);
// convertDb layout (:316-340) z*(num_per*dim0) + ii*dim0 + j -> device layout
// foldCiphertextsDim1 product for `count` ciphertexts: out[b][2] = key[2][K] * d[b][K]   (src/testing.cpp:596-624)
// key_stride: polynomials between the key's two rows (0 = K); addend (pair form: out = L + F * D'): PK ciphertexts, ciphertext b = t * np + i at
// polynomial (t * add_stride + i) * 2 + r of `addend`
void launch_pack_fold_mac(const uint64_t* key, const uint64_t* d, uint64_t* out, uint32_t K, uint32_t count, hipStream_t s, uint32_t key_stride = 0,
                          const uint64_t* addend = nullptr, uint32_t np = 1, uint32_t add_stride = 1);
void launch_db1_relayout(const uint64_t* ref, uint64_t* dev, uint32_t num_per, uint32_t dim0, hipStream_t s);
// gsw[i][r][2j] = tmp[i*ell+j][r], gsw[i][r][2j+1] = cv[2*(i*ell+j)+1][r]   (regevToSimpleGsw, :108-139)
void launch_pack_gsw_assemble(const uint64_t* tmp, const uint64_t* cv, uint64_t* gsw, uint32_t ell, uint32_t nu2, hipStream_t s);
void launch_pack_gsw_from_upload(const uint64_t* query, uint64_t* gsw, uint32_t dim0, uint32_t ell, uint32_t nu2, hipStream_t s);
// key[cur][r][0..2ell) = gadget - F, [2ell..4ell) = F with F = gsw[nu2-1-cur]   (:1027-1032, 611-618)
void launch_pack_fold_key(const uint64_t* gsw, uint64_t* key, uint32_t ell, uint32_t nu2, hipStream_t s);
// pack (:198-241): result[row][c] = sum_r sum_k W_r[row][k] * ginv[r*out_n+c][k] + (row >= 1 ? ct2[(row-1)*out_n+c] : 0)
void launch_pack_mac(const uint64_t* v_w, const uint64_t* ginv, const uint64_t* ct2, uint64_t* result, uint32_t out_n, uint32_t t_conv, hipStream_t s);

}  // namespace spiral
