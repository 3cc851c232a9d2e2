// Host-side helpers shared by server.cpp and pack_server.cpp (error reporting, device buffers, the
// coefficient-expansion driver).  Internal to libspiral_gpu.so.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/spiral_gpu.h"
#include "common.h"
#include "kernels.h"

namespace spiral {
namespace host {

extern thread_local std::string g_err;

inline int fail(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return -1;
}

#define HIP_OK(expr)                                                                                  \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

constexpr size_t kPolyBytes = (size_t)kN * sizeof(uint64_t);  // PK or RAW polynomial
constexpr size_t kRefNtt = 2 * (size_t)kN;                     // words of a reference NTT-form polynomial
constexpr uint64_t kQprimeMods[37] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 12289, 12289, 61441, 65537, 65537, 520193, 786433, 786433,
                                      3604481, 7340033, 16515073, 33292289, 67043329, 132120577, 268369921, 469762049, 1073479681,
                                      2013265921, 4293918721ull, 8588886017ull, 17175674881ull, 34359214081ull, 68718428161ull};  // values.h:74-76

inline uint32_t ceil_log2(uint64_t x) {
    uint32_t r = 0;
    while ((1ull << r) < x) r++;
    return r;
}

inline int shape_of(const spiral_gpu_params* p, spiral_gpu_shape* s) {
    if (!p || !s) return fail("null argument");
    if (p->nu1 > 16 || p->nu2 > 16) return fail("nu1/nu2 out of range");
    // 56 digits at most: the expansion's digit transforms are left lazy ([0, 2m), LD_EXPAND) and expand_mac_round_* sums up to
    // t_exp + t_exp_right of them per u64 accumulator -- raising the cap needs lazy_ok to hold for the new sum
    static_assert(spiral::lazy_ok(56) && spiral::lazy_ok(2 * 56), "the digit-count cap below keeps the lazy expansion digits inside the u64 accumulators");
    if (p->t_gsw < 2 || p->t_gsw > 28 || p->t_conv < 1 || p->t_conv > 56 || p->t_exp < 1 || p->t_exp > 56 || p->t_exp_right < 1 ||
        p->t_exp_right > 56)
        return fail("gadget dimension out of range");
    if (p->qprime_bits >= 37 || kQprimeMods[p->qprime_bits] == 0) return fail("unsupported q' bit width %u", p->qprime_bits);
    if (p->p_db < 2 || p->p_db > (1ull << 40)) return fail("unsupported plaintext modulus");
    s->dim0 = 1u << p->nu1;
    s->num_per = 1u << p->nu2;
    s->ell = p->t_gsw;
    s->m2 = 3 * p->t_gsw;
    s->n_bits = s->dim0 + s->ell * p->nu2;
    s->qprime = kQprimeMods[p->qprime_bits];
    if (p->direct_upload) {
        s->g = s->stopround = s->n_left = s->n_right = 0;
        s->n_query_cts = s->n_bits;
    } else {
        s->g = ceil_log2(s->n_bits);
        s->stopround = p->nu2 ? ceil_log2((uint64_t)s->ell * p->nu2) : 0;
        if (s->ell * p->nu2 > s->dim0) s->stopround = 0;  // src/spiral.cpp:2083
        s->n_left = s->g;
        s->n_right = s->stopround ? s->stopround + 1 : s->g;
        s->n_query_cts = 1;
        if (s->g > kLogN) return fail("query does not fit one polynomial (g = %u)", s->g);
    }
    return 0;
}

// wire form of a switched response (include/spiral_gpu.h): row 0 at qprime_bits, the rest at the bits that hold a value < 4 p_db
inline uint32_t wire_bits_rest(const spiral_gpu_params* p) { return ceil_log2(4 * p->p_db); }
inline size_t wire_bytes(const spiral_gpu_params* p, uint32_t out_n) {
    return ((size_t)out_n * kN * p->qprime_bits + (size_t)out_n * out_n * kN * wire_bits_rest(p)) / 8;
}

inline uint32_t inv_mod_2n(uint32_t t) {  // t odd, inverse modulo 2N = 4096
    uint32_t x = 1;
    for (int i = 0; i < 12; i++) x = x * (2 - t * x);  // Newton, doubles the valid bits
    return x & (2 * kN - 1);
}

struct DevBuf {
    uint64_t* p = nullptr;
    size_t words = 0;
    bool carved = false;  // a piece of an Arena: not freed on its own
    int alloc(size_t w) {
        words = w ? w : 1;
        carved = false;
        HIP_OK(hipMalloc(&p, words * sizeof(uint64_t)));
        return 0;
    }
    void release() {
        if (p && !carved) (void)hipFree(p);
        p = nullptr;
    }
};
// One allocation holding all per-query buffers of a server in a fixed order, so that two servers with the same parameters have the
// same internal layout and lane q's buffer X is lane 0's X + (arena_q - arena_0): what lets one launch serve several query lanes
// (kernels.h Lanes).  Two passes over the same carve sequence: base == nullptr sizes it, then the real base hands out the pieces.
struct Arena {
    uint64_t* base = nullptr;
    size_t used = 0;
    void carve(DevBuf& b, size_t w) {
        b.words = w ? w : 1;
        b.carved = true;
        b.p = base ? base + used : nullptr;
        used += (b.words + 31u) & ~(size_t)31u;  // 256-byte pieces
    }
};

// scoped device scratch for the host-buffer seams
struct Scratch {
    std::vector<void*> ptrs;
    ~Scratch() {
        for (void* p : ptrs) (void)hipFree(p);
    }
    uint64_t* get(size_t words) {
        void* p = nullptr;
        if (hipMalloc(&p, (words ? words : 1) * sizeof(uint64_t)) != hipSuccess) return nullptr;
        ptrs.push_back(p);
        return (uint64_t*)p;
    }
    uint64_t* upload(const uint64_t* host, size_t words) {
        uint64_t* d = get(words);
        if (d && hipMemcpy(d, host, words * sizeof(uint64_t), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
        return d;
    }
};

inline int current_tables(DeviceTables* t) {
    int dev = 0;
    HIP_OK(hipGetDevice(&dev));
    if (tables_get(dev, t) != 0) return fail("twiddle table setup failed on device %d", dev);
    return 0;
}

// ---- expansion on PK buffers, shared by the seam and the resident server ---------------------------------
struct ExpandWork {
    uint64_t* raw;  // per active ct a: [2a] = automorph(c_0) RAW, [2a + 1] = NTT(automorph(c_1)) PK
    uint64_t* g;    // per active ct: t digit polynomials, PK
};
inline size_t expand_g_polys(uint32_t g, uint32_t t_exp, uint32_t t_exp_right) {
    size_t half = (size_t)1 << (g ? g - 1 : 0);  // at most 2^(g-1) active cts per parity
    return half * ((size_t)t_exp + 1 + t_exp_right + 1);
}

// What one rank of a G = 2^g_log rank answer expands (all zero: everything).  The expanded ciphertexts end up at cv[2 j]
// (first-dimension index j) and cv[2 i + 1] (GSW bit i) (reorderFromStopround, src/spiral.cpp:2027-2036), and slot index
// bit r is decided in round r.  A rank needs the first-dimension ciphertexts of its own j in [rank J, (rank + 1) J),
// J = 2^j_log: in round r <= j_log those still descend from every even ciphertext of the round, afterwards only from the
// J whose higher slot bits spell the low bits of `rank`.  Of the GSW bits -- all ranks need all of them, for the folding
// keys -- each rank expands those with i = rank mod G from round g_log on, and the ranks exchange them (one all-gather).
struct ExpandShard {
    uint32_t rank, g_log, j_log;
};

// src/spiral.cpp:1664-1743.  cv: 2^g cts (2 PK polys each).
inline void run_expand(const DeviceTables& tb, uint64_t* cv, uint32_t g, uint32_t t_exp, const uint64_t* w_left, uint32_t t_exp_right,
                const uint64_t* w_right, uint32_t max_bits_right, uint32_t stopround, const ExpandWork& wk, hipStream_t st,
                const uint64_t* query = nullptr,  // query: the packed query ciphertext when cv[0] does not hold it yet
                uint32_t r_begin = 0, uint32_t r_end = 0xffffffffu,  // rounds [r_begin, min(r_end, g))
                const ExpandShard& shard = ExpandShard{},
                uint32_t parity = 3,  // bit 0: the even-index ciphertexts, bit 1: the odd-index ones.  After round 0 the two trees never read each
                                      // other (a ciphertext is created from the one num_in = 2^r slots below it: same parity for r >= 1, and in
                                      // round 0 both come from the query), so with stopround > 0 -- evens = first-dimension ciphertexts, odds = GSW
                                      // bits -- the two halves can run as independent launch sequences on their own work buffers
                const Lanes& lanes = Lanes{}) {  // query lanes: cv, w_left, w_right, wk, query are lane 0's (kernels.h)
    // active odd-index ciphertexts of round r (:1701-1702); the even ones are all 2^r
    auto odd_count = [&](uint32_t r) {
        const uint32_t num_in = 1u << r;
        if (stopround > 0 && r > stopround) return 0u;
        if (stopround > 0 && r == stopround) return std::min(num_in, max_bits_right + 1);
        return num_in;
    };
    for (uint32_t r = r_begin; r < std::min(r_end, g); r++) {
        const uint32_t num_in = 1u << r;
        const uint32_t t = (kN >> r) + 1;
        uint32_t cnt_even = (parity & 1u) ? num_in : 0u, cnt_odd = (parity & 2u) ? odd_count(r) : 0u;
        if (cnt_even + cnt_odd == 0) continue;
        ExpandActive act{};
        if (shard.g_log) {
            if (r > shard.j_log && cnt_even) {  // 2^(r - j_log) blocks of J even ciphertexts: this rank's is the one its low bits name
                cnt_even = 1u << shard.j_log;
                act.e_off = (shard.rank & ((1u << (r - shard.j_log)) - 1u)) << shard.j_log;
            }
            if (r >= shard.g_log) {  // every G-th odd ciphertext, starting at `rank`
                const uint32_t G = 1u << shard.g_log;
                cnt_odd = cnt_odd > shard.rank ? (cnt_odd - shard.rank + G - 1u) / G : 0u;
                act.o_stride_m1 = G - 1u;
                act.o_off = shard.rank;
            }
        }
        // 1) INTT + CRT of row 0 and the automorphed row 1 (a slot permutation) of every active ct, both parities;
        //    cts with i >= num_in are neg1 * cv[i - num_in] (:1709): created inside this kernel in round 0, by the
        //    previous round's MAC afterwards
        const uint32_t cnt = cnt_even + cnt_odd;
        InvParams ip{};
        ip.dst = wk.raw;
        ip.src_map = ip.dst_map = identity_map();
        ip.cv = cv;
        ip.neg1 = tb.neg1 + (size_t)r * kN;
        ip.neg1s = tb.neg1s + (size_t)r * kN;
        ip.num_in = num_in;
        ip.cnt_e = cnt_even;
        ip.act = act;
        ip.auto_t = t;
        ip.create_here = r == 0;
        ip.query = query;
        ip.lanes = lanes;
        launch_ntt_inverse_expand(tb, ip, 2 * cnt, st);
        // 2) G^-1(automorph(c)[0]) digits (t_exp / t_exp_right per ct), one launch
        FwdParams fp{};
        fp.src = wk.raw;
        fp.dst = wk.g;
        fp.src_map = fp.dst_map = identity_map();
        fp.n_digits = 1;
        fp.cnt_e = cnt_even;
        fp.t_e = t_exp;
        fp.t_o = t_exp_right;
        fp.lanes = lanes;
        launch_ntt_forward(tb, fp, LD_EXPAND, ST_PK, cnt_even * t_exp + cnt_odd * t_exp_right, st);
        // 3) cv[i] += W * digits + (0, NTT(c'_1))
        ExpandMacParams mp{};
        mp.cv = cv;
        mp.w_e = w_left + (size_t)r * 2 * t_exp * kN;
        mp.w_o = w_right + (size_t)r * 2 * t_exp_right * kN;  // never dereferenced when cnt_odd == 0
        mp.g = wk.g;
        mp.a1 = wk.raw;
        mp.cnt_e = cnt_even;
        mp.cnt_o = cnt_odd;
        mp.act = act;
        mp.t_e = t_exp;
        mp.t_o = t_exp_right;
        if (r + 1 < g) {
            mp.neg1n = tb.neg1 + (size_t)(r + 1) * kN;
            mp.neg1ns = tb.neg1s + (size_t)(r + 1) * kN;
            mp.next_num_in = 2 * num_in;
            mp.next_cnt_o = odd_count(r + 1);
        }
        mp.lanes = lanes;
        launch_expand_mac_round(mp, st);
    }
}


// Raw database ingest shared by the two servers (the first half of load_db, src/spiral.cpp:1083-1171, on the
// device): items [lo, hi) of a host stream whose first item is `first` are staged a chunk at a time and handed to `launch`
// (items_dev, first_item_of_chunk, n_items_of_chunk), which runs the centred lift + batched transform + layout scatter.
// polys_per_item: n0*n2 = 4 (base) or 1 (SpiralPack).  The error word is set by the kernels when a coefficient is >= p_db.
template <class Launch>
inline int ingest_items(const void* items, uint32_t coeff_bits, uint64_t first, uint64_t lo, uint64_t hi, uint32_t polys_per_item, uint64_t p_db,
                        hipStream_t st, Launch launch) {
    if (!items) return fail("null item stream");
    if (coeff_bits != 64 && (coeff_bits < 1 || coeff_bits > 40)) return fail("coefficient width %u not in 1..40 or 64", coeff_bits);
    if (coeff_bits < 64 && (1ull << coeff_bits) < p_db) return fail("%u-bit coefficients cannot hold values below p_db", coeff_bits);
    if (lo >= hi) return 0;
    const size_t item_bytes = (size_t)polys_per_item * kN * coeff_bits / 8;
    const size_t stage_bytes = options().db_stage_bytes;  // (tests force several passes)
    const uint64_t chunk = std::max<uint64_t>(1, std::min<uint64_t>({stage_bytes / item_bytes, (uint64_t)(1u << 16), hi - lo}));
    uint8_t* d_items = nullptr;
    uint32_t* d_err = nullptr;
    HIP_OK(hipMalloc(&d_items, chunk * item_bytes + 16));
    if (hipMalloc(&d_err, sizeof(uint32_t)) != hipSuccess || hipMemsetAsync(d_err, 0, sizeof(uint32_t), st) != hipSuccess) {
        (void)hipFree(d_items);
        return fail("device allocation failed");
    }
    hipError_t e = hipMemsetAsync(d_items + chunk * item_bytes, 0, 16, st);
    for (uint64_t done = lo; done < hi && e == hipSuccess; done += chunk) {
        const uint64_t n = std::min(chunk, hi - done);
        e = hipMemcpyAsync(d_items, (const uint8_t*)items + (size_t)(done - first) * item_bytes, (size_t)n * item_bytes, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) break;
        launch(d_items, d_err, done, n);
        e = hipStreamSynchronize(st);  // the staging buffer is reused by the next pass
    }
    uint32_t err = 0;
    if (e == hipSuccess) e = hipMemcpy(&err, d_err, sizeof(err), hipMemcpyDeviceToHost);
    (void)hipFree(d_items);
    (void)hipFree(d_err);
    if (e != hipSuccess) return fail("database ingest failed: %s", hipGetErrorString(e));
    if (err) return fail("a plaintext coefficient is not below p_db (the reference asserts, src/spiral.cpp:1117)");
    return 0;
}

// upload reference NTT-form polynomials and convert to PK / the converse
inline uint64_t* upload_pk(Scratch& sc, const uint64_t* host_ref, size_t npolys) {
    uint64_t* d_ref = sc.upload(host_ref, npolys * kRefNtt);
    uint64_t* d_pk = sc.get(npolys * kN);
    if (!d_ref || !d_pk) return nullptr;
    launch_ref_to_pk(d_ref, d_pk, (uint32_t)npolys, identity_map(), 0);
    return d_pk;
}
inline int download_pk(Scratch& sc, const uint64_t* d_pk, IndexMap map, uint64_t* host_ref, size_t npolys) {
    uint64_t* d_ref = sc.get(npolys * kRefNtt);
    if (!d_ref) return fail("device allocation failed");
    launch_pk_to_ref(d_pk, d_ref, (uint32_t)npolys, map, 0);
    HIP_OK(hipMemcpy(host_ref, d_ref, npolys * kRefNtt * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return 0;
}

}  // namespace host
}  // namespace spiral
