/*
 * spiral_oracle.c -- CPU restatement of the Spiral server-answer path.  TEST INFRASTRUCTURE ONLY.
 * See spiral_oracle.h for the pinning statement.  Scalar C (one exception: the first-dimension cell also exists in the
 * reference's AVX-512 / AVX2 form, proven equal to the scalar one by tests/test_oracle.py); semantics are the reference's
 * scalar paths ("mathematical sum mod m", SURVEY.md section 8c hazard 4).  All NTT-domain outputs are
 * canonical residues in [0,m) (the reference's AVX2 tail may leave m instead of 0,
 * src/core.cpp:308,342,347 -- compare NTT-domain buffers mod m).
 */
#include "spiral_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define N ORC_N
#define LOGN 11
#define Q ORC_Q
#define N0 2 /* include/values.h:67 */
#define N1 3 /* include/values.h:68 */
#define N2 2 /* include/values.h:69 */
#define NTTP (2 * N) /* words per NTT-form polynomial */

typedef unsigned __int128 u128;

/* Threading (bench.py's all-cores CPU baseline only): the loops below over independent polynomials / NTT slots / ciphertexts
 * carry `#pragma omp parallel for if (g_threads > 1)`.  The default build (Makefile target liboracle.so, what the parity tests
 * load) has no -fopenmp, so the pragmas are ignored there; the `native` target adds -fopenmp -march=native and
 * orc_set_threads(n) turns them on.  The reference itself is single-threaded (src/spiral.cpp:1231). */
#ifdef _OPENMP
#include <malloc.h>
#include <omp.h>
#endif
static int g_threads = 1;
static void build_tables(void);
int orc_get_threads(void) { return g_threads; }
int orc_set_threads(int n) {
    build_tables(); /* lazily built otherwise: not from inside a parallel region */
#ifdef _OPENMP
    /* the restated functions malloc their scratch per call like the reference's MatPoly does (include/poly.h:32-58); above
     * glibc's mmap threshold every such block is an mmap / munmap pair, and hundreds of threads doing that serialise on the
     * address-space lock: keep those blocks in the per-thread arenas */
    mallopt(M_MMAP_THRESHOLD, 1 << 30);
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    g_threads = n < 1 ? 1 : n;
    omp_set_num_threads(g_threads);
    return g_threads;
#else
    (void)n;
    return 1;
#endif
}

static const uint64_t MODS[2] = {ORC_P, ORC_B};
/* minimal primitive 4096-th roots of unity mod p, b; checked against src/constants.cpp by
 * tests/test_oracle_tables.py (tests/golden/ntt_tables.json) */
static const uint64_t PSI[2] = {66687, 158221};

/* include/values.h:74-76 */
static const uint64_t QPRIME_MODS[37] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 12289, 12289, 61441,
    65537, 65537, 520193, 786433, 786433, 3604481, 7340033, 16515073, 33292289, 67043329, 132120577,
    268369921, 469762049, 1073479681, 2013265921, 4293918721ull, 8588886017ull, 17175674881ull,
    34359214081ull, 68718428161ull};

/* ------------------------------------------------------------------------------------------------ */
/* tables (src/constants.cpp:16 layout, src/core.cpp:6-17)                                          */
/* ------------------------------------------------------------------------------------------------ */
static uint64_t g_tab[8 * N];
static int g_tab_ready = 0;

static uint64_t powmod(uint64_t b, uint64_t e, uint64_t m) {
    uint64_t r = 1;
    b %= m;
    while (e) {
        if (e & 1) r = (uint64_t)((u128)r * b % m);
        b = (uint64_t)((u128)b * b % m);
        e >>= 1;
    }
    return r;
}

static uint32_t bitrev11(uint32_t x) {
    uint32_t r = 0;
    for (int i = 0; i < LOGN; i++) r |= ((x >> i) & 1u) << (LOGN - 1 - i);
    return r;
}

static void build_tables(void) {
    if (g_tab_ready) return;
    for (int n = 0; n < 2; n++) {
        uint64_t m = MODS[n], psi = PSI[n];
        uint64_t ipsi = powmod(psi, 2 * N - 1, m); /* psi^-1 since psi^(2N) = 1 */
        uint64_t half = (m + 1) / 2;
        uint64_t *inv_w = &g_tab[(0 + 2 * n) * N], *inv_ws = &g_tab[(1 + 2 * n) * N];
        uint64_t *fwd_w = &g_tab[(4 + 2 * n) * N], *fwd_ws = &g_tab[(5 + 2 * n) * N];
        uint64_t f = 1, v = half;
        for (uint32_t i = 0; i < N; i++) {
            uint32_t k = bitrev11(i);
            fwd_w[k] = f;
            fwd_ws[k] = (f << 32) / m;
            inv_w[k] = v;
            inv_ws[k] = (v << 32) / m;
            f = (uint64_t)((u128)f * psi % m);
            v = (uint64_t)((u128)v * ipsi % m);
        }
    }
    g_tab_ready = 1;
}

void orc_get_tables(uint64_t *out) {
    build_tables();
    memcpy(out, g_tab, sizeof(g_tab));
}

/* ------------------------------------------------------------------------------------------------ */
/* NTT core                                                                                         */
/* ------------------------------------------------------------------------------------------------ */

/* The reference's vector form of the same transforms (USE_AVX2, src/core.cpp:292-349 and :479-506): the forward butterflies four
 * at a time wherever a group is at least four wide (t >= 4: three _mm256_mul_epu32 per four butterflies), the t < 4 stages scalar,
 * and the closing [0,4m) -> [0,m) corrections of both transforms as vector compare / mask / subtract; the inverse butterflies are
 * scalar in the reference too (:447-474).  orc_set_ntt_simd(1) routes orc_ntt_forward / orc_ntt_inverse through it so that the CPU
 * baseline times the instruction mix the reference runs; the default stays the scalar restatement.  The comparisons here are
 * ">= bound" (cmpgt against bound - 1) where the reference's vector code has "> bound" (:304, :336-343, which can leave m or 2m in
 * place of 0): outputs are then word-for-word those of the scalar path, which tests/test_oracle.py checks. */
static int g_ntt_simd = 0;
#if defined(__AVX2__)
#include <immintrin.h>
#define ORC_NTT_ISA "avx2"
static inline __m256i csub_epi64(__m256i x, __m256i bound, __m256i bound_m1) { /* x >= bound ? x - bound : x, values < 2^63 */
    return _mm256_sub_epi64(x, _mm256_and_si256(_mm256_cmpgt_epi64(x, bound_m1), bound));
}
static void canonicalize_avx2(uint64_t *a, uint64_t m) {
    const __m256i two_m = _mm256_set1_epi64x((long long)(2 * m)), two_m1 = _mm256_set1_epi64x((long long)(2 * m - 1));
    const __m256i one_m = _mm256_set1_epi64x((long long)m), one_m1 = _mm256_set1_epi64x((long long)(m - 1));
    for (uint32_t i = 0; i < N; i += 4) {
        __m256i x = _mm256_loadu_si256((const __m256i *)(a + i));
        x = csub_epi64(csub_epi64(x, two_m, two_m1), one_m, one_m1);
        _mm256_storeu_si256((__m256i *)(a + i), x);
    }
}
static void ntt_forward_avx2(uint64_t *op) {
    for (int n = 0; n < 2; n++) {
        const uint64_t *w = &g_tab[(4 + 2 * n) * N], *ws = &g_tab[(5 + 2 * n) * N];
        uint64_t *a = op + n * N;
        const uint32_t m = (uint32_t)MODS[n], two_m = 2 * m;
        const __m256i vm = _mm256_set1_epi64x(m), v2m = _mm256_set1_epi64x(two_m), v2m1 = _mm256_set1_epi64x((long long)two_m - 1);
        for (uint32_t grp = 1, t = N / 2; grp < N; grp <<= 1, t >>= 1) {
            for (uint32_t i = 0; i < grp; i++) {
                const uint64_t W = w[grp + i], Ws = ws[grp + i];
                uint64_t *x = a + 2 * i * t, *y = x + t;
                if (t >= 4) {
                    const __m256i vw = _mm256_set1_epi64x((long long)W), vws = _mm256_set1_epi64x((long long)Ws);
                    for (uint32_t j = 0; j < t; j += 4) {
                        const __m256i xv = _mm256_loadu_si256((const __m256i *)(x + j)), yv = _mm256_loadu_si256((const __m256i *)(y + j));
                        const __m256i cx = csub_epi64(xv, v2m, v2m1);
                        const __m256i q = _mm256_srli_epi64(_mm256_mul_epu32(yv, vws), 32);
                        const __m256i tt = _mm256_sub_epi64(_mm256_mul_epu32(yv, vw), _mm256_mul_epu32(q, vm)); /* in [0,2m) */
                        _mm256_storeu_si256((__m256i *)(x + j), _mm256_add_epi64(cx, tt));
                        _mm256_storeu_si256((__m256i *)(y + j), _mm256_add_epi64(cx, _mm256_sub_epi64(v2m, tt)));
                    }
                } else {
                    for (uint32_t j = 0; j < t; j++) {
                        uint32_t xv = (uint32_t)x[j], yv = (uint32_t)y[j];
                        uint32_t cx = xv - (xv >= two_m ? two_m : 0);
                        uint64_t q = ((uint64_t)yv * Ws) >> 32;
                        uint64_t tt = W * yv - q * m;
                        x[j] = cx + tt;
                        y[j] = cx + (two_m - tt);
                    }
                }
            }
        }
        canonicalize_avx2(a, m);
    }
}
#else
#define ORC_NTT_ISA "scalar"
#endif
const char *orc_ntt_isa(void) { return ORC_NTT_ISA; }
int orc_set_ntt_simd(int on) {
#if defined(__AVX2__)
    g_ntt_simd = on != 0;
#else
    g_ntt_simd = 0;
#endif
    return g_ntt_simd;
}

/* src/core.cpp:254-351 (scalar branch :274-290): lazy Harvey butterflies, values in [0,4m) */
static void ntt_forward_scalar(uint64_t *op);
static void ntt_inverse_impl(uint64_t *op, int simd_tail);
void orc_ntt_forward(uint64_t *op) {
    build_tables();
#if defined(__AVX2__)
    if (g_ntt_simd) {
        ntt_forward_avx2(op);
        return;
    }
#endif
    ntt_forward_scalar(op);
}
void orc_ntt_forward_scalar(uint64_t *op) {
    build_tables();
    ntt_forward_scalar(op);
}
static void ntt_forward_scalar(uint64_t *op) {
    for (int n = 0; n < 2; n++) {
        const uint64_t *w = &g_tab[(4 + 2 * n) * N], *ws = &g_tab[(5 + 2 * n) * N];
        uint64_t *a = op + n * N;
        uint32_t m = (uint32_t)MODS[n], two_m = 2 * m;
        for (uint32_t grp = 1, t = N / 2; grp < N; grp <<= 1, t >>= 1) {
            for (uint32_t i = 0; i < grp; i++) {
                uint64_t W = w[grp + i], Ws = ws[grp + i];
                uint64_t *x = a + 2 * i * t, *y = x + t;
                for (uint32_t j = 0; j < t; j++) {
                    uint32_t xv = (uint32_t)x[j], yv = (uint32_t)y[j];
                    uint32_t cx = xv - (xv >= two_m ? two_m : 0);
                    uint64_t q = ((uint64_t)yv * Ws) >> 32;
                    uint64_t tt = W * yv - q * m; /* in [0,2m) */
                    x[j] = cx + tt;
                    y[j] = cx + (two_m - tt);
                }
            }
        }
        for (uint32_t i = 0; i < N; i++) { /* :344-349 with >= (scalar form :350-353) */
            uint64_t v = a[i];
            if (v >= two_m) v -= two_m;
            if (v >= m) v -= m;
            a[i] = v;
        }
    }
}

/* src/core.cpp:426-513: Gentleman-Sande with the 1/2 folded into every stage */
void orc_ntt_inverse(uint64_t *op) {
    build_tables();
    ntt_inverse_impl(op, g_ntt_simd);
}
void orc_ntt_inverse_scalar(uint64_t *op) {
    build_tables();
    ntt_inverse_impl(op, 0);
}
static void ntt_inverse_impl(uint64_t *op, int simd_tail) {
    for (int n = 0; n < 2; n++) {
        const uint64_t *w = &g_tab[(0 + 2 * n) * N], *ws = &g_tab[(1 + 2 * n) * N];
        uint64_t *a = op + n * N;
        uint64_t m = MODS[n], two_m = 2 * m;
        uint32_t t = 1;
        for (uint32_t h = N / 2; h >= 1; h >>= 1, t <<= 1) {
            for (uint32_t i = 0; i < h; i++) {
                uint64_t W = w[h + i], Ws = ws[h + i];
                uint64_t *u = a + 2 * i * t, *v = u + t;
                for (uint32_t j = 0; j < t; j++) {
                    uint64_t uu = u[j], vv = v[j];
                    uint64_t T = two_m - vv + uu;
                    uint64_t cu = uu + vv - (((uu << 1) >= T) ? two_m : 0);
                    u[j] = (cu + ((T & 1) ? m : 0)) >> 1;
                    uint64_t H = (T * Ws) >> 32;
                    v[j] = W * T - H * m;
                }
            }
        }
#if defined(__AVX2__)
        if (simd_tail) { /* :479-506 */
            canonicalize_avx2(a, m);
            continue;
        }
#endif
        (void)simd_tail;
        for (uint32_t i = 0; i < N; i++) {
            uint64_t x = a[i];
            if (x >= two_m) x -= two_m;
            if (x >= m) x -= m;
            a[i] = x;
        }
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* polynomial algebra                                                                               */
/* ------------------------------------------------------------------------------------------------ */

/* include/poly.h:137-153: Barrett reduce of a u64 == exact x mod m (quotient estimate off by <=1) */
static inline uint64_t red(uint64_t x, int n) { return x % MODS[n]; }

void orc_to_ntt(uint64_t *out, const uint64_t *in, size_t npolys) {
#pragma omp parallel for if (g_threads > 1 && npolys > 1)
    for (size_t k = 0; k < npolys; k++) {
        uint64_t *o = out + k * NTTP;
        const uint64_t *a = in + k * N;
        for (uint32_t z = 0; z < N; z++) {
            o[z] = red(a[z], 0);
            o[N + z] = red(a[z], 1);
        }
        orc_ntt_forward(o);
    }
}

void orc_to_ntt_no_reduce(uint64_t *out, const uint64_t *in, size_t npolys) {
#pragma omp parallel for if (g_threads > 1 && npolys > 1)
    for (size_t k = 0; k < npolys; k++) {
        uint64_t *o = out + k * NTTP;
        const uint64_t *a = in + k * N;
        for (uint32_t z = 0; z < N; z++) o[z] = o[N + z] = a[z];
        orc_ntt_forward(o);
    }
}

/* src/poly.cpp:344-353 + :11-32: (x*b_inv_pa_i + y*pa_inv_b_i) mod Q, values.h:24-25 */
uint64_t orc_crt_compose(uint64_t x, uint64_t y) {
    const u128 c_p = (u128)163640210ull * ORC_B; /* (b^-1 mod p) * b */
    const u128 c_b = (u128)97389680ull * ORC_P;  /* (p^-1 mod b) * p */
    u128 v = (u128)x * c_p + (u128)y * c_b;
    return (uint64_t)(v % Q);
}

void orc_from_ntt(uint64_t *out, const uint64_t *in, size_t npolys) {
#pragma omp parallel for if (g_threads > 1 && npolys > 1)
    for (size_t k = 0; k < npolys; k++) {
        uint64_t tmp[NTTP];
        memcpy(tmp, in + k * NTTP, sizeof(tmp));
        orc_ntt_inverse(tmp);
        for (uint32_t z = 0; z < N; z++) out[k * N + z] = orc_crt_compose(tmp[z], tmp[N + z]);
    }
}

/* src/poly.cpp:34-78: u64 accumulation without intermediate reduce (wraps like the reference) */
void orc_multiply(uint64_t *out, const uint64_t *a, const uint64_t *b, size_t rs, size_t ms, size_t cs) {
    for (size_t r = 0; r < rs; r++)
        for (size_t c = 0; c < cs; c++) {
            uint64_t *acc = out + (r * cs + c) * NTTP;
            memset(acc, 0, NTTP * sizeof(uint64_t));
            for (size_t m = 0; m < ms; m++) {
                const uint64_t *x = a + (r * ms + m) * NTTP, *y = b + (m * cs + c) * NTTP;
                for (uint32_t k = 0; k < NTTP; k++) acc[k] += x[k] * y[k];
            }
            for (uint32_t z = 0; z < N; z++) {
                acc[z] %= ORC_P;
                acc[N + z] %= ORC_B;
            }
        }
}

void orc_add(uint64_t *out, const uint64_t *a, const uint64_t *b, size_t npolys) {
#pragma omp parallel for if (g_threads > 1 && npolys > 8)
    for (size_t k = 0; k < npolys; k++)
        for (int n = 0; n < 2; n++)
            for (uint32_t z = 0; z < N; z++) {
                size_t i = k * NTTP + n * N + z;
                out[i] = red(a[i] + b[i], n);
            }
}

void orc_mul_by_const(uint64_t *out, const uint64_t *single, const uint64_t *a, size_t npolys) {
    for (size_t k = 0; k < npolys; k++)
        for (int n = 0; n < 2; n++)
            for (uint32_t z = 0; z < N; z++) {
                size_t i = k * NTTP + n * N + z;
                out[i] = red(a[i] * single[n * N + z], n);
            }
}

/* src/poly.cpp:240-261.  Negation is Q - a: a zero coefficient becomes Q, not 0. */
void orc_automorph(uint64_t *out, const uint64_t *in, size_t npolys, uint64_t t) {
    for (size_t k = 0; k < npolys; k++)
        for (uint64_t i = 0; i < N; i++) {
            uint64_t prod = i * t, wraps = prod / N, pos = prod % N;
            out[k * N + pos] = (wraps & 1) ? Q - in[k * N + i] : in[k * N + i];
        }
}

void orc_invert(uint64_t *out, const uint64_t *in, size_t npolys) { /* src/poly.cpp:269-283 */
    for (size_t i = 0; i < npolys * N; i++) out[i] = Q - in[i];
}

/* src/core.cpp:20-30 */
uint64_t orc_read_arbitrary_bits(const uint64_t *p, size_t bit_offs, size_t num_bits) {
    size_t word_off = bit_offs / 64, within = bit_offs % 64;
    if (within + num_bits <= 64) return (p[word_off] >> within) & ((1ULL << num_bits) - 1);
    unsigned __int128 val = (unsigned __int128)p[word_off] | ((unsigned __int128)p[word_off + 1] << 64);
    return (uint64_t)(val >> within) & ((1ULL << num_bits) - 1);
}
/* src/core.cpp:32-52 */
void orc_write_arbitrary_bits(uint64_t *p, uint64_t val, size_t bit_offs, size_t num_bits) {
    size_t word_off = bit_offs / 64, within = bit_offs % 64;
    val &= (1ULL << num_bits) - 1;
    if (within + num_bits <= 64) {
        p[word_off] &= ~(((1ULL << num_bits) - 1) << within);
        p[word_off] |= val << within;
    } else {
        unsigned __int128 cur = (unsigned __int128)p[word_off] | ((unsigned __int128)p[word_off + 1] << 64);
        cur &= ~((unsigned __int128)((1ULL << num_bits) - 1) << within);
        cur |= (unsigned __int128)val << within;
        p[word_off] = (uint64_t)cur;
        p[word_off + 1] = (uint64_t)(cur >> 64);
    }
}
static uint32_t wire_bits_rest(const orc_params *p) { /* bits that hold a value below 4 p_db (pt_mod + 2 for a power of two, spiral.cpp:232) */
    uint32_t b = 0;
    while ((1ULL << b) < 4 * p->p_db) b++;
    return b;
}
size_t orc_response_wire_bytes(const orc_params *p, uint32_t out_n) {
    return ((size_t)out_n * N * p->qprime_bits + (size_t)out_n * out_n * N * wire_bits_rest(p)) / 8;
}
/* the walk of modswitch, src/spiral.cpp:40-76, over an already switched response */
void orc_response_to_wire(const orc_params *p, uint32_t out_n, const uint64_t *resp, uint64_t *wire) {
    size_t rs = out_n + 1, cs = out_n, bit_offs = 0;
    memset(wire, 0, orc_response_wire_bytes(p, out_n) + 8);
    for (size_t r = 0; r < rs; r++) {
        size_t width = r == 0 ? p->qprime_bits : wire_bits_rest(p);
        for (size_t c = 0; c < cs; c++)
            for (size_t m = 0; m < N; m++) {
                orc_write_arbitrary_bits(wire, resp[(r * cs + c) * N + m], bit_offs, width);
                bit_offs += width;
            }
    }
}
/* load_modswitched_into_ct, src/client.cpp:90-110 */
void orc_response_from_wire(const orc_params *p, uint32_t out_n, const uint64_t *wire, uint64_t *resp) {
    size_t rs = out_n + 1, cs = out_n, bit_offs = 0;
    for (size_t r = 0; r < rs; r++) {
        size_t width = r == 0 ? p->qprime_bits : wire_bits_rest(p);
        for (size_t c = 0; c < cs; c++)
            for (size_t m = 0; m < N; m++) {
                resp[(r * cs + c) * N + m] = orc_read_arbitrary_bits(wire, bit_offs, width);
                bit_offs += width;
            }
    }
}

/* src/poly.cpp:578-591 */
uint64_t orc_rescale(uint64_t a, uint64_t inp_mod, uint64_t out_mod) {
    int64_t v = (int64_t)(a % inp_mod);
    if (v >= (int64_t)(inp_mod / 2)) v -= (int64_t)inp_mod;
    int64_t sign = v >= 0 ? 1 : -1;
    __int128 val = (__int128)v * (__int128)out_mod;
    __int128 res = (val + sign * (int64_t)(inp_mod / 2)) / (__int128)inp_mod;
    res = (res + (__int128)((inp_mod / out_mod) * out_mod) + (__int128)(2 * out_mod)) % (__int128)out_mod;
    return (uint64_t)((res + (__int128)out_mod) % (__int128)out_mod);
}

/* ------------------------------------------------------------------------------------------------ */
/* gadget                                                                                           */
/* ------------------------------------------------------------------------------------------------ */

uint32_t orc_get_bits_per(uint32_t dim) { /* include/util.h:34-38, logQ = 56 */
    if (dim == 56) return 1;
    return (uint32_t)floor(56.0 / (double)dim) + 1;
}

void orc_build_gadget(uint64_t *G, size_t rows, size_t cols) { /* src/util.cpp:89-106 */
    memset(G, 0, rows * cols * N * sizeof(uint64_t));
    size_t ne = cols / rows;
    uint32_t bits = orc_get_bits_per((uint32_t)ne);
    for (size_t i = 0; i < rows; i++)
        for (size_t j = 0; j < ne; j++) {
            if ((uint64_t)bits * j >= 64) continue;
            G[(i * cols + (i + j * rows)) * N] = 1ull << (bits * j);
        }
}

/* src/util.cpp:114-144.  A shift count >= 64 (UB in the reference, min(..,64)) is defined as 0. */
void orc_gadget_invert(uint64_t *out, const uint64_t *in, size_t mx, size_t rdim, size_t cols) {
    size_t ne = mx / rdim;
    uint32_t bits = orc_get_bits_per((uint32_t)ne);
    uint64_t mask = (1ull << bits) - 1;
    for (size_t c = 0; c < cols; c++)
        for (size_t j = 0; j < rdim; j++)
            for (uint32_t z = 0; z < N; z++) {
                uint64_t val = in[(j * cols + c) * N + z];
                for (size_t k = 0; k < ne; k++) {
                    uint64_t sh = (uint64_t)k * bits;
                    uint64_t piece = sh >= 64 ? 0 : ((val >> sh) & mask);
                    out[((j + k * rdim) * cols + c) * N + z] = piece;
                }
            }
}

/* ------------------------------------------------------------------------------------------------ */
/* server hot path                                                                                  */
/* ------------------------------------------------------------------------------------------------ */

/* src/spiral.cpp:270-341: balanced digits, two independent carry chains (first half: the last digit
 * of the half never borrows; second half: every digit may), then limb-reduce and forward NTT.
 * in:  raw [num_per][n1][n2][N];  out: NTT [num_per][m2][n2][2][N], row = r + k*n1 */
void orc_split_and_crt(uint64_t *out, const uint64_t *in, size_t num_per, uint32_t t_gsw) {
    uint32_t ell = t_gsw, m2 = t_gsw * N1, half = ell / 2;
    uint32_t bits = orc_get_bits_per(ell);
    uint64_t mask = (1ull << bits) - 1, base = 1ull << bits, thresh = base / 2;
#pragma omp parallel for collapse(3) if (g_threads > 1)
    for (size_t i = 0; i < num_per; i++)
        for (uint32_t r = 0; r < N1; r++)
            for (uint32_t c = 0; c < N2; c++) {
                const uint64_t *src = in + ((i * N1 + r) * N2 + c) * N;
                for (uint32_t z = 0; z < N; z++) {
                    uint64_t val = src[z], carry = 0;
                    for (uint32_t k = 0; k < ell; k++) {
                        if (k == half) carry = 0; /* second chain starts fresh (:311) */
                        uint64_t sh = (uint64_t)k * bits;
                        uint64_t piece = (sh >= 64 ? 0 : ((val >> sh) & mask)) + carry;
                        carry = 0;
                        int may_borrow = (k < half) ? (k + 1 < half) : 1;
                        if (piece > thresh && may_borrow) {
                            piece += Q - base;
                            carry = 1;
                        }
                        uint64_t *o = out + ((i * m2 + (r + k * N1)) * N2 + c) * NTTP;
                        o[z] = red(piece, 0);
                        o[N + z] = red(piece, 1);
                    }
                }
                for (uint32_t k = 0; k < ell; k++)
                    orc_ntt_forward(out + ((i * m2 + (r + k * N1)) * N2 + c) * NTTP);
            }
}

/* src/spiral.cpp:345-384: (i, m, c, n, z) -> packed (z, i, c, m) */
void orc_reorient_C(uint64_t *out, const uint64_t *in, size_t num_per, uint32_t m2) {
#pragma omp parallel for if (g_threads > 1)
    for (size_t i = 0; i < num_per; i++)
        for (uint32_t m = 0; m < m2; m++)
            for (uint32_t c = 0; c < N2; c++) {
                const uint64_t *p = in + ((i * m2 + m) * N2 + c) * NTTP;
                for (uint32_t z = 0; z < N; z++)
                    out[(size_t)z * (num_per * N2 * m2) + i * (N2 * m2) + c * m2 + m] = p[z] | (p[N + z] << 32);
            }
}

/* src/spiral.cpp:388-400: (r, m, n, z) -> packed (z, r, m) */
void orc_reorient_Q(uint64_t *out, const uint64_t *in, uint32_t m2) {
    for (uint32_t r = 0; r < N1; r++)
        for (uint32_t m = 0; m < m2; m++) {
            const uint64_t *p = in + (r * m2 + m) * NTTP;
            for (uint32_t z = 0; z < N; z++) out[(size_t)z * (N1 * m2) + r * m2 + m] = p[z] | (p[N + z] << 32);
        }
}

/* src/spiral.cpp:410-433: (j, r, m, n, z) -> packed (z, j, m, r padded to 4) */
void orc_reorient_ciphertexts(uint64_t *out, const uint64_t *in, size_t dim0) {
    memset(out, 0, dim0 * 2 * 4 * N * sizeof(uint64_t));
#pragma omp parallel for if (g_threads > 1)
    for (size_t j = 0; j < dim0; j++)
        for (uint32_t r = 0; r < N1; r++)
            for (uint32_t m = 0; m < 2; m++) {
                const uint64_t *p = in + ((j * N1 + r) * 2 + m) * NTTP;
                for (uint32_t z = 0; z < N; z++)
                    out[(size_t)z * (dim0 * 2 * 4) + j * 8 + m * 4 + r] = p[z] | (p[N + z] << 32);
            }
}

/* src/spiral.cpp:628-999, scalar semantics (:932-998): out[i][r][c][n][z] =
 * (sum_{j,m} ct[z][j][m][r].n * db[z][i][c][j][m].n) mod m_n */
static void sweep_slabs(uint64_t *out, const uint64_t *cts, const uint64_t *db, size_t dim0, size_t num_per, uint32_t nz, size_t N_out);
void orc_multiply_query_by_database(uint64_t *out, const uint64_t *cts, const uint64_t *db, size_t dim0,
                                    size_t num_per) {
    sweep_slabs(out, cts, db, dim0, num_per, N, N);
}
/* the same for nz slabs only (the sweep is independent per NTT slot z): cts and db hold the nz slabs of the chosen slots in
 * the reference's z-major layouts, out is [num_per][n1][n2][2][nz].  Lets a test check a 32 or 64 GiB database on a sample
 * of slots. */
void orc_multiply_query_by_database_slots(uint64_t *out, const uint64_t *cts, const uint64_t *db, size_t dim0, size_t num_per,
                                          uint32_t nz) {
    sweep_slabs(out, cts, db, dim0, num_per, nz, nz);
}
/* one (z, i, c) cell: the six sums over jm < 2 dim0 of a[jm][r] * b[jm], per CRT limb, reduced.  Scalar form (:932-998). */
static inline void sweep_cell_scalar(const uint64_t *a, const uint64_t *b, size_t jm_total, uint64_t s0_out[3], uint64_t s1_out[3]) {
    u128 s0[3] = {0, 0, 0}, s1[3] = {0, 0, 0};
    for (size_t jm = 0; jm < jm_total; jm++) {
        uint64_t bw = b[jm], blo = (uint32_t)bw, bhi = bw >> 32;
        for (uint32_t r = 0; r < 3; r++) {
            uint64_t aw = a[jm * 4 + r];
            s0[r] += (uint64_t)(uint32_t)aw * blo;
            s1[r] += (aw >> 32) * bhi;
        }
    }
    for (uint32_t r = 0; r < 3; r++) {
        s0_out[r] = (uint64_t)(s0[r] % ORC_P);
        s1_out[r] = (uint64_t)(s1[r] % ORC_B);
    }
}
/* The reference's vectorised forms of the same cell, so that the CPU baseline bench.py times is the loop the reference
 * runs on such a host rather than gcc's reading of the scalar one: AVX-512 src/spiral.cpp:640-745 (two jm per 512-bit
 * vector: lanes 0..3 = rows of jm, 4..7 = rows of jm + 1, _mm512_mul_epu32 on the low and on the shifted-down high halves,
 * a partial reduction every max_summed_pa_or_b_in_u64 = 64 terms, include/values.h:56), AVX2 :746-886 (one jm per 256-bit
 * vector).  Operands must be reduced residues (< 2^28 here), as the reference's are.  tests/test_oracle.py proves these
 * equal to the scalar cell, extremes included. */
#if defined(__AVX512F__)
#include <immintrin.h>
#define ORC_SWEEP_ISA "avx512"
static inline void sweep_cell_simd(const uint64_t *a, const uint64_t *b, size_t jm_total, uint64_t s0_out[3], uint64_t s1_out[3]) {
    size_t inner = 64, outer = jm_total / inner;
    if (jm_total < 64) {
        inner = jm_total;
        outer = 1;
    }
    uint64_t acc0[3] = {0, 0, 0}, acc1[3] = {0, 0, 0};
    for (size_t o = 0; o < outer; o++) {
        __m512i n0 = _mm512_setzero_si512(), n2 = _mm512_setzero_si512();
#pragma GCC unroll 16
        for (size_t k = 0; k < inner / 2; k++) {
            const size_t jm = o * inner + 2 * k;
            const __m512i bv = _mm512_mask_blend_epi64(0xF0, _mm512_set1_epi64((long long)b[jm]), _mm512_set1_epi64((long long)b[jm + 1]));
            const __m512i av = _mm512_loadu_si512((const void *)(a + jm * 4));
            n0 = _mm512_add_epi64(n0, _mm512_mul_epu32(av, bv));
            n2 = _mm512_add_epi64(n2, _mm512_mul_epu32(_mm512_srli_epi64(av, 32), _mm512_srli_epi64(bv, 32)));
        }
        uint64_t t0[8], t2[8];
        _mm512_storeu_si512((void *)t0, n0);
        _mm512_storeu_si512((void *)t2, n2);
        for (int r = 0; r < 3; r++) {
            acc0[r] = (acc0[r] + t0[r] + t0[4 + r]) % ORC_P;
            acc1[r] = (acc1[r] + t2[r] + t2[4 + r]) % ORC_B;
        }
    }
    for (int r = 0; r < 3; r++) {
        s0_out[r] = acc0[r];
        s1_out[r] = acc1[r];
    }
}
#elif defined(__AVX2__)
#include <immintrin.h>
#define ORC_SWEEP_ISA "avx2"
static inline void sweep_cell_simd(const uint64_t *a, const uint64_t *b, size_t jm_total, uint64_t s0_out[3], uint64_t s1_out[3]) {
    size_t inner = 64, outer = jm_total / inner;
    if (jm_total < 64) {
        inner = jm_total;
        outer = 1;
    }
    uint64_t acc0[3] = {0, 0, 0}, acc1[3] = {0, 0, 0};
    for (size_t o = 0; o < outer; o++) {
        __m256i n0 = _mm256_setzero_si256(), n2 = _mm256_setzero_si256();
#pragma GCC unroll 16
        for (size_t k = 0; k < inner; k++) {
            const size_t jm = o * inner + k;
            const __m256i bv = _mm256_set1_epi64x((long long)b[jm]);
            const __m256i av = _mm256_loadu_si256((const __m256i *)(a + jm * 4));
            n0 = _mm256_add_epi64(n0, _mm256_mul_epu32(av, bv));
            n2 = _mm256_add_epi64(n2, _mm256_mul_epu32(_mm256_srli_epi64(av, 32), _mm256_srli_epi64(bv, 32)));
        }
        uint64_t t0[4], t2[4];
        _mm256_storeu_si256((__m256i *)t0, n0);
        _mm256_storeu_si256((__m256i *)t2, n2);
        for (int r = 0; r < 3; r++) {
            acc0[r] = (acc0[r] + t0[r]) % ORC_P;
            acc1[r] = (acc1[r] + t2[r]) % ORC_B;
        }
    }
    for (int r = 0; r < 3; r++) {
        s0_out[r] = acc0[r];
        s1_out[r] = acc1[r];
    }
}
#else
#define ORC_SWEEP_ISA "scalar"
#define sweep_cell_simd sweep_cell_scalar
#endif
const char *orc_sweep_isa(void) { return ORC_SWEEP_ISA; }

static void sweep_slabs_impl(uint64_t *out, const uint64_t *cts, const uint64_t *db, size_t dim0, size_t num_per, uint32_t nz, size_t N_out, int scalar) {
    const size_t NTTP_out = 2 * N_out, jm_total = dim0 * 2;
    /* the vector cells step through jm in blocks of 64 (2 for the AVX-512 pairing): other lengths take the scalar cell */
    const int simd_ok = !scalar && (jm_total % 64 == 0 || (jm_total < 64 && jm_total % 2 == 0));
#pragma omp parallel for if (g_threads > 1)
    for (uint32_t z = 0; z < nz; z++) {
        const uint64_t *a = cts + (size_t)z * (dim0 * 2 * 4);
        const uint64_t *bz = db + (size_t)z * (num_per * N2 * dim0 * N0);
        for (size_t i = 0; i < num_per; i++)
            for (uint32_t c = 0; c < N2; c++) {
                const uint64_t *b = bz + (i * N2 + c) * (dim0 * N0);
                uint64_t s0[3], s1[3];
                if (simd_ok)
                    sweep_cell_simd(a, b, jm_total, s0, s1);
                else
                    sweep_cell_scalar(a, b, jm_total, s0, s1);
                for (uint32_t r = 0; r < 3; r++) {
                    uint64_t *o = out + ((i * N1 + r) * N2 + c) * NTTP_out;
                    o[z] = s0[r];
                    o[N_out + z] = s1[r];
                }
            }
    }
}
static void sweep_slabs(uint64_t *out, const uint64_t *cts, const uint64_t *db, size_t dim0, size_t num_per, uint32_t nz, size_t N_out) {
    sweep_slabs_impl(out, cts, db, dim0, num_per, nz, N_out, 0);
}
/* the scalar cell only (tests: the vectorised cells against it) */
void orc_multiply_query_by_database_scalar(uint64_t *out, const uint64_t *cts, const uint64_t *db, size_t dim0, size_t num_per) {
    sweep_slabs_impl(out, cts, db, dim0, num_per, N, N, 1);
}

/* src/spiral.cpp:464-582: Cn[i][r][c][n][z] = (sum_m Q[z][r][m].n * C[z][i][c][m].n) mod m_n, u64 sums */
void orc_cpu_mul_query_by_ct(uint64_t *c_next, const uint64_t *q, const uint64_t *cm, size_t num_per,
                             uint32_t m2) {
#pragma omp parallel for if (g_threads > 1)
    for (uint32_t z = 0; z < N; z++)
        for (size_t i = 0; i < num_per; i++)
            for (uint32_t r = 0; r < N1; r++)
                for (uint32_t c = 0; c < N2; c++) {
                    const uint64_t *cp = cm + (size_t)z * (num_per * N2 * m2) + i * (N2 * m2) + c * m2;
                    const uint64_t *qp = q + (size_t)z * (N1 * m2) + r * m2;
                    uint64_t s0 = 0, s1 = 0;
                    for (uint32_t m = 0; m < m2; m++) {
                        s0 += (uint64_t)(uint32_t)qp[m] * (uint32_t)cp[m];
                        s1 += (qp[m] >> 32) * (cp[m] >> 32);
                    }
                    uint64_t *o = c_next + ((i * N1 + r) * N2 + c) * NTTP;
                    o[z] = red(s0, 0);
                    o[N + z] = red(s1, 1);
                }
}

/* src/spiral.cpp:1349-1410 */
void orc_fold_one_further_dimension(uint64_t *cts, size_t num_per, const uint64_t *q, const uint64_t *q_neg,
                                    uint32_t t_gsw) {
    uint32_t m2 = t_gsw * N1;
    size_t ct_raw = (size_t)N1 * N2 * N, npoly = num_per * N1 * N2;
    uint64_t *big1 = malloc(num_per * m2 * N2 * NTTP * sizeof(uint64_t));
    uint64_t *big2 = malloc(num_per * m2 * N2 * N * sizeof(uint64_t));
    uint64_t *hi = malloc(npoly * NTTP * sizeof(uint64_t));
    uint64_t *lo = malloc(npoly * NTTP * sizeof(uint64_t));
    orc_split_and_crt(big1, cts + num_per * ct_raw, num_per, t_gsw);
    orc_reorient_C(big2, big1, num_per, m2);
    orc_cpu_mul_query_by_ct(hi, q, big2, num_per, m2);
    orc_split_and_crt(big1, cts, num_per, t_gsw);
    orc_reorient_C(big2, big1, num_per, m2);
    orc_cpu_mul_query_by_ct(lo, q_neg, big2, num_per, m2);
    orc_add(lo, lo, hi, npoly);
    orc_from_ntt(cts, lo, npoly); /* ntt_inverse + cpu_crt (:1386-1407) */
    free(big1);
    free(big2);
    free(hi);
    free(lo);
}

/* neg1s_mp[r] = to_ntt(invert(x^(N - 2^r))) (src/spiral.cpp:171-190) */
static void make_neg1(uint64_t *out_ntt, uint32_t r) {
    uint64_t *raw = calloc(N, sizeof(uint64_t)), *inv = malloc(N * sizeof(uint64_t));
    raw[N - (1u << r)] = 1;
    orc_invert(inv, raw, 1);
    orc_to_ntt(out_ntt, inv, 1);
    free(raw);
    free(inv);
}

/* src/spiral.cpp:1664-1743 */
void orc_expand_improved(uint64_t *cv, uint32_t g, uint32_t t_exp, const uint64_t *w_left, uint32_t t_exp_right,
                         const uint64_t *w_right, uint32_t n_right, uint32_t max_bits_right, uint32_t stopround) {
    uint32_t tmax = t_exp > t_exp_right ? t_exp : t_exp_right;
    const size_t CT = (size_t)N0 * NTTP; /* one n0 x 1 NTT ciphertext */
    uint64_t *neg1 = malloc(NTTP * sizeof(uint64_t));
    /* scratch of one iteration: c, ca (N0*N each), ca1_ntt (NTTP), gi (tmax*N), gi_ntt (tmax*NTTP), wg (CT); one set per thread */
    const size_t per = (size_t)2 * N0 * N + NTTP + (size_t)tmax * N + (size_t)tmax * NTTP + CT;
    uint64_t *scratch = malloc(per * (size_t)g_threads * sizeof(uint64_t));
    for (uint32_t r = 0; r < g; r++) {
        uint32_t num_in = 1u << r, num_out = 2 * num_in;
        uint64_t t = (N >> r) + 1;
        make_neg1(neg1, r);
        /* the reference creates cv[num_in + i] = neg1 * cv[i] at the top of iteration i < num_in (:1709), before that
         * iteration updates cv[i] and before iteration num_in + i reads it; done up front here so that the iterations of a
         * round are independent (same values) */
        for (uint32_t i = 0; i < num_in; i++) {
            /* ... under the same skip predicate as the iteration itself (:1701-1702 come before :1709): an odd ciphertext the
             * round skips creates nothing, so every cv slot holds what the reference leaves there */
            if (stopround > 0 && r > stopround && (i & 1)) continue;
            if (stopround > 0 && r == stopround && (i & 1) && i / 2 > max_bits_right) continue;
            orc_mul_by_const(cv + (size_t)(num_in + i) * CT, neg1, cv + (size_t)i * CT, N0);
        }
#pragma omp parallel for schedule(dynamic, 1) if (g_threads > 1)
        for (uint32_t i = 0; i < num_out; i++) {
            int odd = i & 1;
            if (stopround > 0 && r > stopround && odd) continue;
            if (stopround > 0 && r == stopround && odd && i / 2 > max_bits_right) continue;
            uint32_t gdim = odd ? t_exp_right : t_exp;
            const uint64_t *W;
            if (odd) {
                if (r >= n_right) abort(); /* the reference would read out of bounds here */
                W = w_right + (size_t)r * N0 * t_exp_right * NTTP;
            } else {
                W = w_left + (size_t)r * N0 * t_exp * NTTP;
            }
            int tid = 0;
#ifdef _OPENMP
            tid = omp_get_thread_num();
#endif
            uint64_t *c = scratch + per * (size_t)tid, *ca = c + N0 * N, *ca1_ntt = ca + N0 * N, *gi = ca1_ntt + NTTP;
            uint64_t *gi_ntt = gi + (size_t)tmax * N, *wg = gi_ntt + (size_t)tmax * NTTP;
            uint64_t *cvi = cv + (size_t)i * CT;
            orc_from_ntt(c, cvi, N0);
            orc_automorph(ca, c, N0, t);
            orc_to_ntt(ca1_ntt, ca + N, 1);
            orc_gadget_invert(gi, ca, gdim, 1, 1);
            orc_to_ntt_no_reduce(gi_ntt, gi, gdim);
            orc_multiply(wg, W, gi_ntt, N0, gdim, 1);
            for (uint32_t j = 0; j < N0; j++)
                for (int n = 0; n < 2; n++)
                    for (uint32_t z = 0; z < N; z++) {
                        size_t k = (size_t)j * NTTP + n * N + z;
                        cvi[k] = red(cvi[k] + wg[k] + j * ca1_ntt[n * N + z], n);
                    }
        }
    }
    free(neg1); free(scratch);
}

/* prod(n1 x n0) = W (n1 x 2*t_conv) * special_distribute(g) (src/spiral.cpp:1834-1848):
 * column c of the product only sees W's columns 2k+c */
static void w_times_distributed(uint64_t *prod, const uint64_t *w, const uint64_t *g_ntt, uint32_t t_conv) {
    uint64_t *dist = calloc((size_t)2 * t_conv * N0 * NTTP, sizeof(uint64_t));
    for (uint32_t k = 0; k < t_conv; k++) {
        memcpy(dist + ((size_t)(2 * k) * N0 + 0) * NTTP, g_ntt + (size_t)k * NTTP, NTTP * sizeof(uint64_t));
        memcpy(dist + ((size_t)(2 * k + 1) * N0 + 1) * NTTP, g_ntt + (size_t)k * NTTP, NTTP * sizeof(uint64_t));
    }
    orc_multiply(prod, w, dist, N1, 2 * t_conv, N0);
    free(dist);
}

/* out(3x2) = prod + pad(cv row 1) at (1,0) and (2,1) (src/spiral.cpp:1875-1884, 1909-1915) */
static void add_padded_cv1(uint64_t *out, const uint64_t *prod, const uint64_t *cv) {
    uint64_t *pad = calloc((size_t)N1 * N0 * NTTP, sizeof(uint64_t));
    memcpy(pad + (1 * N0 + 0) * NTTP, cv + NTTP, NTTP * sizeof(uint64_t));
    memcpy(pad + (2 * N0 + 1) * NTTP, cv + NTTP, NTTP * sizeof(uint64_t));
    orc_add(out, prod, pad, N1 * N0);
    free(pad);
}

/* src/spiral.cpp:1850-1885 */
void orc_scal_to_mat(uint64_t *out, const uint64_t *cv, const uint64_t *w, uint32_t t_conv) {
    uint64_t *raw = malloc(N * sizeof(uint64_t));
    uint64_t *gi = malloc((size_t)t_conv * N * sizeof(uint64_t));
    uint64_t *gi_ntt = malloc((size_t)t_conv * NTTP * sizeof(uint64_t));
    uint64_t *prod = malloc((size_t)N1 * N0 * NTTP * sizeof(uint64_t));
    orc_from_ntt(raw, cv, 1);
    orc_gadget_invert(gi, raw, t_conv, 1, 1);
    orc_to_ntt_no_reduce(gi_ntt, gi, t_conv);
    w_times_distributed(prod, w, gi_ntt, t_conv);
    add_padded_cv1(out, prod, cv);
    free(raw); free(gi); free(gi_ntt); free(prod);
}

/* src/spiral.cpp:1985-2025 */
void orc_regev_to_gsw(uint64_t *out, const uint64_t *cv_v, const uint64_t *w, const uint64_t *v, uint32_t t_conv,
                      uint32_t ell) {
    const size_t CT = (size_t)N0 * NTTP;
    uint32_t cols = N1 * ell;
    uint64_t *raw = malloc(N0 * N * sizeof(uint64_t));
    uint64_t *gi = malloc((size_t)t_conv * N * sizeof(uint64_t));
    uint64_t *gi_ntt = malloc((size_t)t_conv * NTTP * sizeof(uint64_t));
    uint64_t *chat = calloc((size_t)2 * t_conv * ell * NTTP, sizeof(uint64_t)); /* 2*t_conv x ell */
    uint64_t *s2m = malloc((size_t)N1 * N0 * NTTP * sizeof(uint64_t));
    uint64_t *prod = malloc((size_t)N1 * N0 * NTTP * sizeof(uint64_t));
    uint64_t *res = calloc((size_t)N1 * cols * NTTP, sizeof(uint64_t));
    uint64_t *vprod = malloc((size_t)N1 * ell * NTTP * sizeof(uint64_t));
    for (uint32_t i = 0; i < ell; i++) {
        const uint64_t *cvi = cv_v + (size_t)i * CT;
        orc_from_ntt(raw, cvi, N0);
        orc_gadget_invert(gi, raw, t_conv, 1, 1);
        orc_to_ntt_no_reduce(gi_ntt, gi, t_conv);
        for (uint32_t k = 0; k < t_conv; k++)
            memcpy(chat + ((size_t)k * ell + i) * NTTP, gi_ntt + (size_t)k * NTTP, NTTP * sizeof(uint64_t));
        w_times_distributed(prod, w, gi_ntt, t_conv); /* scalToMatFast :1887-1916 */
        add_padded_cv1(s2m, prod, cvi);
        for (uint32_t r = 0; r < N1; r++)
            for (uint32_t c = 0; c < N0; c++)
                memcpy(res + ((size_t)r * cols + ell + N0 * i + c) * NTTP, s2m + ((size_t)r * N0 + c) * NTTP,
                       NTTP * sizeof(uint64_t));
        orc_gadget_invert(gi, raw + N, t_conv, 1, 1);
        orc_to_ntt_no_reduce(gi_ntt, gi, t_conv);
        for (uint32_t k = 0; k < t_conv; k++)
            memcpy(chat + ((size_t)(t_conv + k) * ell + i) * NTTP, gi_ntt + (size_t)k * NTTP, NTTP * sizeof(uint64_t));
    }
    orc_multiply(vprod, v, chat, N1, 2 * t_conv, ell);
    for (uint32_t r = 0; r < N1; r++)
        for (uint32_t i = 0; i < ell; i++)
            memcpy(res + ((size_t)r * cols + i) * NTTP, vprod + ((size_t)r * ell + i) * NTTP, NTTP * sizeof(uint64_t));
    /* column permutation (:2019-2022) */
    for (uint32_t r = 0; r < N1; r++)
        for (uint32_t i = 0; i < ell; i++) {
            memcpy(out + ((size_t)r * cols + (N0 + 1) * i) * NTTP, res + ((size_t)r * cols + i) * NTTP,
                   NTTP * sizeof(uint64_t));
            for (uint32_t c = 0; c < N0; c++)
                memcpy(out + ((size_t)r * cols + (N0 + 1) * i + 1 + c) * NTTP,
                       res + ((size_t)r * cols + ell + N0 * i + c) * NTTP, NTTP * sizeof(uint64_t));
        }
    free(raw); free(gi); free(gi_ntt); free(chat); free(s2m); free(prod); free(res); free(vprod);
}

/* ------------------------------------------------------------------------------------------------ */
/* staged pipeline                                                                                  */
/* ------------------------------------------------------------------------------------------------ */

static uint32_t ceil_log2(uint64_t x) {
    uint32_t r = 0;
    while ((1ull << r) < x) r++;
    return r;
}

int orc_get_shape(const orc_params *p, orc_shape *s) { /* src/spiral.cpp:2046-2085 */
    if (!p || !s || p->nu1 > 16 || p->nu2 > 16 || p->t_gsw < 2 || p->t_conv < 1) return -1;
    if (p->qprime_bits >= 37 || QPRIME_MODS[p->qprime_bits] == 0) return -1;
    s->dim0 = 1u << p->nu1;
    s->num_per = 1u << p->nu2;
    s->ell = p->t_gsw;
    s->m2 = p->t_gsw * N1;
    s->n_bits = s->dim0 + s->ell * p->nu2;
    s->qprime = QPRIME_MODS[p->qprime_bits];
    if (p->direct_upload) {
        s->g = 0; s->stopround = 0; s->n_left = 0; s->n_right = 0;
        s->n_query_cts = s->n_bits;
    } else {
        s->g = ceil_log2(s->n_bits);
        s->stopround = ceil_log2((uint64_t)s->ell * p->nu2);
        if (s->ell * p->nu2 > s->dim0) s->stopround = 0;
        if (p->nu2 == 0) s->stopround = 0;
        s->n_left = s->g;
        s->n_right = s->stopround > 0 ? s->stopround + 1 : s->g;
        s->n_query_cts = 1;
        if (s->g > LOGN) return -1;
    }
    return 0;
}

int orc_stage_expand(const orc_params *p, const uint64_t *query, const uint64_t *w_left, const uint64_t *w_right,
                     uint64_t *cv_out) {
    orc_shape s;
    if (orc_get_shape(p, &s)) return -1;
    const size_t CT = (size_t)N0 * NTTP;
    if (p->direct_upload) { /* :2163-2175: cts arrive already expanded */
        memcpy(cv_out, query, (size_t)s.n_bits * CT * sizeof(uint64_t));
        return 0;
    }
    size_t n = (size_t)1 << s.g;
    uint64_t *cv = calloc(n * CT, sizeof(uint64_t));
    memcpy(cv, query, CT * sizeof(uint64_t));
    orc_expand_improved(cv, s.g, p->t_exp, w_left, p->t_exp_right, w_right, s.n_right, s.ell * p->nu2, s.stopround);
    if (s.stopround != 0) { /* reorderFromStopround :2027-2036 */
        for (uint32_t i = 0; i < s.dim0; i++) memcpy(cv_out + (size_t)i * CT, cv + (size_t)(2 * i) * CT, CT * 8);
        for (uint32_t i = 0; i < s.ell * p->nu2; i++)
            memcpy(cv_out + (size_t)(s.dim0 + i) * CT, cv + (size_t)(2 * i + 1) * CT, CT * 8);
    } else {
        memcpy(cv_out, cv, (size_t)s.n_bits * CT * 8);
    }
    free(cv);
    return 0;
}

int orc_stage_convert(const orc_params *p, const uint64_t *cv, const uint64_t *w, const uint64_t *v,
                      uint64_t *cts_out, uint64_t *gsw_out) {
    orc_shape s;
    if (orc_get_shape(p, &s)) return -1;
    const size_t CT = (size_t)N0 * NTTP;
#pragma omp parallel for if (g_threads > 1)
    for (uint32_t i = 0; i < s.dim0; i++) /* :2230-2253 */
        orc_scal_to_mat(cts_out + (size_t)i * N1 * N0 * NTTP, cv + (size_t)i * CT, w, p->t_conv);
#pragma omp parallel for if (g_threads > 1)
    for (uint32_t i = 0; i < p->nu2; i++) /* :2315-2331, stored reversed */
        orc_regev_to_gsw(gsw_out + (size_t)(p->nu2 - 1 - i) * N1 * s.m2 * NTTP, cv + (size_t)(s.dim0 + i * s.ell) * CT,
                         w, v, p->t_conv, s.ell);
    return 0;
}

int orc_stage_first_dim(const orc_params *p, const uint64_t *cts, const uint64_t *db, uint64_t *raw_out) {
    orc_shape s;
    if (orc_get_shape(p, &s)) return -1;
    size_t npoly = (size_t)s.num_per * N1 * N2;
    uint64_t *re = malloc((size_t)s.dim0 * 2 * 4 * N * sizeof(uint64_t));
    uint64_t *acc = malloc(npoly * NTTP * sizeof(uint64_t));
    orc_reorient_ciphertexts(re, cts, s.dim0);
    orc_multiply_query_by_database(acc, re, db, s.dim0, s.num_per);
    orc_from_ntt(raw_out, acc, npoly); /* nttInvAndCrtLiftCiphertexts :437-453 */
    free(re);
    free(acc);
    return 0;
}

int orc_stage_fold(const orc_params *p, uint64_t *raw_cts, const uint64_t *gsw, uint64_t *final_out) {
    orc_shape s;
    if (orc_get_shape(p, &s)) return -1;
    size_t qwords = (size_t)N1 * s.m2 * NTTP, qpolys = (size_t)N1 * s.m2;
    uint64_t *g2 = malloc(qpolys * N * sizeof(uint64_t));
    uint64_t *qraw = malloc(qpolys * N * sizeof(uint64_t));
    uint64_t *qneg_ntt = malloc(qwords * sizeof(uint64_t));
    uint64_t *qre = malloc((size_t)p->nu2 * qpolys * N * sizeof(uint64_t) + 8);
    uint64_t *qnre = malloc((size_t)p->nu2 * qpolys * N * sizeof(uint64_t) + 8);
    orc_build_gadget(g2, N1, s.m2);
    for (uint32_t d = 0; d < p->nu2; d++) { /* :2361-2386 */
        const uint64_t *qn = gsw + (size_t)d * qwords;
        orc_from_ntt(qraw, qn, qpolys);
        for (size_t k = 0; k < qpolys * N; k++) {
            int64_t val = (int64_t)g2[k] - (int64_t)qraw[k];
            if (val < 0) val += (int64_t)Q;
            qraw[k] = (uint64_t)val;
        }
        orc_to_ntt(qneg_ntt, qraw, qpolys); /* cpu_crt_to_ucompressed_and_ntt :597-609 */
        orc_reorient_Q(qre + (size_t)d * qpolys * N, qn, s.m2);
        orc_reorient_Q(qnre + (size_t)d * qpolys * N, qneg_ntt, s.m2);
    }
    size_t num_per = s.num_per;
    uint32_t cur = 0;
    while (num_per >= 2) { /* :1622-1626 */
        num_per /= 2;
        orc_fold_one_further_dimension(raw_cts, num_per, qre + (size_t)cur * qpolys * N, qnre + (size_t)cur * qpolys * N,
                                       p->t_gsw);
        cur++;
    }
    memcpy(final_out, raw_cts, (size_t)N1 * N2 * N * sizeof(uint64_t));
    free(g2); free(qraw); free(qneg_ntt); free(qre); free(qnre);
    return 0;
}

int orc_stage_rescale(const orc_params *p, const uint64_t *final_ct, uint64_t *resp) { /* :1441-1447 */
    orc_shape s;
    if (orc_get_shape(p, &s)) return -1;
    uint64_t q1 = 4 * p->p_db;
    for (uint32_t r = 0; r < N1; r++)
        for (size_t k = 0; k < (size_t)N2 * N; k++) {
            uint64_t a = final_ct[(size_t)r * N2 * N + k] % Q;
            resp[(size_t)r * N2 * N + k] = orc_rescale(a, Q, r == 0 ? s.qprime : q1);
        }
    return 0;
}

int orc_answer(const orc_params *p, const uint64_t *query, const uint64_t *w_left, const uint64_t *w_right,
               const uint64_t *w, const uint64_t *v, const uint64_t *db, uint64_t *final_out) {
    orc_shape s;
    if (orc_get_shape(p, &s)) return -1;
    const size_t CT = (size_t)N0 * NTTP;
    uint64_t *cv = malloc((size_t)s.n_bits * CT * 8);
    uint64_t *cts = malloc((size_t)s.dim0 * N1 * N0 * NTTP * 8);
    uint64_t *gsw = malloc(((size_t)p->nu2 * N1 * s.m2 * NTTP + 8) * 8);
    uint64_t *raw = malloc((size_t)s.num_per * N1 * N2 * N * 8);
    orc_stage_expand(p, query, w_left, w_right, cv);
    orc_stage_convert(p, cv, w, v, cts, gsw);
    orc_stage_first_dim(p, cts, db, raw);
    orc_stage_fold(p, raw, gsw, final_out);
    free(cv); free(cts); free(gsw); free(raw);
    return 0;
}

/* ------------------------------------------------------------------------------------------------ */
/* DB producer                                                                                      */
/* ------------------------------------------------------------------------------------------------ */

static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

uint64_t orc_db_coeff(uint64_t seed, uint64_t item, uint64_t k, uint64_t p_db) {
    return splitmix64(seed ^ (item * (4ull * N) + k)) % p_db;
}

void orc_db_item(const orc_params *p, uint64_t seed, uint64_t item, uint64_t *pt) {
    for (uint64_t k = 0; k < 4ull * N; k++) pt[k] = orc_db_coeff(seed, item, k, p->p_db);
}

/* one plaintext (n0 x n2 raw, coefficients in [0, p_db)) -> pts_encd: centred lift (:1116-1127) + to_ntt (:1128) */
void orc_encode_item(const orc_params *p, const uint64_t *pt, uint64_t *enc) {
    uint64_t lifted[4 * N];
    for (uint32_t k = 0; k < 4 * N; k++) {
        int64_t v = (int64_t)pt[k];
        if (v >= (int64_t)(p->p_db / 2)) v -= (int64_t)p->p_db;
        if (v < 0) v += (int64_t)Q;
        lifted[k] = (uint64_t)v;
    }
    orc_to_ntt(enc, lifted, 4);
}

/* src/spiral.cpp:1083-1171: centred lift (:1116-1127), to_ntt, packed word at
 * z*(num_per*n2*dim0*n0) + ii*(n2*dim0*n0) + c*(dim0*n0) + j*n0 + m, item i -> (ii = i % num_per, j = i / num_per) */
void orc_gen_db(const orc_params *p, uint64_t seed, uint64_t *db) {
    orc_shape s;
    if (orc_get_shape(p, &s)) return;
    uint64_t total = (uint64_t)s.dim0 * s.num_per;
#pragma omp parallel for if (g_threads > 1)
    for (uint64_t i = 0; i < total; i++) {
        uint64_t pt[4 * N], enc[4 * NTTP];
        orc_db_item(p, seed, i, pt);
        for (uint32_t k = 0; k < 4 * N; k++) {
            int64_t v = (int64_t)pt[k];
            if (v >= (int64_t)(p->p_db / 2)) v -= (int64_t)p->p_db;
            if (v < 0) v += (int64_t)Q;
            pt[k] = (uint64_t)v;
        }
        orc_to_ntt(enc, pt, 4);
        uint64_t ii = i % s.num_per, j = i / s.num_per;
        for (uint32_t m = 0; m < N0; m++)
            for (uint32_t c = 0; c < N2; c++) {
                const uint64_t *e = enc + (size_t)(m * N2 + c) * NTTP;
                for (uint32_t z = 0; z < N; z++) {
                    size_t idx = (size_t)z * ((size_t)s.num_per * N2 * s.dim0 * N0) + ii * ((size_t)N2 * s.dim0 * N0) +
                                 (size_t)c * (s.dim0 * N0) + j * N0 + m;
                    db[idx] = e[z] | (e[N + z] << 32);
                }
            }
    }
}

void orc_fill_db_random(uint64_t seed, uint64_t *db, size_t nwords) {
#pragma omp parallel for if (g_threads > 1)
    for (size_t i = 0; i < nwords; i++) {
        uint64_t r = splitmix64(seed + i);
        db[i] = ((r & 0xffffffffull) % ORC_P) | (((r >> 32) % ORC_B) << 32);
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* client restatement (inputs for the server path + the "Is correct?" decode)                       */
/* ------------------------------------------------------------------------------------------------ */

struct orc_client {
    orc_params p;
    orc_shape s;
    uint64_t rng[4];
    int nonoise;
    double cdf[129];
    uint64_t sr[N];      /* 1x1 raw: Regev secret (client.cpp:25-28) */
    uint64_t sp[N0 * N]; /* n0 x 1 raw: matrix-Regev secret (client.cpp:30-38) */
};

static uint64_t rng_next(orc_client *c) { /* xoshiro256** */
    uint64_t *s = c->rng;
    uint64_t r = ((s[1] * 5) << 7 | (s[1] * 5) >> 57) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t;
    s[3] = (s[3] << 45) | (s[3] >> 19);
    return r;
}

/* discrete Gaussian, width 6.4 over [-64,64] (src/core.cpp:182-207); returned mod Q (client.cpp:7-10) */
static uint64_t sample_noise(orc_client *c) {
    if (c->nonoise) return 0;
    double u = (double)(rng_next(c) >> 11) * (1.0 / 9007199254740992.0) * c->cdf[128];
    int lo = 0, hi = 128;
    while (lo < hi) {
        int mid = (lo + hi) / 2;
        if (c->cdf[mid] > u) hi = mid; else lo = mid + 1;
    }
    int64_t v = (int64_t)lo - 64;
    return (uint64_t)((v + (int64_t)Q) % (int64_t)Q);
}

static void fill_noise(orc_client *c, uint64_t *raw, size_t npolys) {
    for (size_t i = 0; i < npolys * N; i++) raw[i] = sample_noise(c);
}
static void fill_uniform(orc_client *c, uint64_t *raw, size_t npolys) { /* util.cpp:80-86 */
    for (size_t i = 0; i < npolys * N; i++) raw[i] = rng_next(c) % Q;
}

orc_client *orc_client_new(const orc_params *p, uint64_t seed, int nonoise) {
    orc_client *c = calloc(1, sizeof(*c));
    c->p = *p;
    if (orc_get_shape(p, &c->s)) { free(c); return NULL; }
    for (int i = 0; i < 4; i++) c->rng[i] = splitmix64(seed + 0x1234567 * (i + 1));
    c->nonoise = nonoise;
    double acc = 0;
    for (int i = -64; i <= 64; i++) {
        acc += exp(-M_PI * (double)i * i / (6.4 * 6.4));
        c->cdf[i + 64] = acc;
    }
    fill_noise(c, c->sr, 1);      /* keygen, client.cpp:21-46 */
    fill_noise(c, c->sp, N0);
    return c;
}

void orc_client_free(orc_client *c) { free(c); }

size_t orc_words_w_left(const orc_params *p) {
    orc_shape s; orc_get_shape(p, &s);
    return (size_t)s.n_left * N0 * p->t_exp * NTTP;
}
size_t orc_words_w_right(const orc_params *p) {
    orc_shape s; orc_get_shape(p, &s);
    return (size_t)s.n_right * N0 * p->t_exp_right * NTTP;
}
size_t orc_words_w(const orc_params *p) { return (size_t)N1 * N0 * p->t_conv * NTTP; }
size_t orc_words_v(const orc_params *p) { return (size_t)N1 * 2 * p->t_conv * NTTP; }
size_t orc_words_query(const orc_params *p) {
    orc_shape s; orc_get_shape(p, &s);
    return (size_t)s.n_query_cts * N0 * NTTP;
}

/* getRegevSample (client.cpp:147-163): [ -a ; a*s + e ] as an n0 x 1 NTT column */
static void regev_sample(orc_client *c, uint64_t *out) {
    uint64_t a[N], e[N], ainv[N], a_ntt[NTTP], s_ntt[NTTP], e_ntt[NTTP], prod[NTTP];
    fill_uniform(c, a, 1);
    fill_noise(c, e, 1);
    orc_invert(ainv, a, 1);
    orc_to_ntt(a_ntt, a, 1);
    orc_to_ntt(s_ntt, c->sr, 1);
    orc_to_ntt(e_ntt, e, 1);
    orc_multiply(prod, a_ntt, s_ntt, 1, 1, 1);
    orc_to_ntt(out, ainv, 1);
    orc_add(out + NTTP, prod, e_ntt, 1);
}

/* encryptSimpleRegev (client.cpp:176-192): sample + (0 ; sigma) */
static void encrypt_simple_regev(orc_client *c, const uint64_t *sigma_raw, uint64_t *out) {
    uint64_t sig_ntt[NTTP];
    regev_sample(c, out);
    orc_to_ntt(sig_ntt, sigma_raw, 1);
    orc_add(out + NTTP, out + NTTP, sig_ntt, 1);
}

/* encryptSimpleRegevMatrix (client.cpp:214-233): n0 x m, column i = sample + (0 ; mat[i]) */
static void encrypt_simple_regev_matrix(orc_client *c, const uint64_t *mat_ntt, size_t m, uint64_t *out) {
    uint64_t col[N0 * NTTP];
    for (size_t i = 0; i < m; i++) {
        regev_sample(c, col);
        memcpy(out + (0 * m + i) * NTTP, col, NTTP * 8);
        orc_add(out + (1 * m + i) * NTTP, col + NTTP, mat_ntt + i * NTTP, 1);
    }
}

/* to_ntt(get_fresh_public_key_raw(Sp, m)) (client.cpp:48-68): [ -A ; Sp*A + E ], n1 x m NTT */
static void fresh_public_key_ntt(orc_client *c, size_t m, uint64_t *out) {
    uint64_t *A = malloc(m * N * 8), *E = malloc(N0 * m * N * 8), *Ainv = malloc(m * N * 8);
    uint64_t *A_ntt = malloc(m * NTTP * 8), *E_ntt = malloc(N0 * m * NTTP * 8), *sp_ntt = malloc(N0 * NTTP * 8);
    uint64_t *Bp = malloc(N0 * m * NTTP * 8);
    fill_uniform(c, A, m);
    fill_noise(c, E, N0 * m);
    orc_to_ntt(A_ntt, A, m);
    orc_to_ntt(sp_ntt, c->sp, N0);
    orc_to_ntt(E_ntt, E, N0 * m);
    orc_multiply(Bp, sp_ntt, A_ntt, N0, 1, m);
    orc_invert(Ainv, A, m);
    orc_to_ntt(out, Ainv, m);
    orc_add(out + m * NTTP, E_ntt, Bp, N0 * m);
    free(A); free(E); free(Ainv); free(A_ntt); free(E_ntt); free(sp_ntt); free(Bp);
}

/* getPublicEncryptions (client.cpp:270-293): W_exp_i = Enc_s0( tau_i(s0) * G_exp ) */
static void public_encryptions(orc_client *c, uint32_t count, uint32_t t_dim, uint64_t *out) {
    uint64_t *G = malloc((size_t)t_dim * N * 8), *G_ntt = malloc((size_t)t_dim * NTTP * 8);
    uint64_t *mat = malloc((size_t)t_dim * NTTP * 8);
    uint64_t tau[N], tau_ntt[NTTP];
    orc_build_gadget(G, 1, t_dim);
    orc_to_ntt(G_ntt, G, t_dim);
    for (uint32_t i = 0; i < count; i++) {
        uint64_t t = (N >> i) + 1;
        orc_automorph(tau, c->sr, 1, t);
        orc_to_ntt(tau_ntt, tau, 1);
        orc_multiply(mat, tau_ntt, G_ntt, 1, 1, t_dim);
        encrypt_simple_regev_matrix(c, mat, t_dim, out + (size_t)i * N0 * t_dim * NTTP);
    }
    free(G); free(G_ntt); free(mat);
}

void orc_client_pub_params(orc_client *c, uint64_t *w_left, uint64_t *w_right, uint64_t *w, uint64_t *v) {
    const orc_params *p = &c->p;
    uint32_t tc = p->t_conv;
    /* :2091-2092 -- right first, then left, as the reference draws them */
    if (c->s.n_right) public_encryptions(c, c->s.n_right, p->t_exp_right, w_right);
    if (c->s.n_left) public_encryptions(c, c->s.n_left, p->t_exp, w_left);
    uint64_t s0_ntt[NTTP];
    orc_to_ntt(s0_ntt, c->sr, 1);
    { /* W = P + pad(s0 * G_scale) (:2205-2219) */
        size_t m = (size_t)N0 * tc;
        uint64_t *G = malloc(N0 * m * N * 8), *G_ntt = malloc(N0 * m * NTTP * 8), *s0G = malloc(N0 * m * NTTP * 8);
        orc_build_gadget(G, N0, m);
        orc_to_ntt(G_ntt, G, N0 * m);
        orc_mul_by_const(s0G, s0_ntt, G_ntt, N0 * m);
        fresh_public_key_ntt(c, m, w);
        orc_add(w + m * NTTP, w + m * NTTP, s0G, N0 * m);
        free(G); free(G_ntt); free(s0G);
    }
    { /* V = P + pad(Sp * [s0*gv | gv]) (:2279-2296) */
        size_t m = (size_t)2 * tc;
        uint64_t *gv = malloc(tc * N * 8), *tog = malloc(m * NTTP * 8), *sp_ntt = malloc(N0 * NTTP * 8);
        uint64_t *res = malloc(N0 * m * NTTP * 8);
        orc_build_gadget(gv, 1, tc);
        orc_to_ntt(tog + (size_t)tc * NTTP, gv, tc);
        orc_mul_by_const(tog, s0_ntt, tog + (size_t)tc * NTTP, tc);
        orc_to_ntt(sp_ntt, c->sp, N0);
        fresh_public_key_ntt(c, m, v);
        orc_multiply(res, sp_ntt, tog, N0, 1, m);
        orc_add(v + m * NTTP, v + m * NTTP, res, N0 * m);
        free(gv); free(tog); free(sp_ntt); free(res);
    }
}

static uint64_t inv_mod_q(uint64_t a) { /* util.cpp:276-288 (a odd power of two -> use Fermat-free egcd) */
    __int128 t = 0, nt = 1, r = Q, nr = a % Q;
    while (nr != 0) {
        __int128 q = r / nr, tmp = t - q * nt;
        t = nt; nt = tmp;
        tmp = r - q * nr; r = nr; nr = tmp;
    }
    if (t < 0) t += Q;
    return (uint64_t)t;
}

void orc_client_query(orc_client *c, uint64_t idx_target, uint64_t *query) {
    const orc_params *p = &c->p;
    const orc_shape *s = &c->s;
    uint64_t idx_dim0 = idx_target / s->num_per, idx_further = idx_target % s->num_per;
    uint64_t scale_k = Q / p->p_db; /* values.h:93 */
    uint32_t bits = orc_get_bits_per(s->ell);
    uint64_t sigma[N];
    const size_t CT = (size_t)N0 * NTTP;
    if (p->direct_upload) { /* :2177-2188 then :2298-2310 */
        for (uint32_t i = 0; i < s->dim0; i++) {
            memset(sigma, 0, sizeof(sigma));
            if (i == idx_dim0) sigma[0] = scale_k % Q;
            encrypt_simple_regev(c, sigma, query + (size_t)i * CT);
        }
        for (uint32_t i = 0; i < p->nu2; i++) {
            uint64_t bit = (idx_further >> i) & 1;
            for (uint32_t j = 0; j < s->ell; j++) {
                memset(sigma, 0, sizeof(sigma));
                sigma[0] = bit ? (1ull << (j * bits)) : 0;
                encrypt_simple_regev(c, sigma, query + (size_t)(s->dim0 + i * s->ell + j) * CT);
            }
        }
        return;
    }
    memset(sigma, 0, sizeof(sigma));
    if (s->stopround != 0) { /* :2104-2116 + :2141-2147 */
        sigma[2 * idx_dim0] = scale_k % Q;
        for (uint32_t i = 0; i < p->nu2; i++) {
            uint64_t bit = (idx_further >> i) & 1;
            for (uint32_t j = 0; j < s->ell; j++) sigma[2 * (i * s->ell + j) + 1] = ((1ull << (bits * j)) * bit) % Q;
        }
        uint64_t inv_first = inv_mod_q(1ull << s->g), inv_rest = inv_mod_q(1ull << (s->stopround + 1));
        for (uint32_t i = 0; i < N / 2; i++) {
            sigma[2 * i] = (uint64_t)((u128)sigma[2 * i] * inv_first % Q);
            sigma[2 * i + 1] = (uint64_t)((u128)sigma[2 * i + 1] * inv_rest % Q);
        }
    } else { /* :2117-2140 + :2148-2152 with qe_rest == 0 */
        sigma[idx_dim0] = scale_k % Q;
        uint32_t ctr = 0;
        for (uint32_t i = 0; i < p->nu2; i++) {
            uint64_t bit = (idx_further >> i) & 1;
            for (uint32_t j = 0; j < s->ell; j++) sigma[s->dim0 + ctr++] = ((1ull << (bits * j)) * bit) % Q;
        }
        uint64_t inv = inv_mod_q(1ull << s->g);
        for (uint32_t i = 0; i < N; i++) sigma[i] = (uint64_t)((u128)sigma[i] * inv % Q);
    }
    encrypt_simple_regev(c, sigma, query);
}

/* negacyclic product mod q' (the reference uses HEXL's NTT mod q' here, util.cpp:213-274; the product
 * is a mathematical function of its inputs, so schoolbook gives the same polynomial) */
static void negacyclic_mul_mod(uint64_t *res, const uint64_t *a, const uint64_t *b, uint64_t qp) {
    memset(res, 0, N * sizeof(uint64_t));
    for (uint32_t i = 0; i < N; i++) {
        if (a[i] == 0) continue;
        for (uint32_t j = 0; j < N; j++) {
            uint64_t pr = (uint64_t)((u128)a[i] * b[j] % qp);
            uint32_t k = i + j;
            if (k < N) res[k] = (res[k] + pr) % qp;
            else res[k - N] = (res[k - N] + qp - pr) % qp;
        }
    }
}

/* check_final client half (src/spiral.cpp:1451-1491) */
void orc_client_decode(orc_client *c, const uint64_t *resp, uint64_t *pt_out) {
    uint64_t qp = c->s.qprime, p_db = c->p.p_db, q1 = 4 * p_db;
    uint64_t spq[N0 * N], prod[N];
    for (size_t i = 0; i < (size_t)N0 * N; i++) { /* to_ntt_qprime's centring, util.cpp:218-223 */
        int64_t a = (int64_t)c->sp[i];
        if (a >= (int64_t)(Q / 2)) a -= (int64_t)Q;
        spq[i] = (uint64_t)(((__int128)a + (__int128)((Q / qp) * qp) + (__int128)(2 * qp)) % (__int128)qp);
    }
    for (uint32_t r = 0; r < N0; r++)
        for (uint32_t col = 0; col < N2; col++) {
            negacyclic_mul_mod(prod, spq + (size_t)r * N, resp + (size_t)col * N, qp); /* Sp[r][0] * row0[col] */
            for (uint32_t z = 0; z < N; z++) {
                int64_t vf = (int64_t)prod[z];
                if (vf >= (int64_t)(qp / 2)) vf -= (int64_t)qp;
                int64_t vr = (int64_t)resp[((size_t)(1 + r) * N2 + col) * N + z];
                if (vr >= (int64_t)(q1 / 2)) vr -= (int64_t)q1;
                uint64_t denom = qp * (q1 / p_db);
                int64_t rr = vf * (int64_t)q1 + vr * (int64_t)qp;
                int64_t sign = rr >= 0 ? 1 : -1;
                __int128 res = ((__int128)rr + sign * (int64_t)(denom / 2)) / (__int128)denom;
                res = (res + (__int128)((denom / p_db) * p_db) + (__int128)(2 * p_db)) % (__int128)p_db;
                pt_out[((size_t)r * N2 + col) * N + z] = (uint64_t)res;
            }
        }
}
