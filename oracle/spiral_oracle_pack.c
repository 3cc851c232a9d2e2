/*
 * spiral_oracle_pack.c -- CPU restatement of the SpiralPack / SpiralStreamPack server path and of the client
 * half needed to check it (reference src/testing.cpp, `--high-rate`).  TEST INFRASTRUCTURE ONLY, same status
 * and pinning as spiral_oracle.c: the end-to-end "Is correct? :" check of src/testing.cpp:1136 is restated in
 * tests/test_oracle_pack.py.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "spiral_oracle.h"
#ifdef _OPENMP
#include <omp.h>
#endif

#define N ORC_N
#define Q ORC_Q
#define NTTP (2 * N)
#define BD 2 /* base_dim, include/values.h:72 */

typedef unsigned __int128 u128;

static const uint64_t QPRIME_MODS[37] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 12289, 12289, 61441, 65537, 65537, 520193, 786433, 786433,
    3604481, 7340033, 16515073, 33292289, 67043329, 132120577, 268369921, 469762049, 1073479681, 2013265921, 4293918721ull, 8588886017ull,
    17175674881ull, 34359214081ull, 68718428161ull};

static uint32_t ceil_log2(uint64_t x) {
    uint32_t r = 0;
    while ((1ull << r) < x) r++;
    return r;
}

int orc_pack_get_shape(const orc_params *p, uint32_t out_n, orc_pack_shape *s) { /* src/testing.cpp:777-801 */
    if (!p || !s || out_n < 1 || out_n > 16 || p->nu1 > 16 || p->nu2 > 16 || p->nu2 < 1 || p->t_gsw < 2) return -1;
    if (p->qprime_bits >= 37 || QPRIME_MODS[p->qprime_bits] == 0) return -1;
    s->dim0 = 1u << p->nu1;
    s->num_per = 1u << p->nu2;
    s->ell = p->t_gsw;
    s->trials = out_n * out_n;
    s->qprime = QPRIME_MODS[p->qprime_bits];
    if (p->direct_upload) {
        s->g = s->stopround = s->n_left = s->n_right = 0;
        s->n_query_cts = s->dim0 + p->nu2 * 2 * s->ell;
    } else {
        s->g = ceil_log2((uint64_t)s->ell * p->nu2 + s->dim0);
        s->stopround = ceil_log2((uint64_t)s->ell * p->nu2);
        s->n_left = s->g;
        s->n_right = s->stopround + 1;
        s->n_query_cts = 1;
        if (s->g > 11 || s->stopround == 0 || s->stopround >= s->g) return -1;
    }
    return 0;
}

/* ---- server ------------------------------------------------------------------------------------------------ */

void orc_regev_to_simple_gsw(uint64_t *gsw, const uint64_t *cv, const uint64_t *v, uint32_t t_conv, uint32_t ell, uint32_t nu2) {
    const size_t CT = (size_t)BD * NTTP;
    uint32_t cols = BD * ell, mc = BD * t_conv;
    uint64_t *raw = malloc(BD * N * 8), *gi = malloc((size_t)mc * N * 8), *gi_ntt = malloc((size_t)mc * NTTP * 8), *tmp = malloc(CT * 8);
    for (uint32_t i = 0; i < nu2; i++) {
        uint64_t *ct = gsw + (size_t)i * BD * cols * NTTP;
        for (uint32_t j = 0; j < ell; j++) {
            const uint64_t *c_inp = cv + (size_t)(BD * (i * ell + j) + 1) * CT; /* idx_factor = base_dim, offset 1 (:1024) */
            for (uint32_t r = 0; r < BD; r++) memcpy(ct + ((size_t)r * cols + BD * j + 1) * NTTP, c_inp + (size_t)r * NTTP, NTTP * 8);
            orc_from_ntt(raw, c_inp, BD);
            orc_gadget_invert(gi, raw, mc, BD, 1);
            orc_to_ntt(gi_ntt, gi, mc);
            orc_multiply(tmp, v, gi_ntt, BD, mc, 1);
            for (uint32_t r = 0; r < BD; r++) memcpy(ct + ((size_t)r * cols + BD * j) * NTTP, tmp + (size_t)r * NTTP, NTTP * 8);
        }
    }
    free(raw); free(gi); free(gi_ntt); free(tmp);
}

void orc_pack_fold_neg(uint64_t *neg, const uint64_t *gsw, uint32_t ell, uint32_t nu2) {
    size_t polys = (size_t)BD * BD * ell;
    uint64_t *g = malloc(polys * N * 8), *g_ntt = malloc(polys * NTTP * 8), *raw = malloc(polys * N * 8), *inv = malloc(polys * N * 8);
    uint64_t *inv_ntt = malloc(polys * NTTP * 8);
    orc_build_gadget(g, BD, BD * ell);
    orc_to_ntt(g_ntt, g, polys);
    for (uint32_t i = 0; i < nu2; i++) {
        orc_from_ntt(raw, gsw + (size_t)i * polys * NTTP, polys);
        orc_invert(inv, raw, polys);
        orc_to_ntt(inv_ntt, inv, polys);
        orc_add(neg + (size_t)i * polys * NTTP, g_ntt, inv_ntt, polys);
    }
    free(g); free(g_ntt); free(raw); free(inv); free(inv_ntt);
}

/* bench.py / full-size tests only: orc_set_threads(n) of the `native` OpenMP build (spiral_oracle.c); 1 otherwise */
int orc_get_threads(void);
#define g_threads orc_get_threads()

void orc_reorient_dim1(uint64_t *out, const uint64_t *cts, size_t dim0, size_t idx_factor) {
#pragma omp parallel for if (g_threads > 1)
    for (size_t j = 0; j < dim0; j++)
        for (uint32_t r = 0; r < BD; r++) {
            const uint64_t *p = cts + ((j * idx_factor) * BD + r) * NTTP;
            for (uint32_t z = 0; z < N; z++) out[(size_t)z * (dim0 * BD) + j * BD + r] = (p[z] % ORC_P) | ((p[N + z] % ORC_B) << 32);
        }
}

void orc_sweep_dim1(uint64_t *out, const uint64_t *db, const uint64_t *re, size_t dim0, size_t num_per) {
#pragma omp parallel for if (g_threads > 1)
    for (uint32_t z = 0; z < N; z++) {
        const uint64_t *a = re + (size_t)z * dim0 * BD, *bz = db + (size_t)z * num_per * dim0;
        for (size_t i = 0; i < num_per; i++) {
            u128 s0[2] = {0, 0}, s1[2] = {0, 0};
            for (size_t j = 0; j < dim0; j++) {
                uint64_t bw = bz[i * dim0 + j], blo = (uint32_t)bw, bhi = bw >> 32;
                for (uint32_t r = 0; r < BD; r++) {
                    uint64_t aw = a[j * BD + r];
                    s0[r] += (uint64_t)(uint32_t)aw * blo;
                    s1[r] += (aw >> 32) * bhi;
                }
            }
            for (uint32_t r = 0; r < BD; r++) {
                out[((size_t)i * BD + r) * NTTP + z] = (uint64_t)(s0[r] % ORC_P);
                out[((size_t)i * BD + r) * NTTP + N + z] = (uint64_t)(s1[r] % ORC_B);
            }
        }
    }
}

void orc_fold_dim1(uint64_t *cts, size_t num_per, const uint64_t *folding, const uint64_t *folding_neg, uint32_t ell, uint32_t nu2) {
    const size_t CTR = (size_t)BD * N, gsw_words = (size_t)BD * BD * ell * NTTP;
    uint32_t k = BD * ell;
    const int nthr = g_threads;
    const size_t per = (size_t)k * N + (size_t)k * NTTP + 2 * BD * NTTP; /* scratch of one iteration, one set per thread */
    uint64_t *scratch = malloc(per * (size_t)nthr * 8);
    for (uint32_t cur = 0; cur < nu2; cur++) {
        num_per /= 2;
        const uint64_t *f = folding + (size_t)(nu2 - 1 - cur) * gsw_words, *fn = folding_neg + (size_t)(nu2 - 1 - cur) * gsw_words;
#pragma omp parallel for if (nthr > 1)
        for (size_t i = 0; i < num_per; i++) {
            int tid = 0;
#ifdef _OPENMP
            tid = omp_get_thread_num();
#endif
            uint64_t *gi = scratch + per * (size_t)tid, *gi_ntt = gi + (size_t)k * N, *prod = gi_ntt + (size_t)k * NTTP, *sum = prod + BD * NTTP;
            orc_gadget_invert(gi, cts + i * CTR, k, BD, 1);
            orc_to_ntt(gi_ntt, gi, k);
            orc_multiply(prod, fn, gi_ntt, BD, k, 1);
            orc_gadget_invert(gi, cts + (num_per + i) * CTR, k, BD, 1);
            orc_to_ntt(gi_ntt, gi, k);
            orc_multiply(sum, f, gi_ntt, BD, k, 1);
            orc_add(sum, sum, prod, BD);
            orc_from_ntt(cts + i * CTR, sum, BD);
        }
    }
    free(scratch);
}

void orc_pack(uint64_t *result, uint32_t out_n, uint32_t t_conv, const uint64_t *v_ct, const uint64_t *v_w) {
    uint32_t rows = out_n + 1;
    uint64_t *v_int = malloc((size_t)rows * NTTP * 8), *gi = malloc((size_t)t_conv * N * 8), *gi_ntt = malloc((size_t)t_conv * NTTP * 8);
    uint64_t *prod = malloc((size_t)rows * NTTP * 8), *ct2_ntt = malloc(NTTP * 8);
    for (uint32_t c = 0; c < out_n; c++) {
        memset(v_int, 0, (size_t)rows * NTTP * 8);
        for (uint32_t r = 0; r < out_n; r++) {
            const uint64_t *W = v_w + (size_t)r * rows * t_conv * NTTP, *ct = v_ct + (size_t)(r * out_n + c) * BD * N;
            orc_to_ntt(ct2_ntt, ct + N, 1);
            orc_gadget_invert(gi, ct, t_conv, 1, 1);
            orc_to_ntt(gi_ntt, gi, t_conv);
            orc_multiply(prod, W, gi_ntt, rows, t_conv, 1);
            orc_add(v_int + (size_t)(1 + r) * NTTP, v_int + (size_t)(1 + r) * NTTP, ct2_ntt, 1); /* add_into (:235) */
            orc_add(v_int, v_int, prod, rows);
        }
        for (uint32_t r = 0; r < rows; r++) memcpy(result + ((size_t)r * out_n + c) * NTTP, v_int + (size_t)r * NTTP, NTTP * 8);
    }
    free(v_int); free(gi); free(gi_ntt); free(prod); free(ct2_ntt);
}

int orc_pack_answer(const orc_params *p, uint32_t out_n, const uint64_t *query, const uint64_t *w_left, const uint64_t *w_right,
                    const uint64_t *v, const uint64_t *v_w, const uint64_t *db, uint64_t *resp, uint64_t *final_ntt) {
    orc_pack_shape s;
    if (orc_pack_get_shape(p, out_n, &s)) return -1;
    const size_t CT = (size_t)BD * NTTP, gsw_words = (size_t)BD * BD * s.ell * NTTP;
    uint64_t *re = malloc((size_t)N * s.dim0 * BD * 8);
    uint64_t *gsw = malloc((size_t)p->nu2 * gsw_words * 8), *neg = malloc((size_t)p->nu2 * gsw_words * 8);
    if (!p->direct_upload) { /* :1009-1025 */
        size_t n = (size_t)1 << s.g;
        uint64_t *cv = calloc(n * CT, 8);
        memcpy(cv, query, CT * 8);
        orc_expand_improved(cv, s.g, p->t_exp, w_left, p->t_exp_right, w_right, s.n_right, s.ell * p->nu2, s.stopround);
        orc_reorient_dim1(re, cv, s.dim0, 2);
        orc_regev_to_simple_gsw(gsw, cv, v, p->t_conv, s.ell, p->nu2);
        free(cv);
    } else { /* uploaded first-dimension cts and GSW columns (:966-989) */
        orc_reorient_dim1(re, query, s.dim0, 1);
        uint32_t cols = BD * s.ell;
        for (uint32_t i = 0; i < p->nu2; i++)
            for (uint32_t col = 0; col < cols; col++)
                for (uint32_t r = 0; r < BD; r++)
                    memcpy(gsw + (size_t)i * gsw_words + ((size_t)r * cols + col) * NTTP,
                           query + ((size_t)(s.dim0 + i * cols + col) * BD + r) * NTTP, NTTP * 8);
    }
    orc_pack_fold_neg(neg, gsw, s.ell, p->nu2);
    uint64_t *acc = malloc((size_t)s.num_per * CT * 8), *raw = malloc((size_t)s.num_per * BD * N * 8);
    uint64_t *v_ct = malloc((size_t)s.trials * BD * N * 8);
    const size_t db_words = (size_t)s.dim0 * s.num_per * N;
    for (uint32_t t = 0; t < s.trials; t++) { /* :1045-1062 */
        orc_sweep_dim1(acc, db + (size_t)t * db_words, re, s.dim0, s.num_per);
        orc_from_ntt(raw, acc, (size_t)s.num_per * BD);
        orc_fold_dim1(raw, s.num_per, gsw, neg, s.ell, p->nu2);
        memcpy(v_ct + (size_t)t * BD * N, raw, BD * N * 8);
    }
    uint32_t rows = out_n + 1;
    uint64_t *packed = malloc((size_t)rows * out_n * NTTP * 8), *praw = malloc((size_t)rows * out_n * N * 8);
    orc_pack(packed, out_n, p->t_conv, v_ct, v_w);
    if (final_ntt) memcpy(final_ntt, packed, (size_t)rows * out_n * NTTP * 8);
    orc_from_ntt(praw, packed, (size_t)rows * out_n);
    uint64_t q1 = 4 * p->p_db;
    for (uint32_t r = 0; r < rows; r++) /* :1074-1081 */
        for (size_t k = 0; k < (size_t)out_n * N; k++)
            resp[(size_t)r * out_n * N + k] = orc_rescale(praw[(size_t)r * out_n * N + k] % Q, Q, r == 0 ? s.qprime : q1);
    free(re); free(gsw); free(neg); free(acc); free(raw); free(v_ct); free(packed); free(praw);
    return 0;
}

/* ---- database ------------------------------------------------------------------------------------------------ */

static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

uint64_t orc_pack_db_coeff(uint64_t seed, uint64_t trial, uint64_t item, uint64_t z, uint64_t total_n, uint64_t p_db) {
    return splitmix64(seed ^ ((trial * total_n + item) * N + z)) % p_db;
}

/* src/testing.cpp:839-902 + convertDb :316-340: word at z*(num_per*dim0) + ii*dim0 + j, item i -> (ii = i % num_per, j = i / num_per) */
void orc_pack_gen_db(const orc_params *p, uint32_t out_n, uint64_t seed, uint64_t *db) {
    orc_pack_shape s;
    if (orc_pack_get_shape(p, out_n, &s)) return;
    uint64_t total = (uint64_t)s.dim0 * s.num_per;
    for (uint32_t t = 0; t < s.trials; t++)
#pragma omp parallel for if (g_threads > 1)
        for (uint64_t i = 0; i < total; i++) {
            uint64_t pt[N], enc[NTTP];
            for (uint32_t z = 0; z < N; z++) {
                int64_t v = (int64_t)orc_pack_db_coeff(seed, t, i, z, total, p->p_db);
                if (v >= (int64_t)(p->p_db / 2)) v -= (int64_t)p->p_db;
                if (v < 0) v += (int64_t)Q;
                pt[z] = (uint64_t)v;
            }
            orc_to_ntt(enc, pt, 1);
            uint64_t ii = i % s.num_per, j = i / s.num_per;
            uint64_t *d = db + (size_t)t * total * N;
            for (uint32_t z = 0; z < N; z++) d[(size_t)z * total + ii * s.dim0 + j] = enc[z] | (enc[N + z] << 32);
        }
}

/* one trial of the above (a 4 GiB slice of config 5's 64 GiB) */
void orc_pack_gen_db_trial(const orc_params *p, uint32_t out_n, uint64_t seed, uint32_t t, uint64_t *d) {
    orc_pack_shape s;
    if (orc_pack_get_shape(p, out_n, &s)) return;
    uint64_t total = (uint64_t)s.dim0 * s.num_per;
#pragma omp parallel for if (g_threads > 1)
    for (uint64_t i = 0; i < total; i++) {
        uint64_t pt[N], enc[NTTP];
        for (uint32_t z = 0; z < N; z++) {
            int64_t v = (int64_t)orc_pack_db_coeff(seed, t, i, z, total, p->p_db);
            if (v >= (int64_t)(p->p_db / 2)) v -= (int64_t)p->p_db;
            if (v < 0) v += (int64_t)Q;
            pt[z] = (uint64_t)v;
        }
        orc_to_ntt(enc, pt, 1);
        uint64_t ii = i % s.num_per, j = i / s.num_per;
        for (uint32_t z = 0; z < N; z++) d[(size_t)z * total + ii * s.dim0 + j] = enc[z] | (enc[N + z] << 32);
    }
}

void orc_pack_db_item(const orc_params *p, uint32_t out_n, uint64_t seed, uint64_t item, uint64_t *pt) {
    uint64_t total = ((uint64_t)1 << p->nu1) << p->nu2;
    for (uint32_t t = 0; t < out_n * out_n; t++)
        for (uint32_t z = 0; z < N; z++) pt[(size_t)t * N + z] = orc_pack_db_coeff(seed, t, item, z, total, p->p_db);
}

/* ---- client ---------------------------------------------------------------------------------------------------- */

struct orc_pack_client {
    orc_params p;
    orc_pack_shape s;
    uint32_t out_n;
    uint64_t rng[4];
    int nonoise;
    double cdf[129];
    uint64_t sr[N];
    uint64_t sp[16 * N]; /* out_n x 1, out_n <= 16 (the published parameter sets go up to 12) */
};

static uint64_t rng_next(orc_pack_client *c) {
    uint64_t *s = c->rng;
    uint64_t r = ((s[1] * 5) << 7 | (s[1] * 5) >> 57) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t;
    s[3] = (s[3] << 45) | (s[3] >> 19);
    return r;
}
static uint64_t sample_noise(orc_pack_client *c) {
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
static void fill_noise(orc_pack_client *c, uint64_t *raw, size_t npolys) {
    for (size_t i = 0; i < npolys * N; i++) raw[i] = sample_noise(c);
}
static void fill_uniform(orc_pack_client *c, uint64_t *raw, size_t npolys) {
    for (size_t i = 0; i < npolys * N; i++) raw[i] = rng_next(c) % Q;
}

orc_pack_client *orc_pack_client_new(const orc_params *p, uint32_t out_n, uint64_t seed, int nonoise) {
    orc_pack_client *c = calloc(1, sizeof(*c));
    c->p = *p;
    c->out_n = out_n;
    if (orc_pack_get_shape(p, out_n, &c->s)) { free(c); return NULL; }
    for (int i = 0; i < 4; i++) c->rng[i] = splitmix64(seed + 0x7654321 * (i + 1));
    c->nonoise = nonoise;
    double acc = 0;
    for (int i = -64; i <= 64; i++) {
        acc += exp(-M_PI * (double)i * i / (6.4 * 6.4));
        c->cdf[i + 64] = acc;
    }
    fill_noise(c, c->sr, 1);       /* keygen(S, Sp, sr, out_n): src/client.cpp:21-46, testing.cpp:907-910 */
    fill_noise(c, c->sp, out_n);
    return c;
}
void orc_pack_client_free(orc_pack_client *c) { free(c); }

size_t orc_pack_words_v(const orc_params *p) { return (size_t)BD * BD * p->t_conv * NTTP; }
size_t orc_pack_words_vw(const orc_params *p, uint32_t out_n) { return (size_t)out_n * (out_n + 1) * p->t_conv * NTTP; }
size_t orc_pack_words_query(const orc_params *p, uint32_t out_n) {
    orc_pack_shape s;
    orc_pack_get_shape(p, out_n, &s);
    return (size_t)s.n_query_cts * BD * NTTP;
}

static void regev_sample(orc_pack_client *c, uint64_t *out) { /* src/client.cpp:147-163 */
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
static void encrypt_simple_regev(orc_pack_client *c, const uint64_t *sigma_raw, uint64_t *out) { /* src/client.cpp:176-192 */
    uint64_t sig_ntt[NTTP];
    regev_sample(c, out);
    orc_to_ntt(sig_ntt, sigma_raw, 1);
    orc_add(out + NTTP, out + NTTP, sig_ntt, 1);
}
/* getExpansionKeySwitchingMatrices, src/testing.cpp:21-38 */
static void expansion_keys(orc_pack_client *c, uint32_t count, uint32_t t_dim, uint64_t *out) {
    uint64_t *G = malloc((size_t)t_dim * N * 8), *G_ntt = malloc((size_t)t_dim * NTTP * 8), *mat = malloc((size_t)t_dim * NTTP * 8);
    uint64_t tau[N], tau_ntt[NTTP], col[BD * NTTP];
    orc_build_gadget(G, 1, t_dim);
    orc_to_ntt(G_ntt, G, t_dim);
    for (uint32_t i = 0; i < count; i++) {
        orc_automorph(tau, c->sr, 1, (N >> i) + 1);
        orc_to_ntt(tau_ntt, tau, 1);
        orc_multiply(mat, tau_ntt, G_ntt, 1, 1, t_dim);
        uint64_t *o = out + (size_t)i * BD * t_dim * NTTP;
        for (uint32_t k = 0; k < t_dim; k++) { /* encryptSimpleRegevMatrix, src/client.cpp:214-233 */
            regev_sample(c, col);
            memcpy(o + (size_t)k * NTTP, col, NTTP * 8);
            orc_add(o + ((size_t)t_dim + k) * NTTP, col + NTTP, mat + (size_t)k * NTTP, 1);
        }
    }
    free(G); free(G_ntt); free(mat);
}

void orc_pack_client_pub_params(orc_pack_client *c, uint64_t *w_left, uint64_t *w_right, uint64_t *v, uint64_t *v_w) {
    const orc_params *p = &c->p;
    uint32_t tc = p->t_conv, out_n = c->out_n, rows = out_n + 1;
    uint64_t s0_ntt[NTTP];
    orc_to_ntt(s0_ntt, c->sr, 1);
    /* v_W[i] = encryptMatrixArbitrary(AG_i), AG_i row i = s0 * g_vec (src/testing.cpp:918-925, 141-194) */
    uint64_t *gv = malloc((size_t)tc * N * 8), *gv_ntt = malloc((size_t)tc * NTTP * 8), *s0g = malloc((size_t)tc * NTTP * 8);
    orc_build_gadget(gv, 1, tc);
    orc_to_ntt(gv_ntt, gv, tc);
    orc_mul_by_const(s0g, s0_ntt, gv_ntt, tc);
    uint64_t *A = malloc((size_t)tc * N * 8), *Ainv = malloc((size_t)tc * N * 8), *E = malloc((size_t)out_n * tc * N * 8);
    uint64_t *A_ntt = malloc((size_t)tc * NTTP * 8), *E_ntt = malloc((size_t)out_n * tc * NTTP * 8), *sp_ntt = malloc((size_t)out_n * NTTP * 8);
    uint64_t *Bp = malloc((size_t)out_n * tc * NTTP * 8);
    orc_to_ntt(sp_ntt, c->sp, out_n);
    for (uint32_t i = 0; i < out_n; i++) {
        uint64_t *W = v_w + (size_t)i * rows * tc * NTTP;
        fill_uniform(c, A, tc);
        fill_noise(c, E, (size_t)out_n * tc);
        orc_to_ntt(A_ntt, A, tc);
        orc_to_ntt(E_ntt, E, (size_t)out_n * tc);
        orc_multiply(Bp, sp_ntt, A_ntt, out_n, 1, tc);
        orc_invert(Ainv, A, tc);
        orc_to_ntt(W, Ainv, tc);
        orc_add(W + (size_t)tc * NTTP, E_ntt, Bp, (size_t)out_n * tc);
        orc_add(W + (size_t)(1 + i) * tc * NTTP, W + (size_t)(1 + i) * tc * NTTP, s0g, tc);
    }
    if (!p->direct_upload) { /* :926-949 */
        expansion_keys(c, c->s.n_left, p->t_exp, w_left);
        expansion_keys(c, c->s.n_right, p->t_exp_right, w_right);
        uint64_t s0sq[NTTP], val_raw[N], val_ntt[NTTP], prod[NTTP], sigma[N], ct[BD * NTTP];
        orc_multiply(s0sq, s0_ntt, s0_ntt, 1, 1, 1);
        uint32_t bits = orc_get_bits_per(tc), cols = BD * tc;
        for (uint32_t i = 0; i < cols; i++) {
            uint64_t sh = (uint64_t)bits * (i / 2);
            memset(val_raw, 0, sizeof(val_raw));
            val_raw[0] = sh >= 64 ? 0 : (1ull << sh); /* G_conv[0][2j] resp. G_conv[1][2j+1] */
            orc_to_ntt(val_ntt, val_raw, 1);
            orc_multiply(prod, (i % 2 == 0) ? s0sq : s0_ntt, val_ntt, 1, 1, 1);
            orc_from_ntt(sigma, prod, 1);
            encrypt_simple_regev(c, sigma, ct);
            for (uint32_t r = 0; r < BD; r++) memcpy(v + ((size_t)r * cols + i) * NTTP, ct + (size_t)r * NTTP, NTTP * 8);
        }
    }
    free(gv); free(gv_ntt); free(s0g); free(A); free(Ainv); free(E); free(A_ntt); free(E_ntt); free(sp_ntt); free(Bp);
}

static uint64_t inv_mod_q(uint64_t a) {
    __int128 t = 0, nt = 1, r = Q, nr = a % Q;
    while (nr != 0) {
        __int128 q = r / nr, tmp = t - q * nt;
        t = nt; nt = tmp;
        tmp = r - q * nr; r = nr; nr = tmp;
    }
    if (t < 0) t += Q;
    return (uint64_t)t;
}

void orc_pack_client_query(orc_pack_client *c, uint64_t idx_target, uint64_t *query) {
    const orc_params *p = &c->p;
    const orc_pack_shape *s = &c->s;
    uint64_t idx_dim0 = idx_target / s->num_per, idx_further = idx_target % s->num_per, scale_k = Q / p->p_db;
    uint32_t bits = orc_get_bits_per(s->ell);
    const size_t CT = (size_t)BD * NTTP;
    uint64_t sigma[N];
    if (p->direct_upload) { /* :966-989 */
        for (uint32_t i = 0; i < s->dim0; i++) {
            memset(sigma, 0, sizeof(sigma));
            sigma[0] = i == idx_dim0 ? scale_k : 0;
            encrypt_simple_regev(c, sigma, query + (size_t)i * CT);
        }
        uint64_t s0_ntt[NTTP], val_ntt[NTTP], prod[NTTP];
        orc_to_ntt(s0_ntt, c->sr, 1);
        for (uint32_t i = 0; i < p->nu2; i++) {
            uint64_t bit = (idx_further >> i) & 1;
            for (uint32_t j = 0; j < s->ell; j++) {
                uint64_t *base = query + (size_t)(s->dim0 + i * 2 * s->ell) * CT;
                memset(sigma, 0, sizeof(sigma));
                sigma[0] = (1ull << (bits * j)) * bit;
                encrypt_simple_regev(c, sigma, base + (size_t)(2 * j + 1) * CT);
                orc_to_ntt(val_ntt, sigma, 1);
                orc_multiply(prod, s0_ntt, val_ntt, 1, 1, 1);
                orc_from_ntt(sigma, prod, 1);
                encrypt_simple_regev(c, sigma, base + (size_t)(2 * j) * CT);
            }
        }
        return;
    }
    memset(sigma, 0, sizeof(sigma)); /* :991-1006 */
    sigma[2 * idx_dim0] = scale_k;
    for (uint32_t i = 0; i < p->nu2; i++) {
        uint64_t bit = (idx_further >> i) & 1;
        for (uint32_t j = 0; j < s->ell; j++) sigma[2 * (i * s->ell + j) + 1] = (1ull << (bits * j)) * bit;
    }
    uint64_t inv_first = inv_mod_q(1ull << s->g), inv_rest = inv_mod_q(1ull << (s->stopround + 1));
    for (uint32_t i = 0; i < N / 2; i++) {
        sigma[2 * i] = (uint64_t)((u128)sigma[2 * i] * inv_first % Q);
        sigma[2 * i + 1] = (uint64_t)((u128)sigma[2 * i + 1] * inv_rest % Q);
    }
    encrypt_simple_regev(c, sigma, query);
}

/* src/testing.cpp:1086-1122 */
void orc_pack_client_decode(orc_pack_client *c, const uint64_t *resp, uint64_t *pt_out) {
    uint32_t out_n = c->out_n;
    uint64_t qp = c->s.qprime, p_db = c->p.p_db, q1 = 4 * p_db;
    uint64_t *spq = malloc((size_t)out_n * N * 8), prod[N];
    for (size_t i = 0; i < (size_t)out_n * N; i++) {
        __int128 a = (__int128)c->sp[i];
        if (a >= (__int128)(Q / 2)) a -= Q;
        spq[i] = (uint64_t)((a + (__int128)((Q / qp) * qp) + (__int128)(2 * qp)) % (__int128)qp);
    }
    for (uint32_t r = 0; r < out_n; r++)
        for (uint32_t col = 0; col < out_n; col++) {
            const uint64_t *a = spq + (size_t)r * N, *b = resp + (size_t)col * N;
            memset(prod, 0, sizeof(prod));
            for (uint32_t i = 0; i < N; i++) {
                if (a[i] == 0) continue;
                for (uint32_t j = 0; j < N; j++) {
                    uint64_t pr = (uint64_t)((u128)a[i] * b[j] % qp);
                    uint32_t k = i + j;
                    if (k < N) prod[k] = (prod[k] + pr) % qp;
                    else prod[k - N] = (prod[k - N] + qp - pr) % qp;
                }
            }
            for (uint32_t z = 0; z < N; z++) {
                int64_t vf = (int64_t)prod[z];
                if (vf >= (int64_t)(qp / 2)) vf -= (int64_t)qp;
                int64_t vr = (int64_t)resp[((size_t)(1 + r) * out_n + col) * N + z];
                if (vr >= (int64_t)(q1 / 2)) vr -= (int64_t)q1;
                uint64_t denom = qp * (q1 / p_db);
                int64_t rr = vf * (int64_t)q1 + vr * (int64_t)qp;
                int64_t sign = rr >= 0 ? 1 : -1;
                __int128 res = ((__int128)rr + sign * (int64_t)(denom / 2)) / (__int128)denom;
                res = (res + (__int128)((denom / p_db) * p_db) + (__int128)(2 * p_db)) % (__int128)p_db;
                pt_out[((size_t)r * out_n + col) * N + z] = (uint64_t)res;
            }
        }
    free(spq);
}
