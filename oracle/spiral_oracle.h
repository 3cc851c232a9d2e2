/*
 * spiral_oracle.h -- CPU restatement of the Spiral server-answer path (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle for the HIP kernels in spiral_amd/csrc.  It is plain scalar C that
 * restates the algorithm of the reference (menonsamir/spiral) function by function; every function
 * cites the reference file:line it follows.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The product library (libspiral_gpu.so) never links or calls it.
 *
 * PINNING STATUS: "parity partially pinned".  The reference cannot be built in this image under the
 * no-stand-in rule (every TU includes hexl/ntt/ntt.hpp via include/core.h:13, HEXL is absent), and
 * the reference ships no golden vectors.  The oracle is pinned against what the reference does hold:
 *   (1) the 8x2048 twiddle table `tables[]` (src/constants.cpp:16) -- tests/golden/ntt_tables.json
 *       holds its SHA-256 per row + samples; orc_get_tables() must reproduce it exactly;
 *   (2) do_MatPol_test (src/spiral.cpp:1181): from_ntt(to_ntt(A)) == A;
 *   (3) the end-to-end "Is correct?" check (src/spiral.cpp:1412-1494): a query generated and decoded
 *       by the oracle's restated client must decrypt to the database item.
 *
 * (Round 2 added data pins for the parameter-selection side only -- tests/golden/scheme_model.json: the published
 * parameter table and 600 outputs of the reference's generate_all_schemes.py -- which pin spiral_amd/scheme.py, not this
 * file.)  The `native` Makefile target builds the same sources with -march=native -fopenmp for the full-size parity tests
 * and bench.py's all-cores baseline; orc_set_threads() enables the `omp parallel for`s, results are identical.
 *
 * Layouts are the reference's: NTT-form polynomial = [2 limbs][2048] u64 residues (limb 0 mod p,
 * limb 1 mod b); raw polynomial = [2048] u64 in [0,Q]; matrices row-major (include/poly.h:24-64).
 */
#ifndef SPIRAL_ORACLE_H
#define SPIRAL_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_N 2048u
#define ORC_P 268369921u /* include/values.h:13 */
#define ORC_B 249561089u /* include/values.h:21 */
#define ORC_Q 66974689739603969ull

/* Scheme parameters: the reference's compile-time -D values (include/values.h:78-93) + argv. */
typedef struct orc_params {
    uint32_t nu1;          /* num_expansions (argv[1]) */
    uint32_t nu2;          /* further_dims   (argv[2]) */
    uint32_t t_gsw;        /* TGSW */
    uint32_t t_conv;       /* TCONV */
    uint32_t t_exp;        /* TEXP */
    uint32_t t_exp_right;  /* TEXPRIGHT */
    uint32_t qprime_bits;  /* QPBITS */
    uint32_t direct_upload;/* 1: QNUMFIRST=2^nu1, QNUMREST=t_gsw*nu2 ; 0: QNUMFIRST=1, QNUMREST=0 */
    uint64_t p_db;         /* PVALUE */
} orc_params;

/* derived sizes (src/spiral.cpp:2046-2085) */
typedef struct orc_shape {
    uint32_t dim0, num_per, ell, m2, g, stopround;
    uint32_t n_left;      /* number of W_exp_left matrices  (g, or 0 when direct upload) */
    uint32_t n_right;     /* number of W_exp_right matrices (stopround+1, or g, or 0)   */
    uint32_t n_query_cts; /* ciphertexts in the query: 1, or dim0 + nu2*ell             */
    uint32_t n_bits;      /* dim0 + ell*nu2 */
    uint64_t qprime;
} orc_shape;

int orc_get_shape(const orc_params *p, orc_shape *s);

/* ---- L1: NTT core (src/core.cpp:247-514, tables src/constants.cpp:16) ---- */
void orc_get_tables(uint64_t *out /* [8][2048] in the reference's row order */);
void orc_ntt_forward(uint64_t *op /* [2][2048] in place */);
void orc_ntt_inverse(uint64_t *op /* [2][2048] in place */);

/* ---- L2: polynomial algebra (src/poly.cpp) ---- */
void orc_to_ntt(uint64_t *out, const uint64_t *in, size_t npolys);            /* poly.cpp:311 */
void orc_to_ntt_no_reduce(uint64_t *out, const uint64_t *in, size_t npolys);  /* poly.cpp:291 */
void orc_from_ntt(uint64_t *out, const uint64_t *in, size_t npolys);          /* poly.cpp:357 */
void orc_multiply(uint64_t *out, const uint64_t *a, const uint64_t *b, size_t rs, size_t ms,
                  size_t cs);                                                 /* poly.cpp:34 */
void orc_add(uint64_t *out, const uint64_t *a, const uint64_t *b, size_t npolys); /* poly.cpp:138 */
void orc_mul_by_const(uint64_t *out, const uint64_t *single, const uint64_t *a,
                      size_t npolys);                                         /* poly.cpp:190 */
void orc_automorph(uint64_t *out, const uint64_t *in, size_t npolys, uint64_t t); /* poly.cpp:240 */
void orc_invert(uint64_t *out, const uint64_t *in, size_t npolys);            /* poly.cpp:269 */
uint64_t orc_rescale(uint64_t a, uint64_t inp_mod, uint64_t out_mod);         /* poly.cpp:578 */
/* bit-granular little-endian field access, core.cpp:20-52 (num_bits < 64; buffers carry one spare word at the end) */
uint64_t orc_read_arbitrary_bits(const uint64_t *p, size_t bit_offs, size_t num_bits);
void orc_write_arbitrary_bits(uint64_t *p, uint64_t val, size_t bit_offs, size_t num_bits);
/* wire form of a switched response [(out_n+1)][out_n][N] (base Spiral: out_n = 2): the walk of modswitch (spiral.cpp:40-76:
 * rows, columns, coefficients, one contiguous bit stream written with write_arbitrary_bits) at the two widths the summary's
 * "Response size" assumes (spiral.cpp:231-233): row 0 -- the q' row -- at qprime_bits per coefficient, the other rows at
 * ceil(log2(4 p_db)).  orc_response_wire_bytes = that size; the buffer handed to to_wire / from_wire holds 8 more bytes. */
size_t orc_response_wire_bytes(const orc_params *p, uint32_t out_n);
void orc_response_to_wire(const orc_params *p, uint32_t out_n, const uint64_t *resp, uint64_t *wire);
void orc_response_from_wire(const orc_params *p, uint32_t out_n, const uint64_t *wire, uint64_t *resp);
uint64_t orc_crt_compose(uint64_t x, uint64_t y);                             /* poly.cpp:344 */

/* ---- L3: gadget (src/util.cpp:89-144, include/util.h:34) ---- */
uint32_t orc_get_bits_per(uint32_t dim);
void orc_build_gadget(uint64_t *G /* raw [rows][cols][N], zeroed by callee */, size_t rows, size_t cols);
void orc_gadget_invert(uint64_t *out /* raw [mx][cols][N] */, const uint64_t *in /* raw [rdim][cols][N] */,
                       size_t mx, size_t rdim, size_t cols);

/* ---- L5: server hot path (src/spiral.cpp) ---- */
void orc_split_and_crt(uint64_t *out, const uint64_t *in, size_t num_per, uint32_t t_gsw); /* :270 */
void orc_reorient_C(uint64_t *out, const uint64_t *in, size_t num_per, uint32_t m2);       /* :345 */
void orc_reorient_Q(uint64_t *out, const uint64_t *in, uint32_t m2);                       /* :388 */
void orc_reorient_ciphertexts(uint64_t *out, const uint64_t *in, size_t dim0);             /* :410 */
void orc_multiply_query_by_database(uint64_t *out, const uint64_t *reoriented_cts,
                                    const uint64_t *db, size_t dim0, size_t num_per);      /* :628 */
void orc_cpu_mul_query_by_ct(uint64_t *c_next, const uint64_t *q, const uint64_t *c, size_t num_per,
                             uint32_t m2);                                                 /* :464 */
/* cts: raw [2*num_per][3][2][N] in, first num_per overwritten (spiral.cpp:1349).  q/q_neg are the
 * reoriented (z,r,m) packed matrices of ONE dimension. */
void orc_fold_one_further_dimension(uint64_t *cts, size_t num_per, const uint64_t *q_reoriented,
                                    const uint64_t *q_neg_reoriented, uint32_t t_gsw);
/* cv: [2^g] ciphertexts, each n0 x 1 NTT ([2][2][N]); cv[0] is the query, the rest zero. */
void orc_expand_improved(uint64_t *cv, uint32_t g, uint32_t t_exp, const uint64_t *w_left,
                         uint32_t t_exp_right, const uint64_t *w_right, uint32_t n_right,
                         uint32_t max_bits_right, uint32_t stopround);                     /* :1664 */
void orc_scal_to_mat(uint64_t *out /* 3x2 NTT */, const uint64_t *cv /* 2x1 NTT */,
                     const uint64_t *w /* 3 x 2*t_conv NTT */, uint32_t t_conv);           /* :1850 */
void orc_regev_to_gsw(uint64_t *out /* 3 x 3*ell NTT */, const uint64_t *cv_v /* ell cts */,
                      const uint64_t *w, const uint64_t *v, uint32_t t_conv, uint32_t ell); /* :1985 */

/* ---- staged pipeline (server halves of runConversionImproved / process_crtd_query /
 *      process_query_fast, src/spiral.cpp:2040-2406, 1584-1629) ---- */
/* query -> n_bits Regev cts (2x1 NTT each), in the order scalToMat/regevToGSW consume them */
int orc_stage_expand(const orc_params *p, const uint64_t *query, const uint64_t *w_left,
                     const uint64_t *w_right, uint64_t *cv_out);
/* cv -> dim0 matrix-Regev cts (3x2 NTT) = expansionLocals.cts, and nu2 GSW cts (3 x m2 NTT) in the
 * reference's reversed order (spiral.cpp:2324) */
int orc_stage_convert(const orc_params *p, const uint64_t *cv, const uint64_t *w, const uint64_t *v,
                      uint64_t *cts_out, uint64_t *gsw_out);
/* first dimension: reorient + sweep + INTT + CRT lift -> raw [num_per][3][2][N] */
int orc_stage_first_dim(const orc_params *p, const uint64_t *cts, const uint64_t *db, uint64_t *raw_out);
/* folding: raw cts [num_per][3][2][N] (clobbered) + GSW (NTT, reversed order) -> raw [3][2][N] */
int orc_stage_fold(const orc_params *p, uint64_t *raw_cts, const uint64_t *gsw, uint64_t *final_out);
/* response modulus switch (spiral.cpp:1441-1447): row 0 -> q', rows 1.. -> 4*p_db */
int orc_stage_rescale(const orc_params *p, const uint64_t *final_ct, uint64_t *resp_out);
/* everything: query + public params + DB -> final raw ct */
int orc_answer(const orc_params *p, const uint64_t *query, const uint64_t *w_left,
               const uint64_t *w_right, const uint64_t *w, const uint64_t *v, const uint64_t *db,
               uint64_t *final_out);

/* ---- DB producer (src/spiral.cpp:1083-1171): explicit DB in the reference's packed layout ---- */
/* coefficient of plaintext `item`, position k in [0, 4*N): splitmix64(seed ^ (item*4N + k)) % p_db */
uint64_t orc_db_coeff(uint64_t seed, uint64_t item, uint64_t k, uint64_t p_db);
void orc_gen_db(const orc_params *p, uint64_t seed, uint64_t *db /* dim0*num_per*4*N u64 */);
/* one plaintext (raw n0 x n2, coefficients in [0, p_db)) -> its NTT-form encoding pts_encd (:1116-1128) */
void orc_encode_item(const orc_params *p, const uint64_t *pt, uint64_t *enc /* [2][2][2][N] */);
/* the sweep on nz chosen NTT slots only: cts / db hold those slots' slabs, out is [num_per][n1][n2][2][nz] */
void orc_multiply_query_by_database_slots(uint64_t *out, const uint64_t *reoriented_cts, const uint64_t *db, size_t dim0,
                                          size_t num_per, uint32_t nz);
/* the scalar cell only, and which vector form this build's orc_multiply_query_by_database uses ("avx512" | "avx2" | "scalar") */
void orc_multiply_query_by_database_scalar(uint64_t *out, const uint64_t *cts, const uint64_t *db, size_t dim0, size_t num_per);
const char *orc_sweep_isa(void);
/* the transforms in the reference's vector form (USE_AVX2: forward butterflies for t >= 4 and both closing corrections, core.cpp:292-349,
 * 479-506; the inverse butterflies are scalar there too): orc_set_ntt_simd(1) routes orc_ntt_forward / orc_ntt_inverse -- and with them
 * every stage built on them -- through it (returns what is in force: 0 on a build without AVX2); orc_ntt_isa() names the ISA compiled in;
 * the *_scalar entries always run the scalar restatement (tests prove the two equal) */
const char *orc_ntt_isa(void);
int orc_set_ntt_simd(int on);
void orc_ntt_forward_scalar(uint64_t *op);
void orc_ntt_inverse_scalar(uint64_t *op);
/* threads for the `native` (-fopenmp) build used by bench.py's all-cores CPU baseline; a no-op returning 1 otherwise */
int orc_set_threads(int n);
void orc_db_item(const orc_params *p, uint64_t seed, uint64_t item, uint64_t *pt /* raw [2][2][N] */);
/* timing-only DB: pseudo-random residues directly in NTT form (valid input, no meaning) */
void orc_fill_db_random(uint64_t seed, uint64_t *db, size_t nwords);

/* ---- client restatement (src/client.cpp, client parts of spiral.cpp:2040-2331,1412-1494) ---- */
typedef struct orc_client orc_client;
orc_client *orc_client_new(const orc_params *p, uint64_t seed, int nonoise);
void orc_client_free(orc_client *c);
/* sizes in u64 words */
size_t orc_words_w_left(const orc_params *p);
size_t orc_words_w_right(const orc_params *p);
size_t orc_words_w(const orc_params *p);
size_t orc_words_v(const orc_params *p);
size_t orc_words_query(const orc_params *p);
void orc_client_pub_params(orc_client *c, uint64_t *w_left, uint64_t *w_right, uint64_t *w, uint64_t *v);
void orc_client_query(orc_client *c, uint64_t idx_target, uint64_t *query);
/* resp: rescaled [3][2][N]; out: plaintext raw [2][2][N] in [0,p_db) */
void orc_client_decode(orc_client *c, const uint64_t *resp, uint64_t *pt_out);

#ifdef __cplusplus
}
#endif

/* =====================================================================================================
 * SpiralPack / SpiralStreamPack path (reference src/testing.cpp, `--high-rate`): base_dim x 1 scalar Regev
 * ciphertexts, 1 x 1 plaintexts, out_n^2 independent database "trials", key-switch packing into one
 * (out_n+1) x out_n ciphertext.  Same test-infrastructure-only status as the rest of this header.
 * ===================================================================================================== */
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_pack_shape {
    uint32_t dim0, num_per, ell, g, stopround;
    uint32_t n_left, n_right;   /* expansion key-switching matrices (0 with direct upload)               */
    uint32_t n_query_cts;       /* 1, or dim0 + nu2 * 2*ell (first-dimension cts + uploaded GSW columns)  */
    uint32_t trials;            /* out_n^2 */
    uint64_t qprime;
} orc_pack_shape;
int orc_pack_get_shape(const orc_params *p, uint32_t out_n, orc_pack_shape *s);

/* regevToSimpleGsw, src/testing.cpp:108-139: cv = expanded cts (2^g, tree order); inputs at 2*(i*ell+j)+1 */
void orc_regev_to_simple_gsw(uint64_t *gsw /* [nu2][2][2*ell] NTT */, const uint64_t *cv, const uint64_t *v /* 2 x 2*t_conv */,
                             uint32_t t_conv, uint32_t ell, uint32_t nu2);
/* v_folding_neg = gadget + to_ntt(invert(from_ntt(v_folding))), src/testing.cpp:1027-1032 */
void orc_pack_fold_neg(uint64_t *neg, const uint64_t *gsw, uint32_t ell, uint32_t nu2);
/* reorientCiphertextsDim1, src/testing.cpp:342-362: cts at index j*idx_factor -> packed (z, j, m=0, r) */
void orc_reorient_dim1(uint64_t *out, const uint64_t *cts, size_t dim0, size_t idx_factor);
/* fastMultiplyQueryByDatabaseDim1, src/testing.cpp:364-593 (scalar semantics :527-592):
 * db word at z*(num_per*dim0) + ii*dim0 + j; out [num_per][2][1] NTT */
void orc_sweep_dim1(uint64_t *out, const uint64_t *db, const uint64_t *reoriented, size_t dim0, size_t num_per);
/* foldCiphertextsDim1, src/testing.cpp:596-624: raw cts [num_per][2][N] folded in place, result in cts[0] */
void orc_fold_dim1(uint64_t *cts, size_t num_per, const uint64_t *folding, const uint64_t *folding_neg, uint32_t ell, uint32_t nu2);
/* pack, src/testing.cpp:198-241: result (out_n+1) x out_n NTT from out_n^2 raw cts and out_n matrices W */
void orc_pack(uint64_t *result, uint32_t out_n, uint32_t t_conv, const uint64_t *v_ct, const uint64_t *v_w);
/* whole server path of testHighRate (src/testing.cpp:1009-1081).  db: trials databases back to back, each
 * dim0*num_per*N words in convertDb's layout (:316-340).  resp: rescaled (out_n+1) x out_n raw.  final_ntt (may be
 * NULL): the packed ciphertext before the modulus switch. */
int orc_pack_answer(const orc_params *p, uint32_t out_n, const uint64_t *query, const uint64_t *w_left, const uint64_t *w_right,
                    const uint64_t *v, const uint64_t *v_w, const uint64_t *db, uint64_t *resp, uint64_t *final_ntt);

/* seeded database: coefficient z of item i of trial t */
uint64_t orc_pack_db_coeff(uint64_t seed, uint64_t trial, uint64_t item, uint64_t z, uint64_t total_n, uint64_t p_db);
void orc_pack_gen_db(const orc_params *p, uint32_t out_n, uint64_t seed, uint64_t *db);
void orc_pack_gen_db_trial(const orc_params *p, uint32_t out_n, uint64_t seed, uint32_t trial, uint64_t *db_trial);
void orc_pack_db_item(const orc_params *p, uint32_t out_n, uint64_t seed, uint64_t item, uint64_t *pt /* raw [out_n][out_n][N] */);

typedef struct orc_pack_client orc_pack_client;
orc_pack_client *orc_pack_client_new(const orc_params *p, uint32_t out_n, uint64_t seed, int nonoise);
void orc_pack_client_free(orc_pack_client *c);
size_t orc_pack_words_v(const orc_params *p);                       /* 2 x 2*t_conv NTT          */
size_t orc_pack_words_vw(const orc_params *p, uint32_t out_n);      /* out_n x (out_n+1) x t_conv */
size_t orc_pack_words_query(const orc_params *p, uint32_t out_n);
void orc_pack_client_pub_params(orc_pack_client *c, uint64_t *w_left, uint64_t *w_right, uint64_t *v, uint64_t *v_w);
void orc_pack_client_query(orc_pack_client *c, uint64_t idx_target, uint64_t *query);
void orc_pack_client_decode(orc_pack_client *c, const uint64_t *resp, uint64_t *pt_out /* raw [out_n][out_n][N] */);

#ifdef __cplusplus
}
#endif
#endif
