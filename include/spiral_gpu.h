/*
 * spiral_gpu.h -- C ABI of libspiral_gpu.so: the MI355X (gfx950) implementation of the Spiral
 * server-answer path.
 *
 * The reference (menonsamir/spiral) has no FFI or plugin interface: it is one C++ executable whose
 * server hot path is a set of free functions over caller-owned uint64_t buffers.
 * This header declares exactly those seams, so that a maintainer can replace the reference function
 * bodies with calls into this library (INTEGRATION.md shows the stubs).  Each entry point cites the
 * reference function it replaces.
 *
 * Conventions
 *   - plain C, no C++/torch types; every function returns 0 on success and a negative code on error
 *     (spiral_gpu_last_error() gives the message); the reference's error behaviour is assert/exit(1).
 *   - buffers use the REFERENCE layouts:
 *       NTT form : [rows][cols][2 limbs][2048] uint64_t residues (limb 0 mod p, limb 1 mod b), NTT slots
 *                  in the order produced by the reference's ntt_forward          (include/poly.h:24-64)
 *       raw form : [rows][cols][2048] uint64_t in [0, Q]
 *     any re-layout (packed u32 limb pairs, lane-major database) is internal to the library.
 *   - "host" entry points take host pointers and copy; the *_server_* stage entry points keep all
 *     state resident in HBM and are asynchronous on the server's HIP stream.
 *   - there is no CPU fallback: without a usable gfx950 device every compute entry point fails.
 */
#ifndef SPIRAL_GPU_H
#define SPIRAL_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPIRAL_GPU_ABI_VERSION 1

/* Scheme parameters = the reference's compile-time -D values (include/values.h:78-93,
 * select_params.py:337) plus argv[1], argv[2] (src/spiral.cpp:1243-1244). */
typedef struct spiral_gpu_params {
    uint32_t nu1;           /* num_expansions: first dimension is 2^nu1                          */
    uint32_t nu2;           /* further_dims:   2^nu2 plaintexts per first-dimension index        */
    uint32_t t_gsw;         /* TGSW   */
    uint32_t t_conv;        /* TCONV  */
    uint32_t t_exp;         /* TEXP   */
    uint32_t t_exp_right;   /* TEXPRIGHT */
    uint32_t qprime_bits;   /* QPBITS */
    uint32_t direct_upload; /* 0: QNUMFIRST=1, QNUMREST=0;  1: QNUMFIRST=2^nu1, QNUMREST=t_gsw*nu2 */
    uint64_t p_db;          /* PVALUE */
} spiral_gpu_params;

/* sizes derived from the parameters (src/spiral.cpp:2046-2085) */
typedef struct spiral_gpu_shape {
    uint32_t dim0, num_per, ell, m2, g, stopround;
    uint32_t n_left;      /* W_exp_left matrices  (n0 x t_exp each)       */
    uint32_t n_right;     /* W_exp_right matrices (n0 x t_exp_right each) */
    uint32_t n_query_cts; /* Regev ciphertexts (n0 x 1) in the query       */
    uint32_t n_bits;      /* dim0 + ell*nu2 expanded ciphertexts           */
    uint64_t qprime;
} spiral_gpu_shape;

typedef struct spiral_gpu_server spiral_gpu_server;

int spiral_gpu_abi_version(void);
const char *spiral_gpu_last_error(void);
int spiral_gpu_device_count(void);
int spiral_gpu_get_shape(const spiral_gpu_params *p, spiral_gpu_shape *out);

/* Process-wide options: schedule forms that compute the same function (the reference has one form of each: its own loops).  A server takes
 * the values in force when it is created; "fwd2" and "db_stage_bytes" apply to every later call.  Unknown names fail.
 *   "fold_pair"       1 (default) foldOneFurtherDimension as C[i] + Q * NTT(G^-1(C[np+i]) - G^-1(C[i])); 0 the reference's two products
 *                     (src/spiral.cpp:1349-1410).  Initial value: environment variable SPIRAL_FOLD_PAIR.
 *   "fold_chain"      1 (default) the two-product form lifts inside its digit transforms; 0 separate lift and digit launches
 *   "fold_blocks"     workgroups a two-product round aims for when it splits a polynomial's digits over workgroups (default 768)
 *   "sweep_mfma_min"  batches of at least this many queries sweep on the matrix cores (default 2; 0 = never).  Initial value: SPIRAL_SWEEP_MFMA.
 *   "one_image"       1 (default) a server that batches on the matrix cores keeps ONE image of its database and converts it in place between the
 *                     packed and the limb-plane form (spiral_gpu_server_set_db_format); 0 = a second image beside the first
 *   "fwd2"            -1 (default) the two-digits-per-workgroup transform kernel from "fwd2_min" transforms per launch; 0 never; 1 always
 *   "fwd2_min"        that threshold (default 8192 transforms per launch, all query lanes together)
 *   "db_stage_bytes"  bytes of the staging buffer of load_db / load_db_items (default 64 MiB).  Initial value: SPIRAL_DB_STAGE_BYTES.
 * These three environment variables are the only ones the library reads. */
int spiral_gpu_set_option(const char *name, int64_t value);
int spiral_gpu_get_option(const char *name, int64_t *value);

/* ------------------------------------------------------------------------------------------------
 * L1/L2 seams: NTT core and polynomial algebra (host buffers)
 * ------------------------------------------------------------------------------------------------ */
/* the 8 x 2048 twiddle rows in the order of `tables[]`, src/constants.cpp:16 (host only, no GPU) */
int spiral_gpu_get_tables(uint64_t *out);
/* void ntt_forward(uint64_t*) / ntt_inverse(uint64_t*), src/core.cpp:247,419; batched over npolys */
int spiral_gpu_ntt_forward(uint64_t *operand, size_t npolys);
int spiral_gpu_ntt_inverse(uint64_t *operand, size_t npolys);
/* to_ntt / to_ntt_no_reduce, src/poly.cpp:311,291 : raw [npolys][N] -> NTT [npolys][2][N] */
int spiral_gpu_to_ntt(uint64_t *out, const uint64_t *in, size_t npolys, int reduce);
/* from_ntt, src/poly.cpp:357 : NTT -> raw in [0, Q) */
int spiral_gpu_from_ntt(uint64_t *out, const uint64_t *in, size_t npolys);
/* measurement helper: average duration (ms) of one batched to_ntt and one batched from_ntt launch over npolys polynomials
 * resident in HBM (HIP events, default stream): the transform kernels' cost per limb-pair transform */
int spiral_gpu_time_ntt(size_t npolys, int iters, float *fwd_ms, float *inv_ms);
/* the same for the digit-transform launch the stages are built from: n_digits gadget digits of each of npolys raw polynomials
 * (gadget_invert + to_ntt_no_reduce), npolys * n_digits transforms per launch; ms per launch */
int spiral_gpu_time_ntt_digits(size_t npolys, uint32_t n_digits, int iters, float *ms);
/* multiply, src/poly.cpp:34 : out(rs x cs) = a(rs x ms) * b(ms x cs), NTT form */
int spiral_gpu_multiply(uint64_t *out, const uint64_t *a, const uint64_t *b, size_t rs, size_t ms, size_t cs);
/* add, mul_by_const, src/poly.cpp:138,190 */
int spiral_gpu_add(uint64_t *out, const uint64_t *a, const uint64_t *b, size_t npolys);
int spiral_gpu_mul_by_const(uint64_t *out, const uint64_t *single_poly, const uint64_t *a, size_t npolys);
/* automorph, invert, src/poly.cpp:240,269 (raw form; negation is Q - a) */
int spiral_gpu_automorph(uint64_t *out, const uint64_t *in, size_t npolys, uint64_t t);
int spiral_gpu_invert(uint64_t *out, const uint64_t *in, size_t npolys);
/* gadget_invert, src/util.cpp:114 : raw [rdim][cols][N] -> raw [mx][cols][N].  A digit whose shift count k*bits is >= 64
 * is defined as 0 here (and in every fused digit loader); the reference shifts a uint64_t by that count (src/util.cpp:136),
 * which is undefined behaviour -- x86 would yield the low digit again.  Unreachable for every published parameter set
 * (k*bits < 64 for all t in all_parameter_choices.txt). */
int spiral_gpu_gadget_invert(uint64_t *out, const uint64_t *in, size_t mx, size_t rdim, size_t cols);
/* getRescaled, src/poly.cpp:593 : element-wise rescale(a % Q, inp_mod, out_mod) */
int spiral_gpu_get_rescaled(uint64_t *out, const uint64_t *in, size_t n, uint64_t inp_mod, uint64_t out_mod);

/* Wire form of a switched response [(out_n+1)][out_n][N] (base Spiral: out_n = 2): one little-endian bit stream written the way
 * write_arbitrary_bits does (src/core.cpp:32-52) along modswitch's walk over rows, columns and coefficients (src/spiral.cpp:
 * 40-76), at the two widths the summary's "Response size" assumes (src/spiral.cpp:231-233): row 0 -- the q' row -- at qprime_bits
 * per coefficient, the other rows at ceil(log2(4 p_db)) bits.  20 480 bytes instead of 98 304 at config 2.
 * spiral_gpu_response_wire_bytes: its size (0 for unsupported parameters); ..._server_read_response_wire: packs the last
 * answer's response on the device and downloads it; spiral_gpu_response_from_wire: the client's half (load_modswitched_into_ct,
 * src/client.cpp:90-110), plain host code. */
size_t spiral_gpu_response_wire_bytes(const spiral_gpu_params *p, uint32_t out_n);
int spiral_gpu_response_from_wire(const spiral_gpu_params *p, uint32_t out_n, const void *wire, uint64_t *response);

/* ------------------------------------------------------------------------------------------------
 * L5 seams: the server hot-path functions (host buffers, reference layouts)
 * ------------------------------------------------------------------------------------------------ */
/* multiplyQueryByDatabase, src/spiral.cpp:628.  reorientedCts: (z, j, m, r_pad4) packed words
 * (reorientCiphertexts, :410); database: load_db's layout (:1139-1153); output: NTT form
 * [num_per][n1][n2][2][N]. */
int spiral_gpu_multiply_query_by_database(uint64_t *output, const uint64_t *reorientedCiphertexts,
                                          const uint64_t *database, size_t dim0, size_t num_per);
/* The same for n <= 8 queries against ONE pass over the database (no reference counterpart: the reference answers one query per
 * call): reorientedCts = the n queries' buffers one after the other, outputs = [n][num_per][n1][n2][2][N].  Where the geometry allows
 * (num_per >= 64, dim0 a multiple of 64, <= 2048) the pass runs on the matrix cores (csrc/sweep_mfma.hip), else as passes of two /
 * one on the vector ALU; every output equals multiply_query_by_database's for that query. */
int spiral_gpu_multiply_queries_by_database(uint64_t *outputs, const uint64_t *reorientedCiphertexts, size_t n,
                                            const uint64_t *database, size_t dim0, size_t num_per);
/* split_and_crt, src/spiral.cpp:270 : raw [num_per][n1][n2][N] -> NTT [num_per][m2][n2][2][N] */
int spiral_gpu_split_and_crt(uint64_t *out, const uint64_t *in, size_t num_per, uint32_t t_gsw);
/* foldOneFurtherDimension, src/spiral.cpp:1349.  cts: raw [2*num_per][n1][n2][N], the first num_per
 * are overwritten; query_ct / query_ct_neg: ONE dimension's reoriented (z, r, m) packed GSW matrices
 * (reorient_Q, :388). */
int spiral_gpu_fold_one_further_dimension(uint64_t *cts, size_t num_per, const uint64_t *query_ct,
                                          const uint64_t *query_ct_neg, uint32_t t_gsw);
/* expandImproved, src/spiral.cpp:1664.  cv_v: 2^g ciphertexts (n0 x 1, NTT form), element 0 is the
 * query, updated in place; W_left: g matrices n0 x t_exp; W_right: n_right matrices n0 x t_exp_right. */
int spiral_gpu_expand_improved(uint64_t *cv_v, uint32_t g, uint32_t t_exp, const uint64_t *w_left,
                               uint32_t t_exp_right, const uint64_t *w_right, uint32_t n_right,
                               uint32_t max_bits_to_gen_right, uint32_t stopround);
/* scalToMat, src/spiral.cpp:1918 : out n1 x n0 from cv n0 x 1 and W n1 x (n0*t_conv) */
int spiral_gpu_scal_to_mat(uint64_t *out, const uint64_t *cv, const uint64_t *w, uint32_t t_conv);
/* regevToGSW, src/spiral.cpp:1985 : out n1 x (n1*ell) from ell ciphertexts, W and V */
int spiral_gpu_regev_to_gsw(uint64_t *out, const uint64_t *cv_v, const uint64_t *w, const uint64_t *v,
                            uint32_t t_conv, uint32_t ell);

/* ------------------------------------------------------------------------------------------------
 * Resident server: do_test's server half (src/spiral.cpp:2337-2406, 1584-1629) with the database,
 * public parameters and all intermediates in HBM.
 * ------------------------------------------------------------------------------------------------ */
/* The server holds first-dimension indices j in [j_begin, j_end) of the database; (j_begin, j_end) = (0, 0) means the
 * whole first dimension [0, 2^nu1) (a single GPU).  Shards are summed by the caller between first_dim and lift (see
 * spiral_gpu_server_acc). */
int spiral_gpu_server_create(const spiral_gpu_params *p, int device, uint32_t j_begin, uint32_t j_end,
                             spiral_gpu_server **out);
void spiral_gpu_server_destroy(spiral_gpu_server *s);
/* use an external HIP stream (hipStream_t as void*); NULL = the server's own stream */
int spiral_gpu_server_set_stream(spiral_gpu_server *s, void *hip_stream);
/* the stream the server launches on now (hipStream_t as void*): e.g. to put the lanes of a batch on lanes[0]'s stream */
void *spiral_gpu_server_get_stream(spiral_gpu_server *s);

/* database producers: load_db, src/spiral.cpp:1028-1172 */
int spiral_gpu_server_load_db(spiral_gpu_server *s, const uint64_t *database /* full, reference layout */);
/* explicit DB generated on the device: plaintext coefficient k of item i is
 * splitmix64(seed ^ (i*4N + k)) % p_db (the rand() % p_db of :25-29 with a counter-based generator) */
int spiral_gpu_server_gen_db(spiral_gpu_server *s, uint64_t seed);
/* Raw ingest: the whole of load_db on the device -- plaintext coefficients in, device database out
 * (centred lift :1116-1127, to_ntt, layout :1139-1153), so the host ships log2(p_db) bits per coefficient instead of
 * the 8x larger NTT form.  `items` holds n_items consecutive plaintexts starting at item first_item (item i = database
 * entry (ii = i % num_per, j = i / num_per); a sharded server keeps the items of its own j-range and skips the rest);
 * one plaintext = n0*n2*2048 coefficients in [0, p_db), polynomial (m, c) at (m*n2 + c)*2048, each coeff_bits wide and
 * bit-packed little-endian like read_arbitrary_bits (src/core.cpp:20-30) -- coeff_bits = log2(p_db) is the item size
 * of select_params.py:297 (8192 B at p = 256, 15360 B at p = 2^15) -- or coeff_bits = 64: the reference's raw MatPoly
 * words.  Fails if a coefficient is >= p_db (the reference asserts).  May be called several times (streaming a database
 * larger than host memory). */
int spiral_gpu_server_load_db_items(spiral_gpu_server *s, const void *items, uint32_t coeff_bits, uint64_t first_item,
                                    uint64_t n_items);
/* read the device database back in reference layouts (tests): plaintext `item` as its n0 x n2 NTT-form MatPoly
 * (pts_encd, :1128), or slabs z_begin .. z_begin+nz-1 of load_db's layout restricted to this server's j-range:
 * word (z, ii, c, j, m) at ((((z - z_begin)*num_per + ii)*n2 + c)*(j_end - j_begin) + (j - j_begin))*n0 + m */
int spiral_gpu_server_read_db_item(spiral_gpu_server *s, uint64_t item, uint64_t *out);
int spiral_gpu_server_read_db_slots(spiral_gpu_server *s, uint32_t z_begin, uint32_t nz, uint64_t *out);
/* the same for ALL 2048 slots but only the plaintext columns ii in [ii_begin, ii_begin + n_ii): load_db's layout of a database
 * with num_per = n_ii, i.e. exactly what multiplyQueryByDatabase (src/spiral.cpp:628) needs to produce the full output
 * polynomials of those ciphertexts (2048 * n_ii * n2 * (j_end - j_begin) * n0 words; tests at sizes whose whole image
 * does not fit a host-side reference) */
int spiral_gpu_server_read_db_columns(spiral_gpu_server *s, uint32_t ii_begin, uint32_t n_ii, uint64_t *out);
/* The two forms of the device image.  PACKED (every loader writes it): a word = two 28-bit residues in 7 bytes, streamed by the vector-ALU
 * sweep (the single-query path).  LIMBS: every residue as three signed bytes and a 4-bit top limb, still 3.5 bytes, laid out as MFMA operands
 * for the batched sweep on the matrix cores (csrc/sweep_mfma.hip); single queries then sweep it with the one-query instance of the same kernel.
 * set_db_format converts the holder's image in place through a bounded staging buffer (the maps are bijective: packed -> limbs -> packed
 * reproduces every byte), so one image serves both kernels whatever the database's size; a batch converts to LIMBS by itself (option
 * "one_image"), a partial load_db_items and set_sweep_stages(K > 1) convert back.  LIMBS exists where the matrix-core sweep does (>= 64
 * ciphertexts per slot, the shard's first dimension a power of two in [64, 2048]); elsewhere set_db_format(LIMBS) fails.  Call it on the
 * image's owner (not on a lane), outside stream capture; the lanes' captured graphs are re-captured by themselves.
 * db_device_bytes: device memory the holder of this server's image keeps for database images (one image, unless "one_image" is 0). */
enum spiral_gpu_db_format { SPIRAL_GPU_DB_PACKED = 0, SPIRAL_GPU_DB_LIMBS = 1 };
int spiral_gpu_server_set_db_format(spiral_gpu_server *s, int format);
int spiral_gpu_server_db_format(spiral_gpu_server *s);
uint64_t spiral_gpu_server_db_device_bytes(spiral_gpu_server *s);
/* --random-data analogue: arbitrary valid NTT-form words, timing only */
int spiral_gpu_server_fill_db_random(spiral_gpu_server *s, uint64_t seed);
/* a second in-flight query on one database: `s` releases its own image and sweeps `owner`'s (same parameters, shard and device;
 * the owner does not reload while `s` answers; loads through `s` fail).  One handle per query lane, each on
 * its own stream: the latency-bound expansion / folding of one query runs under the HBM-bound sweep of another.
 * Lifetime: the owner counts its lanes.  spiral_gpu_server_destroy(owner) while lanes exist frees everything of the owner except the
 * image and invalidates the handle; the image itself is freed with the last lane, so a lane never sweeps freed memory. */
int spiral_gpu_server_share_db(spiral_gpu_server *s, spiral_gpu_server *owner);
/* the same in one step and without ever allocating a second image: a new server with `owner`'s parameters, device and shard
 * whose database IS the owner's (a query lane).  Works for images larger than half of HBM, where create + share_db cannot. */
int spiral_gpu_server_create_lane(spiral_gpu_server *owner, spiral_gpu_server **out);

/* public parameters (NTT form): W_exp_left g x (n0 x t_exp), W_exp_right n_right x (n0 x t_exp_right),
 * W n1 x (n0*t_conv), V n1 x (2*t_conv)   (src/spiral.cpp:2091-2092, 2216-2227, 2279-2296) */
int spiral_gpu_server_set_pub_params(spiral_gpu_server *s, const uint64_t *w_left, const uint64_t *w_right,
                                     const uint64_t *w, const uint64_t *v);
/* query: n_query_cts Regev ciphertexts, n0 x 1 NTT form */
int spiral_gpu_server_set_query(spiral_gpu_server *s, const uint64_t *query);

/* stages, asynchronous on the server stream */
int spiral_gpu_server_expand(spiral_gpu_server *s);    /* expandImproved + reorderFromStopround      */
int spiral_gpu_server_convert(spiral_gpu_server *s);   /* scalToMat x dim0, regevToGSW x nu2 (Q_neg = G2 - Q is derived where a fold round needs it) */
int spiral_gpu_server_first_dim(spiral_gpu_server *s); /* multiplyQueryByDatabase on this shard       */
int spiral_gpu_server_lift(spiral_gpu_server *s, int reduce_first); /* nttInvAndCrtLiftCiphertexts     */
/* Throughput, beyond the reference (which answers one query at a time): multiplyQueryByDatabase for the queries of n <= 8
 * servers that share one database image (an owner and its lanes, create_lane) in ONE pass over the database -- server b's
 * converted query against the image into server b's accumulators, each bit-identical to its own first_dim().  The pass runs on
 * the matrix cores (csrc/sweep_mfma.hip: both operands as signed 8-bit limbs, v_mfma_i32_16x16x64_i8, exact recombination mod
 * the primes): at config 2 two to five queries take the time of one (0.30 ms), eight take 0.39 ms.  It reads the database as "limb
 * planes": the image's holder converts its image to that form IN PLACE the first time a batch needs it (option "one_image"; no second
 * image -- see spiral_gpu_server_set_db_format) and again after the database is reloaded; option "sweep_mfma_min" = 0 turns it off.
 * Without it (that option, fewer than 128 output columns, a first dimension -- of this shard -- that is not a power of two in [64, 2048])
 * the call makes passes of two queries on the vector ALU.  Asynchronous:
 * the launch runs on servers[0]'s stream and the other lanes' streams are ordered around it with events, so per lane the
 * sequence run_pre(lane) ... first_dim_batch(all) ... run_post(lane, 0) needs no host synchronisation.  Pays where the sweep is
 * most of a query (large databases); a single query's latency is first_dim().
 * Every server is checked (same image and layout, database present, query converted since its last set_query) before anything is
 * launched: a failing call leaves no lane swept.  Geometries with fewer than 64 output columns (nu2 <= 4: 2 num_per < 64) or without the
 * packed database layout have no batched kernel: the call then runs one first_dim() per server, in order -- same results, no shared pass.
 * The lanes are ordered with hipEventRecord / hipStreamWaitEvent on their streams (a lane on servers[0]'s own stream needs none and
 * costs none): call it OUTSIDE stream capture (none of the lanes' streams may be capturing a hipGraph; run_pre / run_post capture
 * and replay their own groups either side of it). */
int spiral_gpu_server_first_dim_batch(spiral_gpu_server *const *servers, uint32_t n);
/* The same idea for the WHOLE answer: n <= 8 queries -- one per server, an owner and its lanes (create_lane), equal parameters, each with its
 * own client's public parameters and query -- as one launch sequence in which every launch carries all n queries: the expansion, conversion,
 * lift, folding and switch kernels take a query dimension (the reference runs them once per query, src/spiral.cpp:1664-1743, 1850-2025,
 * 1349-1410, inside process_crtd_query :2337-2406) and the sweep is first_dim_batch's: one pass over the database for all n on the matrix
 * cores.  A query's ~50 dependent launches outside the sweep are
 * launch-bound (~5 us each whatever they carry), so n queries cost little more than one there.  Throughput only: each query's latency is the
 * batch's.  Afterwards every server's buffers (accumulators, GSW matrices, final ciphertext, response) hold exactly what its own run_query
 * would have left.  Runs on servers[0]'s stream -- one hipGraph replay per batch when servers[0] has use_graphs on -- with the other lanes'
 * streams ordered around it by events (call it outside stream capture; ~20 us per batch and other stream: put the lanes of a batch on one stream).  Needs the default schedule on every server: own accumulators, no
 * keep_cts, no split / sharded / staged options; every server is checked before anything is launched.  n = 1 is run_query. */
int spiral_gpu_server_run_query_batch(spiral_gpu_server *const *servers, uint32_t n);
/* One query against n INSTANCES of the database.  An item larger than one plaintext -- configs[3]: 100 KB items of 15 360-byte plaintexts -- is
 * factor = ceil(item size / plaintext size) database instances (select_params.py:297-298): the client sends ONE query, the server expands and converts it
 * once and answers it against every instance -- first dimension, folding and response switch per instance, `factor` responses.  (The reference runs a
 * single instance and multiplies its first-dimension time, folding time and response size by the factor, select_params.py:409-418.)  `s` holds the query;
 * instances[k] hold the images: servers of the same geometry on the same device, each with its own database (s may be one of them; lanes are fine).
 * pre != 0 runs expansion + conversion first, else s must have converted its query (run_pre).  responses (device pointer): n x 6 x 2048 words, instance k's
 * switched response n1 x n2 at k * 6 * 2048; finals (device pointer or NULL): the folded ciphertexts likewise.  One launch sequence on s's stream, a
 * hipGraph per instance set when s has use_graphs on.  Across GPUs the instances are independent: rank r holds instances r, r + N, ... and the responses are
 * gathered -- no reduce (spiral_amd/dist.py all_gather_instance_responses). */
int spiral_gpu_server_run_query_instances(spiral_gpu_server *s, spiral_gpu_server *const *instances, uint32_t n, int pre,
                                          void *responses, void *finals);
/* the same from host buffers (as spiral_gpu_server_answer): the query in, the n responses (n x n1 x n2 x 2048 words) and optionally the folded
 * ciphertexts out; total_us (may be NULL): device time of the whole item query */
int spiral_gpu_server_answer_instances(spiral_gpu_server *s, spiral_gpu_server *const *instances, uint32_t n,
                                       const uint64_t *query, uint64_t *responses, uint64_t *finals, double *total_us);
int spiral_gpu_server_fold(spiral_gpu_server *s);      /* foldOneFurtherDimension x nu2               */
int spiral_gpu_server_finish(spiral_gpu_server *s);    /* response modulus switch, :1441-1447         */
int spiral_gpu_server_sync(spiral_gpu_server *s);
/* stage groups either side of the (possibly distributed) first-dimension reduce: run_pre = expand + convert,
 * run_post = lift + fold + finish.  With use_graphs on, each group is captured once into a hipGraph on the
 * server stream (which must not be the default stream) and replayed afterwards. */
int spiral_gpu_server_use_graphs(spiral_gpu_server *s, int on);
int spiral_gpu_server_run_pre(spiral_gpu_server *s);
/* Schedule option, on = 0 (default: one stream) or 2 (the split schedule; any other value is an error).  After round 0 the even-index and
 * the odd-index trees of expandImproved never read each other (a ciphertext is created from the one 2^r slots below it,
 * src/spiral.cpp:1709), and with stopround > 0 the evens are the first-dimension ciphertexts and the odds the GSW bits -- so the whole GSW
 * side of a query (odd tree + regevToGSW + fold keys) runs as its own launch sequence on an internal side stream, forked when the query is
 * set, beside the even tree + scalToMat + sweep on the server stream; fold / fold_local / fold_root / run_post / sync join it.  run_query
 * then issues three launch groups on two streams instead of one graph.  Needs query compression with stopround > 0 and an unsharded
 * expansion.  Results are identical, only the schedule changes; measured within noise of the in-order schedule
 * (profiles/r04_split_overlap.txt; the forms that forked only the conversion, under the sweep, were slower and are gone). */
int spiral_gpu_server_set_overlap(spiral_gpu_server *s, int on);
int spiral_gpu_server_run_post(spiral_gpu_server *s, int reduce_first);
/* the whole single-GPU answer (run_pre, first_dim, run_post(0)) as one group: with use_graphs on, one hipGraph launch
 * per query and no host-visible seam between the stages */
int spiral_gpu_server_run_query(spiral_gpu_server *s);
/* run_pre + first_dim as one group: everything a rank does before the reduce of a sharded answer */
int spiral_gpu_server_run_pre_sweep(spiral_gpu_server *s);
/* per-shard first-dimension accumulators: num_per*n1*n2*2048 packed words (p-limb | b-limb << 32, each
 * field < 2^28).  Summing the shards' buffers as uint64 (one RCCL reduce) and calling lift with
 * reduce_first = 1 gives the unsharded result.  Returns a device pointer. */
void *spiral_gpu_server_acc(spiral_gpu_server *s, size_t *bytes);
/* Distributed folding over G = 2^k ranks (reduce-scatter variant).  set_fold_ranks(G) makes the
 * sweep group its accumulators by ii mod G, so that ONE reduce-scatter of the acc buffers hands rank g the summed
 * chunk of the num_per/G ciphertexts ii = g + G*k.  fold_local lifts that chunk (device pointer, packed words) and
 * runs the first nu2-k folding rounds, leaving one raw n1 x n2 ciphertext in out_ct (device pointer, 6*2048 words);
 * after an all-gather of those G ciphertexts in rank order, fold_root runs the last k rounds and the response
 * switch on the root.  Bit-identical to lift + fold on one device. */
int spiral_gpu_server_set_fold_ranks(spiral_gpu_server *s, uint32_t n_ranks);
int spiral_gpu_server_fold_local(spiral_gpu_server *s, const void *acc_chunk, void *out_ct);
int spiral_gpu_server_fold_root(spiral_gpu_server *s, const void *gathered_cts);
/* Sharded expansion over G = 2^k ranks (the query expansion is database-independent, so a G-GPU answer would otherwise repeat all
 * of it on every GPU).  set_expand_shard(rank, G): expand() then computes only what this rank needs of expandImproved's tree
 * (src/spiral.cpp:1664-1743): the first-dimension ciphertexts of its own j-range [rank dim0/G, (rank+1) dim0/G) -- the server
 * must have been created on exactly that range -- and every G-th GSW-bit ciphertext (i = rank mod G).  All ranks need all GSW
 * bits (regevToGSW, :2315-2331), so they are exchanged: gsw_bits_pack writes this rank's block (gsw_bits_words() uint64 words,
 * device pointer), ONE all-gather of the blocks in rank order, gsw_bits_unpack stores the gathered blocks; then convert() as
 * usual.  run_expand_pack / run_unpack_convert_sweep are those steps as launch groups (hipGraphs with use_graphs on) either
 * side of the all-gather.  Results are identical to the unsharded expansion.  Needs query compression with stopround > 0. */
int spiral_gpu_server_set_expand_shard(spiral_gpu_server *s, uint32_t rank, uint32_t n_ranks);
size_t spiral_gpu_server_gsw_bits_words(spiral_gpu_server *s);
int spiral_gpu_server_gsw_bits_pack(spiral_gpu_server *s, void *block_out);
int spiral_gpu_server_gsw_bits_unpack(spiral_gpu_server *s, const void *gathered);
int spiral_gpu_server_run_expand_pack(spiral_gpu_server *s, void *bits_out);
int spiral_gpu_server_run_unpack_convert_sweep(spiral_gpu_server *s, const void *gathered);
/* the same work split so that the all-gather can overlap the database-dependent part: run_scal2mat_sweep (ScalToMat + sweep:
 * needs no GSW bit) while the all-gather is in flight, run_unpack_gsw (unpack + regevToGSW + fold keys) once it has landed */
int spiral_gpu_server_run_scal2mat_sweep(spiral_gpu_server *s);
int spiral_gpu_server_run_unpack_gsw(spiral_gpu_server *s, const void *gathered);
/* Pipelined sweep for N > 1.  The output columns of multiplyQueryByDatabase are independent (src/spiral.cpp:628-999: the i and c
 * loops enclose the j loop), so a rank can sweep them in K = 2^k stages of num_per/K ciphertexts and start the reduce-scatter of
 * a stage's accumulators while the next stage streams the database.  set_sweep_stages(K) -- after set_fold_ranks -- lays the
 * accumulator buffer out [stage][rank][ciphertext]: stage s is the contiguous 1/K of the buffer at offset s/K, and reduce-scattering
 * it over the G ranks gives rank g rows [s L/K, (s+1) L/K) of the chunk fold_local expects (L = num_per/G), so K reduce-scatters
 * of 1/K each replace the one.  first_dim_stage(s) launches stage s only; first_dim() all stages at once (same layout).
 * K must leave whole 64-column blocks per stage (K <= num_per/32, max_sweep_stages) and needs the packed database layout.
 * run_scal2mat: ScalToMat alone (run_scal2mat_sweep without the sweep), after which the stages are issued one by one. */
int spiral_gpu_server_set_sweep_stages(spiral_gpu_server *s, uint32_t n_stages);
uint32_t spiral_gpu_server_max_sweep_stages(spiral_gpu_server *s);
int spiral_gpu_server_first_dim_stage(spiral_gpu_server *s, uint32_t stage);
int spiral_gpu_server_run_scal2mat(spiral_gpu_server *s);
/* make the sweep write into caller-owned device memory (e.g. a torch tensor) */
int spiral_gpu_server_set_acc(spiral_gpu_server *s, void *device_ptr);

/* whole path for one query: set_query, expand, convert, first_dim, lift, fold, finish, sync.
 * final_ct: raw n1 x n2 (may be NULL); response: rescaled n1 x n2 (may be NULL).
 * stage_us (may be NULL): [0] expansion [1] conversion [2] first-dimension multiply (sweep + lift)
 * [3] folding [4] response switch [5] sweep kernel alone [6] total device time [7] ScalToMat share of [1],
 * the reference's buckets of src/spiral.cpp:246-257, measured with HIP events on the server stream. */
int spiral_gpu_server_answer(spiral_gpu_server *s, const uint64_t *query, uint64_t *final_ct,
                             uint64_t *response, double stage_us[8]);
/* the same without touching host memory: the query must have been set; results stay on the device */
int spiral_gpu_server_answer_resident(spiral_gpu_server *s, double stage_us[8]);

/* read back intermediates in reference layouts (stage parity tests) */
enum spiral_gpu_buffer {
    SPIRAL_GPU_BUF_EXPANDED = 0, /* n_bits cts n0 x 1 NTT, in the order scalToMat/regevToGSW consume them */
    SPIRAL_GPU_BUF_CTS = 1,      /* (j_end-j_begin) cts n1 x n0 NTT = expansionLocals.cts (needs keep_cts)  */
    SPIRAL_GPU_BUF_GSW = 2,      /* nu2 matrices n1 x m2 NTT, reference's reversed order (:2324)           */
    SPIRAL_GPU_BUF_ACC = 3,      /* num_per cts n1 x n2 NTT: the sweep output                              */
    SPIRAL_GPU_BUF_RAW = 4,      /* num_per cts n1 x n2 raw: after lift / after folding rounds             */
    SPIRAL_GPU_BUF_FINAL = 5,    /* n1 x n2 raw                                                            */
    SPIRAL_GPU_BUF_RESPONSE = 6  /* n1 x n2 rescaled                                                       */
};
int spiral_gpu_server_keep_cts(spiral_gpu_server *s, int on);
size_t spiral_gpu_server_buffer_words(spiral_gpu_server *s, int which);
int spiral_gpu_server_read(spiral_gpu_server *s, int which, uint64_t *out);
/* the last answer's response in its wire form (see spiral_gpu_response_wire_bytes above): packed on the device, downloaded
 * into `out` (capacity in bytes; fails when it is smaller than the wire form) */
int spiral_gpu_server_read_response_wire(spiral_gpu_server *s, void *out, size_t capacity);
/* overwrite the lifted ciphertexts (raw [num_per][n1][n2][N]) -- lets a test drive fold() alone */
int spiral_gpu_server_write_raw(spiral_gpu_server *s, const uint64_t *raw_cts);

/* measurement helper: average duration (ms) of the sweep kernel alone over `iters` launches, timed
 * with HIP events on the server stream */
int spiral_gpu_server_time_sweep(spiral_gpu_server *s, int iters, float *avg_ms);
/* the same for first_dim_batch's kernel launch alone: the (already converted) queries of the n servers, `iters` launches */
int spiral_gpu_server_time_sweep_batch(spiral_gpu_server *const *servers, uint32_t n, int iters, float *avg_ms);
/* algorithmic bytes of one sweep on this shard: DB + query records + accumulators */
uint64_t spiral_gpu_server_sweep_bytes(spiral_gpu_server *s);
/* bytes one launch actually has to move on this device: the database in its device layout (two 28-bit residues
 * packed in 7 bytes), the query records and the accumulators -- below the algorithmic figure above */
uint64_t spiral_gpu_server_sweep_device_bytes(spiral_gpu_server *s);

/* ------------------------------------------------------------------------------------------------
 * SpiralPack / SpiralStreamPack (`--high-rate`, src/testing.cpp): base_dim x 1 scalar Regev ciphertexts,
 * 1 x 1 plaintexts, out_n^2 database trials packed into one (out_n+1) x out_n response.
 * ------------------------------------------------------------------------------------------------ */
typedef struct spiral_gpu_pack_shape {
    uint32_t dim0, num_per, ell, g, stopround;
    uint32_t n_left, n_right; /* expansion key-switching matrices n0 x t_exp / n0 x t_exp_right (0 with direct upload) */
    uint32_t n_query_cts;     /* 1, or dim0 + nu2*2*ell uploaded ciphertexts */
    uint32_t trials;          /* out_n^2 */
    uint64_t qprime;
} spiral_gpu_pack_shape;
typedef struct spiral_gpu_pack_server spiral_gpu_pack_server;
int spiral_gpu_pack_get_shape(const spiral_gpu_params *p, uint32_t out_n, spiral_gpu_pack_shape *out);

/* pack, include/testing.h:36-42, src/testing.cpp:198.  v_ct: out_n^2 raw base_dim x 1 ciphertexts; v_W: out_n
 * matrices (out_n+1) x m_conv NTT; result: (out_n+1) x out_n NTT */
int spiral_gpu_pack(uint64_t *result, uint32_t out_n, uint32_t m_conv, const uint64_t *v_ct, const uint64_t *v_W);
/* fastMultiplyQueryByDatabaseDim1, src/testing.cpp:364.  db: convertDb's layout (:316-340); v_firstdim:
 * reorientCiphertextsDim1's layout (:342-362); out: num_per ciphertexts base_dim x 1 NTT */
int spiral_gpu_fast_multiply_query_by_database_dim1(uint64_t *out, const uint64_t *db, const uint64_t *v_firstdim,
                                                    size_t dim0, size_t num_per);

/* resident server for testHighRate's server half (src/testing.cpp:1009-1081) */
int spiral_gpu_pack_server_create(const spiral_gpu_params *p, uint32_t out_n, int device, spiral_gpu_pack_server **out);
/* N GPUs (pack variant): the out_n^2 trials are independent up to the packing step, so the ranks split THEM -- a server
 * for trials [trial0, trial1) holds only those database images (trial indices in load_db / load_db_items / read_acc stay global).
 * Per query: every rank runs fold_trials (expansion + conversion, replicated; its trials' sweeps and folding; leaves their folded
 * ciphertexts, [n_local][2][N] raw words, in the caller's device buffer), ONE all-gather of out_n^2 x 32 KiB collects them in trial
 * order, the root runs pack_gathered (pack + modulus switch).  No reduction of accumulators is needed: the j-shard + reduce of the
 * base path would move out_n^2 x num_per x 32 KiB here (128 MiB at configs[4]) over point-to-point xGMI, this moves 512 KiB.
 * (0, 0) = all trials = spiral_gpu_pack_server_create.  set_stream: run on the caller's stream (the one its collectives use). */
int spiral_gpu_pack_server_create_sharded(const spiral_gpu_params *p, uint32_t out_n, int device, uint32_t trial0, uint32_t trial1,
                                          spiral_gpu_pack_server **out);
int spiral_gpu_pack_server_set_stream(spiral_gpu_pack_server *s, void *hip_stream);
int spiral_gpu_pack_server_fold_trials(spiral_gpu_pack_server *s, const uint64_t *query, void *folded_dev);
/* stage times (as answer's stage_us) of the last answer or fold_trials from the events between its stages; synchronises */
int spiral_gpu_pack_server_stage_us(spiral_gpu_pack_server *s, double stage_us[8]);
int spiral_gpu_pack_server_pack_gathered(spiral_gpu_pack_server *s, const void *gathered_dev, uint64_t *response, uint64_t *packed_ct);
void spiral_gpu_pack_server_destroy(spiral_gpu_pack_server *s);
/* the out_n^2 trial databases: seeded explicit data generated on the device (coefficient z of item i of trial t =
 * splitmix64(seed ^ ((t*n + i)*N + z)) % p_db), one trial from host memory in convertDb's layout, or arbitrary words */
int spiral_gpu_pack_server_gen_db(spiral_gpu_pack_server *s, uint64_t seed);
int spiral_gpu_pack_server_load_db(spiral_gpu_pack_server *s, uint32_t trial, const uint64_t *db);
int spiral_gpu_pack_server_fill_db_random(spiral_gpu_pack_server *s, uint64_t seed);
/* raw ingest of one trial (src/testing.cpp:845-869 + convertDb :316-340 on the device): 1 x 1 plaintexts of 2048
 * coefficients, bit-packed as for spiral_gpu_server_load_db_items */
int spiral_gpu_pack_server_load_db_items(spiral_gpu_pack_server *s, uint32_t trial, const void *items, uint32_t coeff_bits,
                                         uint64_t first_item, uint64_t n_items);
/* W_exp_left / W_exp_right (expansion only), V base_dim x base_dim*t_conv (expansion only), v_W out_n x ((out_n+1) x t_conv) */
int spiral_gpu_pack_server_set_pub_params(spiral_gpu_pack_server *s, const uint64_t *w_left, const uint64_t *w_right,
                                          const uint64_t *v, const uint64_t *v_w);
/* response: (out_n+1) x out_n raw, row 0 mod q', rows 1.. mod 4p; packed_ct (may be NULL): the (out_n+1) x out_n NTT
 * ciphertext before the modulus switch.  stage_us (may be NULL): [0] expansion [1] conversion [2] first dimension
 * (out_n^2 sweeps + lift) [3] folding [4] packing + modulus switch [5] the sweep kernels alone [6] total. */
int spiral_gpu_pack_server_answer(spiral_gpu_pack_server *s, const uint64_t *query, uint64_t *response, uint64_t *packed_ct,
                                  double stage_us[8]);
/* the last answer's response in its wire form ((out_n+1) x out_n, spiral_gpu_response_wire_bytes(p, out_n) bytes) */
int spiral_gpu_pack_server_read_response_wire(spiral_gpu_pack_server *s, void *out, size_t capacity);
/* the first-dimension accumulators of one trial of the last answer (fastMultiplyQueryByDatabaseDim1's output, :1050):
 * num_per ciphertexts base_dim x 1, NTT form (tests) */
int spiral_gpu_pack_server_read_acc(spiral_gpu_pack_server *s, uint32_t trial, uint64_t *out);
uint64_t spiral_gpu_pack_server_sweep_bytes(spiral_gpu_pack_server *s); /* algorithmic bytes of ONE trial's sweep */

#ifdef __cplusplus
}
#endif
#endif
