// SpiralPack / SpiralStreamPack: host orchestration and C ABI (server half of testHighRate, reference
// src/testing.cpp:1009-1081).  Kernels: pack.hip, ntt.hip (LD_PDIGIT / LD_DBGEN1), poly.hip (matmul, rescale).
#include "host_common.h"

using namespace spiral;
using namespace spiral::host;

struct spiral_gpu_pack_server {
    spiral_gpu_params p;
    spiral_gpu_pack_shape s;
    uint32_t out_n = 0;
    uint32_t t0 = 0, nt = 0;  // this server's trials [t0, t0 + nt) of the out_n^2 (all of them unless created sharded)
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = true;
    DeviceTables tb;
    bool have_db = false, have_pp = false;
    bool packed_after_front = false;  // event 6 belongs to the same answer as events 0..5
    uint32_t n_cv = 0;
    size_t db_words = 0;  // per trial
    DevBuf db, w_left, w_right, v, v_w, query, cv, ex_raw, ex_g;
    DevBuf gs_raw, gs_chat, gs_tmp, gsw, key, qs1, acc, raw, fold_d, fold_c, fold_c2, pk_ginv, pk_ct2, pk_res, pk_raw, resp, stage, wire;
    hipEvent_t ev[8] = {};
};

namespace {

int pack_shape_of(const spiral_gpu_params* p, uint32_t out_n, spiral_gpu_pack_shape* s) {  // src/testing.cpp:777-801
    spiral_gpu_shape base;
    spiral_gpu_params q = *p;
    q.direct_upload = 1;  // validate the common fields without the base path's query-size rule
    if (shape_of(&q, &base)) return -1;
    if (out_n < 1 || out_n > 16) return fail("out_n out of range");
    if (p->nu2 < 1 || p->nu1 < 1) return fail("SpiralPack needs nu1 >= 1 and nu2 >= 1");
    s->dim0 = base.dim0;
    s->num_per = base.num_per;
    s->ell = base.ell;
    s->trials = out_n * out_n;
    s->qprime = base.qprime;
    if (p->direct_upload) {
        s->g = s->stopround = s->n_left = s->n_right = 0;
        s->n_query_cts = s->dim0 + p->nu2 * 2 * s->ell;
    } else {
        s->g = ceil_log2((uint64_t)s->ell * p->nu2 + s->dim0);
        s->stopround = ceil_log2((uint64_t)s->ell * p->nu2);
        s->n_left = s->g;
        s->n_right = s->stopround + 1;
        s->n_query_cts = 1;
        if (s->g > kLogN || s->stopround == 0 || s->stopround >= s->g) return fail("query does not fit one polynomial");
    }
    return 0;
}

void pk_free(spiral_gpu_pack_server* S) {
    DevBuf* all[] = {&S->db, &S->w_left, &S->w_right, &S->v, &S->v_w, &S->query, &S->cv, &S->ex_raw, &S->ex_g, &S->gs_raw, &S->gs_chat, &S->gs_tmp,
                     &S->gsw, &S->key, &S->qs1, &S->acc, &S->raw, &S->fold_d, &S->fold_c, &S->fold_c2, &S->pk_ginv, &S->pk_ct2, &S->pk_res, &S->pk_raw, &S->resp,
                     &S->stage, &S->wire};
    for (DevBuf* b : all) b->release();
    for (auto& e : S->ev)
        if (e) (void)hipEventDestroy(e);
    if (S->stream && S->own_stream) (void)hipStreamDestroy(S->stream);
}

int pk_alloc(spiral_gpu_pack_server* S) {
    const spiral_gpu_params& p = S->p;
    const spiral_gpu_pack_shape& s = S->s;
    const size_t ngs = (size_t)p.nu2 * s.ell, rows = S->out_n + 1;
    S->db_words = db1_device_words(s.num_per, s.dim0);  // u64 words of one trial in the device layout
    if (S->db.alloc(S->db_words * S->nt)) return -1;
    if (S->w_left.alloc((size_t)s.n_left * 2 * p.t_exp * kN)) return -1;
    if (S->w_right.alloc((size_t)s.n_right * 2 * p.t_exp_right * kN)) return -1;
    if (S->v.alloc((size_t)2 * 2 * p.t_conv * kN)) return -1;
    if (S->v_w.alloc((size_t)S->out_n * rows * p.t_conv * kN)) return -1;
    if (S->query.alloc((size_t)s.n_query_cts * 2 * kN)) return -1;
    S->n_cv = p.direct_upload ? s.n_query_cts : (1u << s.g);
    if (S->cv.alloc((size_t)S->n_cv * 2 * kN)) return -1;
    HIP_OK(hipMemset(S->cv.p, 0, S->cv.words * sizeof(uint64_t)));
    if (!p.direct_upload) {
        if (S->ex_raw.alloc((size_t)S->n_cv * 2 * kN)) return -1;
        if (S->ex_g.alloc(expand_g_polys(s.g, p.t_exp, p.t_exp_right) * kN)) return -1;
        if (S->gs_raw.alloc(ngs * 2 * kN)) return -1;
        if (S->gs_chat.alloc(ngs * 2 * p.t_conv * kN)) return -1;
        if (S->gs_tmp.alloc(ngs * 2 * kN)) return -1;
    }
    if (S->gsw.alloc((size_t)p.nu2 * 2 * 2 * s.ell * kN)) return -1;
    if (S->key.alloc((size_t)p.nu2 * 2 * 4 * s.ell * kN)) return -1;
    if (S->qs1.alloc((size_t)kN * s.dim0 * 2)) return -1;  // 4 u32 per (z, j)
    if (S->acc.alloc((size_t)S->nt * s.num_per * 2 * kN)) return -1;
    if (S->raw.alloc((size_t)S->nt * s.num_per * 2 * kN)) return -1;
    const size_t half = s.num_per / 2;
    if (S->fold_d.alloc((size_t)S->nt * half * 4 * s.ell * kN)) return -1;
    if (S->fold_c.alloc((size_t)S->nt * half * 2 * kN)) return -1;
    if (S->fold_c2.alloc((size_t)S->nt * half * 2 * kN)) return -1;
    if (S->pk_ginv.alloc((size_t)s.trials * p.t_conv * kN)) return -1;
    if (S->pk_ct2.alloc((size_t)s.trials * kN)) return -1;
    if (S->pk_res.alloc(rows * S->out_n * kN)) return -1;
    if (S->pk_raw.alloc(rows * S->out_n * kN)) return -1;
    if (S->resp.alloc(rows * S->out_n * kN)) return -1;
    return 0;
}

int pk_upload_ref_ntt(spiral_gpu_pack_server* S, const uint64_t* host, uint64_t* pk, size_t npolys) {
    if (npolys == 0) return 0;
    if (!host) return fail("null host buffer");
    const size_t chunk = 4096;
    if (S->stage.words < std::min(npolys, chunk) * kRefNtt) {
        S->stage.release();
        if (S->stage.alloc(std::min(npolys, chunk) * kRefNtt)) return -1;
    }
    for (size_t done = 0; done < npolys; done += chunk) {
        const size_t n = std::min(chunk, npolys - done);
        HIP_OK(hipMemcpyAsync(S->stage.p, host + done * kRefNtt, n * kRefNtt * sizeof(uint64_t), hipMemcpyHostToDevice, S->stream));
        launch_ref_to_pk(S->stage.p, pk + done * kN, (uint32_t)n, identity_map(), S->stream);
        HIP_OK(hipStreamSynchronize(S->stream));
    }
    return 0;
}

// pack (src/testing.cpp:198-241) on device buffers: raw cts at trial stride `ct_stride` polynomials
void run_pack(const DeviceTables& tb, const uint64_t* raw_cts, uint32_t ct_stride_cts, const uint64_t* v_w, uint64_t* ginv, uint64_t* ct2, uint64_t* result,
              uint32_t out_n, uint32_t t_conv, hipStream_t st) {
    const uint32_t trials = out_n * out_n;
    FwdParams fp{};
    fp.src = raw_cts;
    fp.dst = ginv;
    fp.src_map = fp.dst_map = identity_map();
    fp.n_digits = t_conv;
    fp.bits = get_bits_per(t_conv);
    fp.pmode = PM_PACK;
    fp.pk_num_per = ct_stride_cts;
    launch_ntt_forward(tb, fp, LD_PDIGIT, ST_PK, trials * t_conv, st);
    FwdParams fa{};
    fa.src = raw_cts;
    fa.dst = ct2;
    fa.src_map = IndexMap{1, 2 * ct_stride_cts, 1};
    fa.dst_map = identity_map();
    fa.n_digits = 1;
    launch_ntt_forward(tb, fa, LD_RAW, ST_PK, trials, st);
    launch_pack_mac(v_w, ginv, ct2, result, out_n, t_conv, st);
}

}  // namespace

extern "C" {

int spiral_gpu_pack_get_shape(const spiral_gpu_params* p, uint32_t out_n, spiral_gpu_pack_shape* out) {
    if (!p || !out) return fail("null argument");
    return pack_shape_of(p, out_n, out);
}

int spiral_gpu_pack(uint64_t* result, uint32_t out_n, uint32_t m_conv, const uint64_t* v_ct, const uint64_t* v_W) {
    DeviceTables tb;
    if (current_tables(&tb)) return -1;
    if (out_n < 1 || out_n > 16 || m_conv < 1 || m_conv > 56) return fail("bad pack dimensions");
    Scratch sc;
    const uint32_t trials = out_n * out_n, rows = out_n + 1;
    uint64_t* d_ct = sc.upload(v_ct, (size_t)trials * 2 * kN);
    uint64_t* d_w = upload_pk(sc, v_W, (size_t)out_n * rows * m_conv);
    uint64_t* d_g = sc.get((size_t)trials * m_conv * kN);
    uint64_t* d_c2 = sc.get((size_t)trials * kN);
    uint64_t* d_res = sc.get((size_t)rows * out_n * kN);
    if (!d_ct || !d_w || !d_g || !d_c2 || !d_res) return fail("device allocation/upload failed");
    run_pack(tb, d_ct, 1, d_w, d_g, d_c2, d_res, out_n, m_conv, 0);
    return download_pk(sc, d_res, identity_map(), result, (size_t)rows * out_n);
}

int spiral_gpu_fast_multiply_query_by_database_dim1(uint64_t* out, const uint64_t* db, const uint64_t* v_firstdim, size_t dim0, size_t num_per) {
    if (dim0 < 2 || (dim0 & 1) || num_per == 0) return fail("unsupported geometry");
    Scratch sc;
    const size_t words = (size_t)kN * dim0 * num_per;
    uint64_t* d_ref = sc.upload(db, words);
    uint64_t* d_db = sc.get(db1_device_words((uint32_t)num_per, (uint32_t)dim0));
    uint64_t* d_re = sc.upload(v_firstdim, (size_t)kN * dim0 * 2);
    uint64_t* d_qs = sc.get((size_t)kN * dim0 * 2);
    uint64_t* d_acc = sc.get(num_per * 2 * kN);
    if (!d_ref || !d_db || !d_re || !d_qs || !d_acc) return fail("device allocation/upload failed");
    launch_db1_relayout(d_ref, d_db, (uint32_t)num_per, (uint32_t)dim0, 0);
    launch_qs1_from_reoriented(d_re, (uint32_t*)d_qs, (uint32_t)dim0, 0);
    launch_sweep1(d_db, (const uint32_t*)d_qs, d_acc, (uint32_t)num_per, (uint32_t)dim0, 1, 0, 0, 0);
    return download_pk(sc, d_acc, identity_map(), out, num_per * 2);
}

int spiral_gpu_pack_server_create(const spiral_gpu_params* p, uint32_t out_n, int device, spiral_gpu_pack_server** out) {
    return spiral_gpu_pack_server_create_sharded(p, out_n, device, 0, 0, out);
}

// The out_n^2 trials are independent up to the packing step (each has its own database image, sweep and folding), so N GPUs split
// them: a server created for trials [trial0, trial1) holds only those images; fold_trials leaves their folded ciphertexts in a
// caller-provided device buffer, one all-gather of out_n^2 x 2 polynomials collects them, pack_gathered finishes on the root.
int spiral_gpu_pack_server_create_sharded(const spiral_gpu_params* p, uint32_t out_n, int device, uint32_t trial0, uint32_t trial1,
                                          spiral_gpu_pack_server** out) {
    if (!p || !out) return fail("null argument");
    spiral_gpu_pack_shape s;
    if (pack_shape_of(p, out_n, &s)) return -1;
    if (trial0 == 0 && trial1 == 0) trial1 = s.trials;
    if (trial0 >= trial1 || trial1 > s.trials) return fail("bad trial range [%u, %u) of %u", trial0, trial1, s.trials);
    HIP_OK(hipSetDevice(device));
    auto* S = new spiral_gpu_pack_server();
    S->p = *p;
    S->s = s;
    S->out_n = out_n;
    S->t0 = trial0;
    S->nt = trial1 - trial0;
    S->device = device;
    if (tables_get(device, &S->tb) != 0) {
        delete S;
        return fail("twiddle table setup failed on device %d", device);
    }
    bool ok = hipStreamCreate(&S->stream) == hipSuccess;
    for (auto& e : S->ev) ok = ok && hipEventCreate(&e) == hipSuccess;
    if (!ok || pk_alloc(S)) {
        if (ok == false) fail("stream/event creation failed");
        pk_free(S);
        delete S;
        return -1;
    }
    *out = S;
    return 0;
}

void spiral_gpu_pack_server_destroy(spiral_gpu_pack_server* S) {
    if (!S) return;
    (void)hipSetDevice(S->device);
    (void)hipDeviceSynchronize();
    pk_free(S);
    delete S;
}

int spiral_gpu_pack_server_gen_db(spiral_gpu_pack_server* S, uint64_t seed) {
    if (!S) return fail("null server");
    HIP_OK(hipSetDevice(S->device));
    const uint64_t total = (uint64_t)S->s.dim0 * S->s.num_per, chunk = 1u << 18;
    for (uint32_t t = 0; t < S->nt; t++) {
        FwdParams fp{};
        fp.dst = S->db.p + (size_t)t * S->db_words;
        fp.src_map = fp.dst_map = identity_map();
        fp.n_digits = 1;
        fp.seed = seed;
        fp.p_db = S->p.p_db;
        fp.num_per = S->s.num_per;
        fp.dim0_shard = S->s.dim0;
        fp.trial = S->t0 + t;
        fp.total_n = total;
        for (uint64_t done = 0; done < total; done += chunk) {
            fp.item_base = done;
            launch_ntt_forward(S->tb, fp, LD_DBGEN1, ST_DB1, (uint32_t)std::min(chunk, total - done), S->stream);
        }
    }
    HIP_OK(hipStreamSynchronize(S->stream));
    S->have_db = true;
    return 0;
}

int spiral_gpu_pack_server_load_db(spiral_gpu_pack_server* S, uint32_t trial, const uint64_t* db) {
    if (!S || !db) return fail("null argument");
    HIP_OK(hipSetDevice(S->device));
    if (trial < S->t0 || trial >= S->t0 + S->nt) return fail("trial %u is not one of this server's [%u, %u)", trial, S->t0, S->t0 + S->nt);
    DevBuf st;
    const size_t ref_words = (size_t)kN * S->s.dim0 * S->s.num_per;
    if (st.alloc(ref_words)) return -1;
    hipError_t e = hipMemcpy(st.p, db, ref_words * sizeof(uint64_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        launch_db1_relayout(st.p, S->db.p + (size_t)(trial - S->t0) * S->db_words, S->s.num_per, S->s.dim0, S->stream);
        e = hipStreamSynchronize(S->stream);
    }
    st.release();
    if (e != hipSuccess) return fail("database upload failed: %s", hipGetErrorString(e));
    S->have_db = true;
    return 0;
}

// raw ingest of one trial's 1 x 1 plaintexts (src/testing.cpp:845-869 + convertDb :316-340 on the device)
int spiral_gpu_pack_server_load_db_items(spiral_gpu_pack_server* S, uint32_t trial, const void* items, uint32_t coeff_bits, uint64_t first_item,
                                         uint64_t n_items) {
    if (!S) return fail("null server");
    HIP_OK(hipSetDevice(S->device));
    if (trial < S->t0 || trial >= S->t0 + S->nt) return fail("trial %u is not one of this server's [%u, %u)", trial, S->t0, S->t0 + S->nt);
    const uint64_t total = (uint64_t)S->s.dim0 * S->s.num_per;
    if (first_item > total || n_items > total - first_item) return fail("items outside the database");
    FwdParams fp{};
    fp.dst = S->db.p + (size_t)(trial - S->t0) * S->db_words;
    fp.src_map = fp.dst_map = identity_map();
    fp.n_digits = 1;
    fp.p_db = S->p.p_db;
    fp.num_per = S->s.num_per;
    fp.dim0_shard = S->s.dim0;
    fp.trial = trial;
    fp.total_n = total;
    fp.coeff_bits = coeff_bits;
    if (ingest_items(items, coeff_bits, first_item, first_item, first_item + n_items, 1, S->p.p_db, S->stream,
                     [&](const uint8_t* d_items, uint32_t* d_err, uint64_t first, uint64_t n) {
                         fp.items = d_items;
                         fp.err = d_err;
                         fp.items_first = fp.item_base = first;
                         launch_ntt_forward(S->tb, fp, LD_DBGEN1, ST_DB1, (uint32_t)n, S->stream);
                     }))
        return -1;
    S->have_db = true;
    return 0;
}

int spiral_gpu_pack_server_fill_db_random(spiral_gpu_pack_server* S, uint64_t seed) {
    if (!S) return fail("null server");
    HIP_OK(hipSetDevice(S->device));
    for (uint32_t t = 0; t < S->nt; t++) launch_fill_db1_random(S->db.p + (size_t)t * S->db_words, S->s.num_per, S->s.dim0, seed + S->t0 + t, S->stream);
    HIP_OK(hipStreamSynchronize(S->stream));
    S->have_db = true;
    return 0;
}

int spiral_gpu_pack_server_set_pub_params(spiral_gpu_pack_server* S, const uint64_t* w_left, const uint64_t* w_right, const uint64_t* v,
                                          const uint64_t* v_w) {
    if (!S) return fail("null server");
    HIP_OK(hipSetDevice(S->device));
    const spiral_gpu_params& p = S->p;
    if (!p.direct_upload) {
        if (pk_upload_ref_ntt(S, w_left, S->w_left.p, (size_t)S->s.n_left * 2 * p.t_exp)) return -1;
        if (pk_upload_ref_ntt(S, w_right, S->w_right.p, (size_t)S->s.n_right * 2 * p.t_exp_right)) return -1;
        if (pk_upload_ref_ntt(S, v, S->v.p, (size_t)2 * 2 * p.t_conv)) return -1;
    }
    if (pk_upload_ref_ntt(S, v_w, S->v_w.p, (size_t)S->out_n * (S->out_n + 1) * p.t_conv)) return -1;
    S->have_pp = true;
    return 0;
}

// everything up to and including the folding, for this server's trials: their folded ciphertexts end up at the head of each trial's
// num_per slots of S->raw (events 0..5 bracket the stages)
static int pk_front(spiral_gpu_pack_server* S, const uint64_t* query) {
    const spiral_gpu_params& p = S->p;
    const spiral_gpu_pack_shape& s = S->s;
    hipStream_t st = S->stream;
    const uint32_t ell = s.ell, ngs = p.nu2 * ell, nt = S->nt;
    if (pk_upload_ref_ntt(S, query, S->query.p, (size_t)s.n_query_cts * 2)) return -1;

    HIP_OK(hipEventRecord(S->ev[0], st));
    // ---- coefficientExpansion + reorientCiphertextsDim1 (src/testing.cpp:1009-1020)
    if (!p.direct_upload) {
        ExpandWork wk{S->ex_raw.p, S->ex_g.p};
        run_expand(S->tb, S->cv.p, s.g, p.t_exp, S->w_left.p, p.t_exp_right, S->w_right.p, ell * p.nu2, s.stopround, wk, st,
                   s.g ? S->query.p : nullptr);
        if (s.g == 0) HIP_OK(hipMemcpyAsync(S->cv.p, S->query.p, 2 * kPolyBytes, hipMemcpyDeviceToDevice, st));
        launch_qs1_from_cv(S->cv.p, (uint32_t*)S->qs1.p, s.dim0, 2, st);
    } else {
        launch_qs1_from_cv(S->query.p, (uint32_t*)S->qs1.p, s.dim0, 1, st);
    }
    HIP_OK(hipEventRecord(S->ev[1], st));
    // ---- regevToSimpleGsw + the negated GSW ciphertexts (:1022-1033)
    if (!p.direct_upload) {
        InvParams ip{};
        ip.src = S->cv.p;
        ip.dst = S->gs_raw.p;
        ip.src_map = IndexMap{2, 4, 2};  // both rows of ct 2*ij + 1
        ip.dst_map = identity_map();
        launch_ntt_inverse(S->tb, ip, IST_CRT, 2 * ngs, st);
        FwdParams fp{};
        fp.src = S->gs_raw.p;
        fp.dst = S->gs_chat.p;
        fp.src_map = fp.dst_map = identity_map();
        fp.n_digits = p.t_conv;
        fp.bits = get_bits_per(p.t_conv);
        fp.pmode = PM_GSW;
        launch_ntt_forward(S->tb, fp, LD_PDIGIT, ST_PK, 2 * ngs * p.t_conv, st);
        MatmulParams mp{S->v.p, S->gs_chat.p, S->gs_tmp.p, 2, 2 * p.t_conv, 1, 0, 2 * p.t_conv, 2};
        launch_matmul(mp, ngs, st);
        launch_pack_gsw_assemble(S->gs_tmp.p, S->cv.p, S->gsw.p, ell, p.nu2, st);
    } else {
        launch_pack_gsw_from_upload(S->query.p, S->gsw.p, s.dim0, ell, p.nu2, st);
    }
    launch_pack_fold_key(S->gsw.p, S->key.p, ell, p.nu2, st);
    HIP_OK(hipEventRecord(S->ev[2], st));
    // ---- first dimension for every trial (:1049-1051), then one INTT + CRT lift (:1055-1057)
    launch_sweep1(S->db.p, (const uint32_t*)S->qs1.p, S->acc.p, s.num_per, s.dim0, nt, S->db_words, (size_t)s.num_per * 2 * kN, st);
    HIP_OK(hipEventRecord(S->ev[3], st));
    HIP_OK(hipEventRecord(S->ev[4], st));  // (the lift is chained into the first fold round's digit transforms)
    // ---- foldCiphertextsDim1 (:596-624), all trials batched: each round = fold_chain_kernel (lift of the previous
    // product or of the accumulators + unsigned digits + forward transforms) and one product; a last lift to raw
    uint32_t np = s.num_per;
    const uint64_t* src = S->acc.p;
    uint32_t src_stride = s.num_per;
    // Pair form (option fold_pair = 0 keeps the reference's two products): folding_neg = gadget - F (:1027-1032), so
    // F_neg G^-1(L) + F G^-1(H) = L + F (G^-1(H) - G^-1(L)) -- the unsigned digits always recompose their value -- i.e. per round one lift
    // of the 2 np ciphertexts, ell digit-difference transforms per polynomial pair (LD_PDIFF) and a product of K = 2 ell terms + L.
    const bool pair = options().fold_pair != 0;  // (read per call: tests switch it inside one process)
    uint64_t* out = S->fold_c.p;
    for (uint32_t cur = 0; cur < p.nu2; cur++) {
        np /= 2;
        if (src == out) out = out == S->fold_c.p ? S->fold_c2.p : S->fold_c.p;
        if (pair) {
            InvParams ip{};
            ip.src = src;
            ip.dst = S->raw.p;  // [t][2 np][2], compact
            ip.src_map = IndexMap{4 * np, 2 * src_stride, 0};
            ip.dst_map = identity_map();
            launch_ntt_inverse(S->tb, ip, IST_CRT, nt * 4 * np, st);
            FwdParams fp{};
            fp.src = S->raw.p;
            fp.dst = S->fold_d.p;
            fp.src_map = fp.dst_map = identity_map();
            fp.n_digits = ell;
            fp.bits = get_bits_per(ell);
            fp.fold_np = np;
            fp.lazy_out = lazy_ok(2 * ell + 1) ? 1 : 0;  // pack_fold_mac sums 2 ell products and the addend per accumulator
            launch_ntt_forward(S->tb, fp, LD_PDIFF, ST_PK, nt * np * 2 * ell, st);
            launch_pack_fold_mac(S->key.p + ((size_t)cur * 2 * 4 * ell + 2 * ell) * kN, S->fold_d.p, out, 2 * ell, nt * np, st, 4 * ell, src, np, src_stride);
            src = out;
            src_stride = np;
            continue;
        }
        FoldChainParams cp{};
        cp.src = src;
        cp.dst = S->fold_d.p;
        cp.ell = ell;
        cp.bits = get_bits_per(ell);
        cp.fold_np = np;
        cp.pack = 1;
        cp.src_stride = src_stride;
        const uint32_t n_src = nt * 2 * np * 2;
        uint32_t dpb = ell;
        while (dpb > 1 && n_src * ((ell + dpb - 1) / dpb) < 768u) dpb = (dpb + 1) / 2;
        cp.dpb = dpb;
        launch_fold_chain(S->tb, cp, n_src, st);
        launch_pack_fold_mac(S->key.p + (size_t)cur * 2 * 4 * ell * kN, S->fold_d.p, out, 4 * ell, nt * np, st);
        src = out;
        src_stride = np;
    }
    {
        InvParams ip{};
        ip.src = src;
        ip.dst = S->raw.p;
        ip.src_map = p.nu2 ? identity_map() : IndexMap{2 * s.num_per, 2 * s.num_per, 0};
        ip.dst_map = IndexMap{2 * np, 2 * s.num_per, 0};  // the trial's surviving np cts at the head of its num_per slots
        launch_ntt_inverse(S->tb, ip, IST_CRT, nt * np * 2, st);
    }
    HIP_OK(hipEventRecord(S->ev[5], st));
    S->packed_after_front = false;
    return 0;
}

// pack + modulus switch (:1064-1081) of out_n^2 folded ciphertexts at a stride of `ct_stride` ciphertexts; event 6 closes it
static int pk_back(spiral_gpu_pack_server* S, const uint64_t* folded, uint32_t ct_stride) {
    const spiral_gpu_params& p = S->p;
    hipStream_t st = S->stream;
    const uint32_t rows = S->out_n + 1;
    run_pack(S->tb, folded, ct_stride, S->v_w.p, S->pk_ginv.p, S->pk_ct2.p, S->pk_res.p, S->out_n, p.t_conv, st);
    InvParams ip{};
    ip.src = S->pk_res.p;
    ip.dst = S->pk_raw.p;
    ip.src_map = ip.dst_map = identity_map();
    launch_ntt_inverse(S->tb, ip, IST_CRT, rows * S->out_n, st);
    launch_rescale(S->pk_raw.p, S->resp.p, S->out_n * kN, kQ, S->s.qprime, st);
    launch_rescale(S->pk_raw.p + (size_t)S->out_n * kN, S->resp.p + (size_t)S->out_n * kN, S->out_n * S->out_n * kN, kQ, 4 * p.p_db, st);
    HIP_OK(hipEventRecord(S->ev[6], st));
    S->packed_after_front = folded == S->raw.p;  // (a gathered buffer was filled by other servers too: no common time line)
    return 0;
}

static int pk_download(spiral_gpu_pack_server* S, uint64_t* response, uint64_t* packed_ct) {
    const uint32_t rows = S->out_n + 1;
    HIP_OK(hipStreamSynchronize(S->stream));
    HIP_OK(hipGetLastError());
    if (response) HIP_OK(hipMemcpy(response, S->resp.p, (size_t)rows * S->out_n * kPolyBytes, hipMemcpyDeviceToHost));
    if (packed_ct) {
        Scratch sc;
        if (download_pk(sc, S->pk_res.p, identity_map(), packed_ct, (size_t)rows * S->out_n)) return -1;
    }
    return 0;
}

int spiral_gpu_pack_server_answer(spiral_gpu_pack_server* S, const uint64_t* query, uint64_t* response, uint64_t* packed_ct, double stage_us[8]) {
    if (!S || !query) return fail("null argument");
    HIP_OK(hipSetDevice(S->device));
    if (!S->have_db || !S->have_pp) return fail("database and public parameters must be set first");
    if (S->nt != S->s.trials) return fail("this server holds trials [%u, %u) only: fold_trials + pack_gathered", S->t0, S->t0 + S->nt);
    if (pk_front(S, query) || pk_back(S, S->raw.p, S->s.num_per) || pk_download(S, response, packed_ct)) return -1;
    return stage_us ? spiral_gpu_pack_server_stage_us(S, stage_us) : 0;
}

// stage times of the last answer (or fold_trials [+ pack_gathered]) from the events between its stages; synchronises the stream.
// [4] packing and [6] total are 0 when no packing followed the last fold_trials on this server.
int spiral_gpu_pack_server_stage_us(spiral_gpu_pack_server* S, double stage_us[8]) {
    if (!S || !stage_us) return fail("null argument");
    HIP_OK(hipSetDevice(S->device));
    HIP_OK(hipStreamSynchronize(S->stream));
    float ms[6] = {}, total = 0;
    for (int i = 0; i < (S->packed_after_front ? 6 : 5); i++) HIP_OK(hipEventElapsedTime(&ms[i], S->ev[i], S->ev[i + 1]));
    if (S->packed_after_front) HIP_OK(hipEventElapsedTime(&total, S->ev[0], S->ev[6]));
    stage_us[0] = ms[0] * 1e3;
    stage_us[1] = ms[1] * 1e3;
    stage_us[2] = (ms[2] + ms[3]) * 1e3;
    stage_us[3] = ms[4] * 1e3;
    stage_us[4] = ms[5] * 1e3;
    stage_us[5] = ms[2] * 1e3;
    stage_us[6] = total * 1e3;
    stage_us[7] = 0;
    return 0;
}

// trial-sharded answer, part 1: expansion and conversion (replicated: they are database-independent), then the sweeps and the folding
// of this server's trials; their folded ciphertexts ([nt][2][N] raw words) are left at `folded_dev`, a device buffer of the caller
// (the send buffer of the all-gather).  Asynchronous on the server's stream.
int spiral_gpu_pack_server_fold_trials(spiral_gpu_pack_server* S, const uint64_t* query, void* folded_dev) {
    if (!S || !query || !folded_dev) return fail("null argument");
    HIP_OK(hipSetDevice(S->device));
    if (!S->have_db || !S->have_pp) return fail("database and public parameters must be set first");
    if (pk_front(S, query)) return -1;
    HIP_OK(hipMemcpy2DAsync(folded_dev, 2 * kPolyBytes, S->raw.p, (size_t)S->s.num_per * 2 * kPolyBytes, 2 * kPolyBytes, S->nt, hipMemcpyDeviceToDevice,
                            S->stream));
    return 0;
}

// part 2, on the root: pack + modulus switch of all out_n^2 folded ciphertexts (`gathered_dev`: [out_n^2][2][N] raw words in trial
// order, device memory -- the receive buffer of the all-gather)
int spiral_gpu_pack_server_pack_gathered(spiral_gpu_pack_server* S, const void* gathered_dev, uint64_t* response, uint64_t* packed_ct) {
    if (!S || !gathered_dev) return fail("null argument");
    HIP_OK(hipSetDevice(S->device));
    if (!S->have_pp) return fail("public parameters must be set first");
    if (pk_back(S, (const uint64_t*)gathered_dev, 1)) return -1;
    return pk_download(S, response, packed_ct);
}

int spiral_gpu_pack_server_set_stream(spiral_gpu_pack_server* S, void* stream) {
    if (!S) return fail("null server");
    HIP_OK(hipSetDevice(S->device));
    HIP_OK(hipStreamSynchronize(S->stream));
    if (S->own_stream) {
        (void)hipStreamDestroy(S->stream);
        S->own_stream = false;
    }
    S->stream = (hipStream_t)stream;
    return 0;
}

int spiral_gpu_pack_server_read_response_wire(spiral_gpu_pack_server* S, void* out, size_t capacity) {
    if (!S || !out) return fail("null argument");
    HIP_OK(hipSetDevice(S->device));
    const size_t nbytes = wire_bytes(&S->p, S->out_n);
    if (capacity < nbytes) return fail("response buffer of %zu bytes, the wire form needs %zu", capacity, nbytes);
    if (S->wire.words * 8 < nbytes && (S->wire.release(), S->wire.alloc(nbytes / 8))) return -1;
    launch_response_wire(S->resp.p, S->wire.p, S->out_n * kN, S->p.qprime_bits, S->out_n * S->out_n * kN, wire_bits_rest(&S->p), S->stream);
    HIP_OK(hipMemcpyAsync(out, S->wire.p, nbytes, hipMemcpyDeviceToHost, S->stream));
    HIP_OK(hipStreamSynchronize(S->stream));
    return 0;
}

int spiral_gpu_pack_server_read_acc(spiral_gpu_pack_server* S, uint32_t trial, uint64_t* out) {
    if (!S || !out) return fail("null argument");
    HIP_OK(hipSetDevice(S->device));
    if (trial < S->t0 || trial >= S->t0 + S->nt) return fail("trial %u is not one of this server's [%u, %u)", trial, S->t0, S->t0 + S->nt);
    HIP_OK(hipStreamSynchronize(S->stream));
    Scratch sc;
    return download_pk(sc, S->acc.p + (size_t)(trial - S->t0) * S->s.num_per * 2 * kN, identity_map(), out, (size_t)S->s.num_per * 2);
}

uint64_t spiral_gpu_pack_server_sweep_bytes(spiral_gpu_pack_server* S) {
    if (!S) return 0;
    // SURVEY.md 8d, pack form: 2^(nu1+nu2)*N*8 + 2^nu1*2*N*8 + 2^nu2*2*2*N*8 per trial
    const uint64_t n = kN;
    return (uint64_t)S->s.dim0 * S->s.num_per * n * 8 + (uint64_t)S->s.dim0 * 2 * n * 8 + (uint64_t)S->s.num_per * 4 * n * 8;
}

}  // extern "C"
