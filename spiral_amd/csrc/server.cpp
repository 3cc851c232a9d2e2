// Host side of libspiral_gpu.so: the C ABI of include/spiral_gpu.h and the orchestration of the server
// answer path (server halves of runConversionImproved / process_crtd_query / process_query_fast,
// reference src/spiral.cpp:2040-2406, 1584-1629).  Everything is kernel launches on one HIP stream;
// there is no CPU arithmetic path -- without a device every compute entry point fails.
#include "host_common.h"

using namespace spiral;

thread_local std::string spiral::host::g_err;
using namespace spiral::host;

// the process-wide options (kernels.h); the three documented environment variables give their initial values, once
spiral::Options& spiral::options() {
    static Options o = [] {
        Options v;
        if (const char* e = getenv("SPIRAL_FOLD_PAIR")) v.fold_pair = atoi(e) != 0;
        if (const char* e = getenv("SPIRAL_SWEEP_MFMA")) v.sweep_mfma_min = (uint32_t)strtoul(e, nullptr, 10);
        if (const char* e = getenv("SPIRAL_DB_STAGE_BYTES")) v.db_stage_bytes = (size_t)strtoull(e, nullptr, 10);
        return v;
    }();
    return o;
}

// =================================================================================================
// resident server
// =================================================================================================
struct spiral_gpu_server {
    spiral_gpu_params p;
    spiral_gpu_shape s;
    int device = 0;
    uint32_t j0 = 0, j1 = 0, dim0_shard = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    DeviceTables tb;
    bool keep_cts = false, have_db = false, have_pp = false, have_query = false;
    bool raw_from_acc = false;  // S->raw holds the lift of what S->acc holds now (lift ran, no sweep / write_raw / fold since): the stage fold may use the pair form
    bool have_records = false;  // the sweep's query records of the current query have been enqueued (ScalToMat ran since set_query)
    DevBuf wire;  // bit-packed response (read_response_wire)
    bool db_shared = false;  // db.p is another server's image (share_db): never written, never freed here
    // lifetime of a shared image: a lane points at its owner, the owner counts its lanes.  Destroying an owner that still has lanes
    // frees everything but the image and leaves a husk (zombie) that the last lane to go deletes -- a lane never sweeps freed memory.
    spiral_gpu_server* db_owner = nullptr;
    uint32_t n_lanes = 0;
    bool zombie = false;
    // expanded-ciphertext positions inside cv: first-dim j at j*pos_stride + pos_first, rest i at i*pos_stride + pos_rest
    uint32_t pos_stride = 1, pos_first = 0, pos_rest = 0, n_cv = 0;

    // every per-query buffer below except the lazily allocated ones (ex_raw2, ex_g2, cts_keep, stage, wire) is a piece of `arena`, carved in one
    // fixed order (srv_alloc): servers with equal parameters and shard have equal layouts, which is what run_query_batch relies on
    DevBuf arena;
    DevBuf db, w_left, w_right, w, v, query, cv, ex_raw, ex_g, ex_raw2, ex_g2;  // (the second work set: the odd tree of a split expansion)
    DevBuf cv_raw, cv_g, key, cts_keep;  // key: [d][3][m2]: the GSW matrices Q (src/spiral.cpp:2324) -- the fold key; Q_neg = G2 - Q (:2361-2379) is never stored (poly.hip fold_mac_two_kernel)
    uint64_t *gs_raw_p = nullptr, *gs_chat_p = nullptr;  // the Regev->GSW halves of cv_raw / cv_g
    DevBuf qs, acc_own, raw, fold_d, fold_c, fold_c2, resp, stage;
    uint64_t* acc = nullptr;
    hipEvent_t ev[8] = {};
    // captured stage groups (hipGraph): [0] expand + convert, [1] lift + fold + finish, [2] the same with
    // reduce_first; instantiated lazily, invalidated when a captured pointer or flag changes
    bool use_graphs = false;
    // [3] = Regev->GSW conversion on the side stream, [4] = whole query, [5] = fold_local, [6] = fold_root, [7] = run_pre + sweep
    hipGraphExec_t graph[13] = {};  // [12] = ScalToMat alone  // [8] = sharded expansion + pack, [9] = unpack + convert + sweep, [10] = ScalToMat + sweep, [11] = unpack + Regev->GSW
    const void *cap_chunk = nullptr, *cap_gathered = nullptr;
    void* cap_ct = nullptr;  // the caller's buffers captured into graphs 5 and 6
    // overlap 2 ("split"): the whole GSW side of the query -- the odd-index tree of the expansion AND the Regev->GSW conversion -- runs as its
    // own launch sequence on side_stream, beside the even tree + ScalToMat + sweep on the main stream; only the folding needs it.
    // (Modes 1 and 3 -- only the conversion forked, under the sweep -- measured slower and were removed in round 5, HISTORY.md.)
    int overlap = 0;
    bool side_pending = false;
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_batch = nullptr;  // first_dim_batch: this lane's records are ready / the shared sweep is done
    // Fold round forms.  Default: the pair form, unchained (lift launch + LD_SDIFF digit-difference launch + product with addend).
    // fold_pair = false (SPIRAL_FOLD_PAIR=0) or a gadget dimension whose digits do not recompose (!fold_pair_exact): the reference's
    // two-product form, lift chained into the digit transforms (fold_chain_kernel: a block lifts one source polynomial and transforms
    // dpb of its digits; dpb is halved from ell until the round has at least fold_blocks blocks, SPIRAL_FOLD_BLOCKS) or, with
    // SPIRAL_FOLD_CHAIN=0, as separate lift + LD_SDIGIT launches.  (The chained pair forms fold_pair_kernel / fold_team_kernel tied
    // with the unchained one and were removed in round 5; HISTORY.md has the numbers and the commit.)
    bool fold_chain = true;
    bool fold_pair = true;
    uint32_t fold_blocks = 768;
    uint32_t fold_g_log = 0;  // distributed fold over 2^fold_g_log ranks: the sweep groups its output by ii mod G
    uint32_t sweep_k_log = 0; // pipelined sweep in 2^sweep_k_log stages (set_sweep_stages): accumulators laid out [stage][rank][ct]
    ExpandShard ex_shard{};   // sharded expansion (set_expand_shard): what this rank expands itself
    const void* cap_bits_out = nullptr;  // the caller's exchange buffers captured into graphs 8 and 9
    const void* cap_bits_in = nullptr;
    // run_query_batch with this server as lane 0: the captured launch sequence and the lane set it was captured for
    hipGraphExec_t graph_batch = nullptr;
    const uint64_t* batch_key[kMaxLanes] = {};  // the lanes' arenas: every pointer the capture holds is one of them plus a fixed offset
    const uint64_t* batch_limbs = nullptr;      // the image the captured sweep reads (null: the packed one, vector-ALU passes)
    // run_query_instances with this server as the query's server: the captured launch sequence and what it was captured for
    hipGraphExec_t graph_inst = nullptr;
    std::vector<uint64_t> inst_key;
    uint32_t batch_n = 0;
    // batched sweeps of sweep_mfma_min or more queries run on the matrix cores (sweep_mfma.hip) from a second image of the database, the "limb
    // planes": built from db on first use by the image's holder (the owner of a shared image), as large as db, dropped when db is reloaded.
    // SPIRAL_SWEEP_MFMA=n sets the threshold (0 = never: at most kSweepMaxBatch queries per pass, on the vector ALU)
    // With the option one_image (default) there is no second image: the holder converts its one image to limb-plane form IN PLACE the first time a
    // batch wants it (srv_db_set_format; single queries then sweep it with sweep_mfma_kernel<1>, which ties with the vector-ALU kernel) and back
    // when something needs the packed form (a partial reload, a staged sweep).
    DevBuf db_limbs;
    bool limbs_valid = false, limbs_refused = false;  // refused: the allocation failed once, do not try again
    uint32_t sweep_mfma_min = 2;
    uint32_t db_format = SPIRAL_GPU_DB_PACKED;  // holder only: the form db is in now
    uint64_t db_epoch = 1;    // holder only: bumped when the image is reloaded or changes form -- captured sweeps of the old form must not replay
    uint64_t epoch_seen = 0;  // the holder's epoch this server's graphs were captured under
};

namespace {

void lane_attach(spiral_gpu_server* lane, spiral_gpu_server* owner);
void lane_detach(spiral_gpu_server* lane);

// digits per workgroup of a fold round with n_src source polynomials: as few workgroups as keep the chip busy (every
// workgroup repeats the inverse transform once), halved from ell until the round has about fold_blocks of them
uint32_t fold_dpb(const spiral_gpu_server* S, uint32_t n_src) {
    uint32_t dpb = S->s.ell;
    while (dpb > 1 && n_src * ((S->s.ell + dpb - 1) / dpb) < S->fold_blocks) dpb = (dpb + 1) / 2;
    return dpb;
}

int srv_alloc(spiral_gpu_server* S, const spiral_gpu_server* db_owner) {
    const spiral_gpu_params& p = S->p;
    const spiral_gpu_shape& s = S->s;
    const size_t nic = 2 * (size_t)s.num_per;
    if (db_owner) {  // a query lane: the owner's image, never written or freed here
        S->db.p = db_owner->db.p;
        S->db.words = db_owner->db.words;
        S->db_shared = true;
        S->have_db = true;
        lane_attach(S, const_cast<spiral_gpu_server*>(db_owner));
    } else if (S->db.alloc(db_device_words((uint32_t)nic, S->dim0_shard))) {
        return -1;
    }
    S->n_cv = p.direct_upload ? s.n_bits : (1u << s.g);
    const size_t ngs = (size_t)p.nu2 * s.ell;
    const size_t half = s.num_per > 1 ? s.num_per / 2 : 1;
    auto layout = [&](Arena& a) {
        a.carve(S->w_left, (size_t)s.n_left * 2 * p.t_exp * kN);
        a.carve(S->w_right, (size_t)s.n_right * 2 * p.t_exp_right * kN);
        a.carve(S->w, (size_t)3 * 2 * p.t_conv * kN);
        a.carve(S->v, (size_t)3 * 2 * p.t_conv * kN);
        a.carve(S->query, (size_t)s.n_query_cts * 2 * kN);
        a.carve(S->cv, (size_t)S->n_cv * 2 * kN);
        if (!p.direct_upload) {
            a.carve(S->ex_raw, (size_t)S->n_cv * 2 * kN);
            a.carve(S->ex_g, expand_g_polys(s.g, p.t_exp, p.t_exp_right) * kN);
        }
        // conversion scratch: the ScalToMat sources/digits followed by the Regev->GSW ones, so both sets go through one lift
        // launch and one digit-transform launch
        a.carve(S->cv_raw, ((size_t)S->dim0_shard + ngs * 2) * kN);
        a.carve(S->cv_g, ((size_t)S->dim0_shard + ngs * 2) * p.t_conv * kN);
        a.carve(S->key, (size_t)p.nu2 * 3 * s.m2 * kN);
        a.carve(S->qs, (size_t)kN * S->dim0_shard * 6);  // 12 u32 per (z, j)
        a.carve(S->acc_own, (size_t)s.num_per * 6 * kN);
        a.carve(S->raw, (size_t)s.num_per * 6 * kN);
        a.carve(S->fold_d, half * 2 * s.m2 * 2 * kN);
        a.carve(S->fold_c, half * 6 * kN);
        a.carve(S->fold_c2, half * 6 * kN);
        a.carve(S->resp, (size_t)6 * kN);
    };
    Arena sizing;
    layout(sizing);
    if (S->arena.alloc(sizing.used)) return -1;
    Arena real;
    real.base = S->arena.p;
    layout(real);
    HIP_OK(hipMemset(S->cv.p, 0, S->cv.words * sizeof(uint64_t)));
    S->gs_raw_p = S->cv_raw.p + (size_t)S->dim0_shard * kN;
    S->gs_chat_p = S->cv_g.p + (size_t)S->dim0_shard * p.t_conv * kN;
    S->acc = S->acc_own.p;
    return 0;
}

void srv_drop_graphs(spiral_gpu_server* S) {
    for (auto& g : S->graph)
        if (g) {
            (void)hipGraphExecDestroy(g);
            g = nullptr;
        }
    if (S->graph_batch) (void)hipGraphExecDestroy(S->graph_batch);
    S->graph_batch = nullptr;
    S->batch_n = 0;
    if (S->graph_inst) (void)hipGraphExecDestroy(S->graph_inst);
    S->graph_inst = nullptr;
    S->inst_key.clear();
}

void srv_free(spiral_gpu_server* S, bool keep_db = false) {
    srv_drop_graphs(S);
    DevBuf keep, keep_limbs;
    if (keep_db) {
        keep = S->db;
        keep_limbs = S->db_limbs;
        S->db.p = S->db_limbs.p = nullptr;
    }
    DevBuf* all[] = {&S->db, &S->w_left, &S->w_right, &S->w, &S->v, &S->query, &S->cv, &S->ex_raw, &S->ex_g, &S->ex_raw2, &S->ex_g2, &S->cv_raw,
                     &S->cv_g, &S->key, &S->cts_keep, &S->qs, &S->acc_own, &S->raw, &S->fold_d, &S->fold_c, &S->fold_c2,
                     &S->resp, &S->stage, &S->wire, &S->db_limbs, &S->arena};  // (the arena after its pieces)
    if (S->db_shared) S->db.p = nullptr;
    for (DevBuf* b : all) b->release();
    for (auto& e : S->ev)
        if (e) (void)hipEventDestroy(e);
    if (S->ev_fork) (void)hipEventDestroy(S->ev_fork);
    if (S->ev_join) (void)hipEventDestroy(S->ev_join);
    if (S->ev_batch) (void)hipEventDestroy(S->ev_batch);
    if (S->side_stream) (void)hipStreamDestroy(S->side_stream);
    if (S->own_stream) (void)hipStreamDestroy(S->own_stream);
    S->side_stream = S->own_stream = nullptr;
    S->ev_fork = S->ev_join = S->ev_batch = nullptr;
    for (auto& e : S->ev) e = nullptr;
    if (keep_db) S->db = keep, S->db_limbs = keep_limbs;
}

// lane <-> owner bookkeeping of a shared database image
void lane_attach(spiral_gpu_server* lane, spiral_gpu_server* owner) {
    lane->db_owner = owner;
    owner->n_lanes++;
}
void lane_detach(spiral_gpu_server* lane) {
    spiral_gpu_server* owner = lane->db_owner;
    if (!owner) return;
    lane->db_owner = nullptr;
    if (--owner->n_lanes == 0 && owner->zombie) {  // the owner was destroyed first: its image goes with its last lane
        owner->db.release();
        owner->db_limbs.release();
        delete owner;
    }
}

spiral_gpu_server* holder_of(spiral_gpu_server* S) { return S->db_owner ? S->db_owner : S; }
const spiral_gpu_server* holder_of(const spiral_gpu_server* S) { return S->db_owner ? S->db_owner : S; }

// graphs hold a sweep kernel chosen for the image's form at capture time: drop them when the holder's image has changed since
void srv_check_epoch(spiral_gpu_server* S) {
    const uint64_t e = holder_of(S)->db_epoch;
    if (S->epoch_seen != e) {
        srv_drop_graphs(S);
        S->epoch_seen = e;
    }
}

// Converts the holder's database image between the packed form (common.h; the vector-ALU sweep) and the limb planes (sweep_mfma.hip; the
// matrix-core sweep) IN PLACE: a slot z's region is the same byte range in both forms, so the image goes through a staging buffer a few slots at a
// time -- no second image, whatever the database's size.  Offline (database load time or the first batch), never inside a capture.
int srv_db_set_format(spiral_gpu_server* H, uint32_t fmt, hipStream_t st) {
    if (H->db_format == fmt) return 0;
    const uint32_t np = H->s.num_per, jm = 2 * H->dim0_shard;
    if (!sweep_mfma_ok(np, jm)) return fail("this geometry has no limb-plane form (needs >= 64 ciphertexts per slot and a power-of-two first dimension in [64, 2048])");
    if (hipDeviceSynchronize() != hipSuccess) return fail("hipDeviceSynchronize failed");  // whatever reads or writes the image, on whichever stream
    const size_t per_z = H->db.words / kN;
    const uint32_t nzc = (uint32_t)std::max<size_t>(1, std::min<size_t>(kN, ((size_t)256 << 20) / (per_z * sizeof(uint64_t))));
    DevBuf stage;
    if (stage.alloc(per_z * nzc)) return -1;
    hipError_t e = hipSuccess;
    for (uint32_t z = 0; z < kN && e == hipSuccess; z += nzc) {
        const uint32_t nz = std::min(nzc, kN - z);
        uint64_t* region = H->db.p + (size_t)z * per_z;
        if (fmt == SPIRAL_GPU_DB_LIMBS)
            launch_db_limb_planes(region, stage.p, np, jm, st, nz);
        else
            launch_db_limb_unplanes(region, stage.p, np, jm, st, nz);
        e = hipMemcpyAsync(region, stage.p, (size_t)nz * per_z * sizeof(uint64_t), hipMemcpyDeviceToDevice, st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    stage.release();
    if (e != hipSuccess) return fail("converting the database image failed: %s", hipGetErrorString(e));
    H->db_format = fmt;
    H->db_epoch++;
    H->db_limbs.release();  // (a second image from before the option was switched on)
    H->limbs_valid = false;
    return 0;
}
// the image was (re)written in packed form by a loader
void srv_db_loaded(spiral_gpu_server* S) {
    S->have_db = true;
    S->limbs_valid = false;
    S->db_format = SPIRAL_GPU_DB_PACKED;
    S->db_epoch++;
}

// the first-dimension sweep of n queries against one database image: one pass on the matrix cores when the limb-plane image is given, else passes of
// up to kSweepMaxBatch queries on the vector ALU (wide packed geometries), else one sweep per query
int sweep_queries(const spiral_gpu_server* S, const uint64_t* limbs, const uint32_t* const* qs, uint64_t* const* acc, uint32_t n, uint32_t g_log, hipStream_t st,
                  uint32_t k_log = 0) {
    const uint32_t np = S->s.num_per, jm = 2 * S->dim0_shard;
    const spiral_gpu_server* H = holder_of(S);
    if (!limbs && H->db_format == SPIRAL_GPU_DB_LIMBS) limbs = H->db.p;  // the one image is in limb-plane form: every sweep is the matrix-core one
    if (limbs) {
        const hipError_t e = launch_sweep_mfma(limbs, qs, acc, n, np, jm, g_log, st, k_log);
        return e == hipSuccess ? 0 : fail("the matrix-core sweep could not be launched: %s", hipGetErrorString(e));
    }
    const uint32_t step = sweep_batch_ok(np, jm) ? kSweepMaxBatch : 1;
    for (uint32_t b0 = 0; b0 < n; b0 += step) {
        const uint32_t nb = n - b0 < step ? n - b0 : step;
        if (nb == 1)
            launch_sweep(S->db.p, qs[b0], acc[b0], np, jm, g_log, st, k_log);
        else
            launch_sweep_batch(S->db.p, qs + b0, acc + b0, nb, np, jm, g_log, st);
    }
    return 0;
}
// one query's sweep (all stages, or one stage of a pipelined sweep: packed image only) of the image H holds, in whichever form it is in, with S's
// query records into S's accumulators on S's stream (H = S's own holder, or another instance of the database: run_query_instances)
int sweep_with(spiral_gpu_server* S, const spiral_gpu_server* H, int stage) {
    if (H->db_format == SPIRAL_GPU_DB_LIMBS) {
        if (stage >= 0 && S->sweep_k_log)
            return fail("a staged sweep needs the packed database image, and a batch has since converted it to limb planes: call set_sweep_stages again (or set option one_image = 0)");
        const uint32_t* qs[1] = {(const uint32_t*)S->qs.p};
        uint64_t* acc[1] = {S->acc};
        const hipError_t e = launch_sweep_mfma(H->db.p, qs, acc, 1, S->s.num_per, 2 * S->dim0_shard, S->fold_g_log, S->stream, S->sweep_k_log);
        return e == hipSuccess ? 0 : fail("the matrix-core sweep could not be launched: %s", hipGetErrorString(e));
    }
    launch_sweep(H->db.p, (const uint32_t*)S->qs.p, S->acc, S->s.num_per, 2 * S->dim0_shard, S->fold_g_log, S->stream, S->sweep_k_log, stage);
    return 0;
}
int sweep_one(spiral_gpu_server* S, int stage) { return sweep_with(S, holder_of(S), stage); }

// The limb-plane image for a batched sweep of n queries on the matrix cores, or nullptr when that sweep does not apply (threshold, geometry) --
// then *rc stays 0 -- or could not be built (*rc = -1).  Built once per database load by the image's holder; never call this inside a capture.
const uint64_t* limb_image(spiral_gpu_server* S, uint32_t n, int* rc) {
    *rc = 0;
    spiral_gpu_server* H = holder_of(S);
    if (H->db_format == SPIRAL_GPU_DB_LIMBS) return H->db.p;  // (whatever the threshold says: there is no other image to sweep)
    if (S->sweep_mfma_min == 0 || n < S->sweep_mfma_min || !sweep_mfma_ok(S->s.num_per, 2 * S->dim0_shard)) return nullptr;
    if (options().one_image) {  // the one image changes form, in place
        *rc = srv_db_set_format(H, SPIRAL_GPU_DB_LIMBS, S->stream);
        return *rc ? nullptr : H->db.p;
    }
    if (H->limbs_valid) return H->db_limbs.p;
    if (H->limbs_refused) return nullptr;
    if (!H->db_limbs.p && H->db_limbs.alloc(H->db.words)) {
        // the second image does not fit beside the first (databases beyond ~120 GiB on one device): the batch sweeps in passes of two on the vector ALU
        fprintf(stderr, "spiral_gpu: no memory for the limb-plane image of the database (%zu MiB): batched sweeps stay on the vector ALU\n", (size_t)(H->db.words * 8 >> 20));
        (void)hipGetLastError();
        H->limbs_refused = true;
        return nullptr;
    }
    if (hipDeviceSynchronize() != hipSuccess) {  // whatever wrote the packed image, on whichever stream
        *rc = fail("hipDeviceSynchronize failed");
        return nullptr;
    }
    launch_db_limb_planes(H->db.p, H->db_limbs.p, H->s.num_per, 2 * H->dim0_shard, S->stream);
    if (hipStreamSynchronize(S->stream) != hipSuccess) {
        *rc = fail("building the limb-plane image failed");
        return nullptr;
    }
    H->limbs_valid = true;
    return H->db_limbs.p;
}

// the fold needs the keys the forked conversion produces
int srv_join_side(spiral_gpu_server* S) {
    if (S->side_pending) {
        HIP_OK(hipStreamWaitEvent(S->stream, S->ev_join, 0));
        S->side_pending = false;
    }
    return 0;
}

// stage a host buffer through a device staging area and convert reference NTT layout -> PK
int upload_ref_ntt(spiral_gpu_server* S, const uint64_t* host, uint64_t* pk, size_t npolys) {
    if (npolys == 0) return 0;
    if (!host) return fail("null host buffer");
    const size_t chunk = 4096;  // polynomials per staging pass (128 MiB)
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

int download_pk_as_ref(spiral_gpu_server* S, const uint64_t* pk, IndexMap map, uint64_t* host, size_t npolys) {
    if (npolys == 0) return 0;
    const size_t chunk = 4096;
    if (S->stage.words < std::min(npolys, chunk) * kRefNtt) {
        S->stage.release();
        if (S->stage.alloc(std::min(npolys, chunk) * kRefNtt)) return -1;
    }
    for (size_t done = 0; done < npolys; done += chunk) {
        const size_t n = std::min(chunk, npolys - done);
        IndexMap m = map;
        // shift the map by `done` polynomials: valid because chunk is a multiple of every inner we use
        m.off += (uint32_t)(done / map.inner) * map.outer_stride;
        launch_pk_to_ref(pk, S->stage.p, (uint32_t)n, m, S->stream);
        HIP_OK(hipMemcpyAsync(host + done * kRefNtt, S->stage.p, n * kRefNtt * sizeof(uint64_t), hipMemcpyDeviceToHost, S->stream));
        HIP_OK(hipStreamSynchronize(S->stream));
    }
    return 0;
}

}  // namespace

extern "C" {

int spiral_gpu_abi_version(void) { return SPIRAL_GPU_ABI_VERSION; }

int spiral_gpu_set_option(const char* name, int64_t value) {
    if (!name) return fail("null option name");
    Options& o = options();
    const std::string n = name;
    if (n == "fold_pair") o.fold_pair = value != 0;
    else if (n == "fold_chain") o.fold_chain = value != 0;
    else if (n == "fold_blocks" && value >= 0) o.fold_blocks = (uint32_t)value;
    else if (n == "sweep_mfma_min" && value >= 0) o.sweep_mfma_min = (uint32_t)value;
    else if (n == "one_image") o.one_image = value != 0;
    else if (n == "fwd2" && value >= -1 && value <= 1) o.fwd2 = (int)value;
    else if (n == "fwd2_min" && value >= 0) o.fwd2_min = (uint32_t)value;
    else if (n == "db_stage_bytes" && value > 0) o.db_stage_bytes = (size_t)value;
    else return fail("unknown option '%s' or value %lld out of range", name, (long long)value);
    return 0;
}
int spiral_gpu_get_option(const char* name, int64_t* value) {
    if (!name || !value) return fail("null argument");
    const Options& o = options();
    const std::string n = name;
    if (n == "fold_pair") *value = o.fold_pair;
    else if (n == "fold_chain") *value = o.fold_chain;
    else if (n == "fold_blocks") *value = o.fold_blocks;
    else if (n == "sweep_mfma_min") *value = o.sweep_mfma_min;
    else if (n == "one_image") *value = o.one_image;
    else if (n == "fwd2") *value = o.fwd2;
    else if (n == "fwd2_min") *value = o.fwd2_min;
    else if (n == "db_stage_bytes") *value = (int64_t)o.db_stage_bytes;
    else return fail("unknown option '%s'", name);
    return 0;
}
const char* spiral_gpu_last_error(void) { return g_err.c_str(); }
int spiral_gpu_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int spiral_gpu_get_shape(const spiral_gpu_params* p, spiral_gpu_shape* out) { return shape_of(p, out); }
int spiral_gpu_get_tables(uint64_t* out) {
    if (!out) return fail("null argument");
    tables_host_rows(out);
    return 0;
}

// ------------------------------------------------------------------------------------------------
// host-buffer seams
// ------------------------------------------------------------------------------------------------
int spiral_gpu_ntt_forward(uint64_t* operand, size_t npolys) {
    DeviceTables tb;
    if (current_tables(&tb)) return -1;
    Scratch sc;
    uint64_t* d = sc.upload(operand, npolys * kRefNtt);
    if (!d) return fail("device allocation/upload failed");
    FwdParams fp{};
    fp.src = d;
    fp.dst = d;
    fp.src_map = fp.dst_map = identity_map();
    fp.n_digits = 1;
    launch_ntt_forward(tb, fp, LD_LIMBS, ST_REF, (uint32_t)npolys, 0);
    HIP_OK(hipMemcpy(operand, d, npolys * kRefNtt * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return 0;
}

int spiral_gpu_ntt_inverse(uint64_t* operand, size_t npolys) {
    DeviceTables tb;
    if (current_tables(&tb)) return -1;
    Scratch sc;
    uint64_t* d = sc.upload(operand, npolys * kRefNtt);
    if (!d) return fail("device allocation/upload failed");
    InvParams ip{};
    ip.src = d;
    ip.dst = d;
    ip.src_map = ip.dst_map = identity_map();
    ip.src_ref = 1;
    ip.pre_reduce = 1;
    launch_ntt_inverse(tb, ip, IST_LIMBS, (uint32_t)npolys, 0);
    HIP_OK(hipMemcpy(operand, d, npolys * kRefNtt * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return 0;
}

int spiral_gpu_to_ntt(uint64_t* out, const uint64_t* in, size_t npolys, int reduce) {
    DeviceTables tb;
    if (current_tables(&tb)) return -1;
    Scratch sc;
    uint64_t* d_in = sc.upload(in, npolys * kN);
    uint64_t* d_out = sc.get(npolys * kRefNtt);
    if (!d_in || !d_out) return fail("device allocation/upload failed");
    FwdParams fp{};
    fp.src = d_in;
    fp.dst = d_out;
    fp.src_map = fp.dst_map = identity_map();
    fp.n_digits = 1;
    if (reduce) {
        launch_ntt_forward(tb, fp, LD_RAW, ST_REF, (uint32_t)npolys, 0);
    } else {
        // to_ntt_no_reduce copies the raw value into both limbs (src/poly.cpp:291-309): it is digit 0 of width 32
        uint64_t* d_pk = sc.get(npolys * kN);
        if (!d_pk) return fail("device allocation failed");
        fp.dst = d_pk;
        fp.bits = 32;
        launch_ntt_forward(tb, fp, LD_DIGIT, ST_PK, (uint32_t)npolys, 0);
        launch_pk_to_ref(d_pk, d_out, (uint32_t)npolys, identity_map(), 0);
    }
    HIP_OK(hipMemcpy(out, d_out, npolys * kRefNtt * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return 0;
}

int spiral_gpu_from_ntt(uint64_t* out, const uint64_t* in, size_t npolys) {
    DeviceTables tb;
    if (current_tables(&tb)) return -1;
    Scratch sc;
    uint64_t* d_in = sc.upload(in, npolys * kRefNtt);
    uint64_t* d_out = sc.get(npolys * kN);
    if (!d_in || !d_out) return fail("device allocation/upload failed");
    InvParams ip{};
    ip.src = d_in;
    ip.dst = d_out;
    ip.src_map = ip.dst_map = identity_map();
    ip.src_ref = 1;
    ip.pre_reduce = 1;
    launch_ntt_inverse(tb, ip, IST_CRT, (uint32_t)npolys, 0);
    HIP_OK(hipMemcpy(out, d_out, npolys * kN * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return 0;
}


// measurement helper: average duration of one batched forward (to_ntt: raw -> packed NTT form) and one batched inverse
// (from_ntt: packed NTT form -> CRT-lifted raw) launch over npolys polynomials, HIP events on the default stream
int spiral_gpu_time_ntt(size_t npolys, int iters, float* fwd_ms, float* inv_ms) {
    if (!fwd_ms || !inv_ms || iters <= 0 || npolys == 0) return fail("bad argument");
    DeviceTables tb;
    if (current_tables(&tb)) return -1;
    Scratch sc;
    uint64_t* d_raw = sc.get(npolys * kN);
    uint64_t* d_pk = sc.get(npolys * kN);
    if (!d_raw || !d_pk) return fail("device allocation failed");
    HIP_OK(hipMemset(d_raw, 0x5a, npolys * kN * sizeof(uint64_t)));
    hipEvent_t e[3];
    for (auto& x : e) HIP_OK(hipEventCreate(&x));
    FwdParams fp{};
    fp.src = d_raw;
    fp.dst = d_pk;
    fp.src_map = fp.dst_map = identity_map();
    fp.n_digits = 1;
    InvParams ip{};
    ip.src = d_pk;
    ip.dst = d_raw;
    ip.src_map = ip.dst_map = identity_map();
    launch_ntt_forward(tb, fp, LD_RAW, ST_PK, (uint32_t)npolys, 0);  // warm
    launch_ntt_inverse(tb, ip, IST_CRT, (uint32_t)npolys, 0);
    HIP_OK(hipEventRecord(e[0], 0));
    for (int i = 0; i < iters; i++) launch_ntt_forward(tb, fp, LD_RAW, ST_PK, (uint32_t)npolys, 0);
    HIP_OK(hipEventRecord(e[1], 0));
    for (int i = 0; i < iters; i++) launch_ntt_inverse(tb, ip, IST_CRT, (uint32_t)npolys, 0);
    HIP_OK(hipEventRecord(e[2], 0));
    HIP_OK(hipEventSynchronize(e[2]));
    HIP_OK(hipEventElapsedTime(fwd_ms, e[0], e[1]));
    HIP_OK(hipEventElapsedTime(inv_ms, e[1], e[2]));
    *fwd_ms /= iters;
    *inv_ms /= iters;
    for (auto& x : e) (void)hipEventDestroy(x);
    return 0;
}

// the gadget-digit transform launch the conversion / expansion / folding stages are made of: n_digits unsigned digits of each of
// npolys raw polynomials (gadget_invert + to_ntt_no_reduce), one workgroup per digit polynomial -- the source polynomial is read
// n_digits times (cache hits after the first), every transform writes its 16 KiB
int spiral_gpu_time_ntt_digits(size_t npolys, uint32_t n_digits, int iters, float* ms) {
    if (!ms || iters <= 0 || npolys == 0 || n_digits < 1 || n_digits > 56) return fail("bad argument");
    DeviceTables tb;
    if (current_tables(&tb)) return -1;
    Scratch sc;
    uint64_t* d_raw = sc.get(npolys * kN);
    uint64_t* d_pk = sc.get(npolys * n_digits * kN);
    if (!d_raw || !d_pk) return fail("device allocation failed");
    HIP_OK(hipMemset(d_raw, 0x5a, npolys * kN * sizeof(uint64_t)));
    hipEvent_t e[2];
    for (auto& x : e) HIP_OK(hipEventCreate(&x));
    FwdParams fp{};
    fp.src = d_raw;
    fp.dst = d_pk;
    fp.src_map = fp.dst_map = identity_map();
    fp.n_digits = n_digits;
    fp.bits = get_bits_per(n_digits);
    launch_ntt_forward(tb, fp, LD_DIGIT, ST_PK, (uint32_t)(npolys * n_digits), 0);  // warm
    HIP_OK(hipEventRecord(e[0], 0));
    for (int i = 0; i < iters; i++) launch_ntt_forward(tb, fp, LD_DIGIT, ST_PK, (uint32_t)(npolys * n_digits), 0);
    HIP_OK(hipEventRecord(e[1], 0));
    HIP_OK(hipEventSynchronize(e[1]));
    HIP_OK(hipEventElapsedTime(ms, e[0], e[1]));
    *ms /= iters;
    for (auto& x : e) (void)hipEventDestroy(x);
    return 0;
}

int spiral_gpu_multiply(uint64_t* out, const uint64_t* a, const uint64_t* b, size_t rs, size_t ms, size_t cs) {
    Scratch sc;
    uint64_t* da = upload_pk(sc, a, rs * ms);
    uint64_t* db = upload_pk(sc, b, ms * cs);
    uint64_t* dout = sc.get(rs * cs * kN);
    if (!da || !db || !dout) return fail("device allocation/upload failed");
    MatmulParams mp{da, db, dout, (uint32_t)rs, (uint32_t)ms, (uint32_t)cs, 0, 0, 0};
    launch_matmul(mp, 1, 0);
    return download_pk(sc, dout, identity_map(), out, rs * cs);
}

int spiral_gpu_add(uint64_t* out, const uint64_t* a, const uint64_t* b, size_t npolys) {
    Scratch sc;
    uint64_t* da = upload_pk(sc, a, npolys);
    uint64_t* db = upload_pk(sc, b, npolys);
    if (!da || !db) return fail("device allocation/upload failed");
    launch_add(da, db, da, (uint32_t)npolys, 0);
    return download_pk(sc, da, identity_map(), out, npolys);
}

int spiral_gpu_mul_by_const(uint64_t* out, const uint64_t* single_poly, const uint64_t* a, size_t npolys) {
    Scratch sc;
    uint64_t* ds = upload_pk(sc, single_poly, 1);
    uint64_t* da = upload_pk(sc, a, npolys);
    if (!ds || !da) return fail("device allocation/upload failed");
    launch_mul_by_const(ds, da, da, (uint32_t)npolys, 0);
    return download_pk(sc, da, identity_map(), out, npolys);
}

int spiral_gpu_automorph(uint64_t* out, const uint64_t* in, size_t npolys, uint64_t t) {
    if ((t & 1) == 0) return fail("automorphism exponent must be odd");
    Scratch sc;
    uint64_t* di = sc.upload(in, npolys * kN);
    uint64_t* dout = sc.get(npolys * kN);
    if (!di || !dout) return fail("device allocation/upload failed");
    launch_automorph(di, dout, (uint32_t)npolys, (uint32_t)t, 0);
    HIP_OK(hipMemcpy(out, dout, npolys * kN * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return 0;
}

int spiral_gpu_invert(uint64_t* out, const uint64_t* in, size_t npolys) {
    Scratch sc;
    uint64_t* di = sc.upload(in, npolys * kN);
    if (!di) return fail("device allocation/upload failed");
    launch_invert(di, di, (uint32_t)npolys, 0);
    HIP_OK(hipMemcpy(out, di, npolys * kN * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return 0;
}

int spiral_gpu_gadget_invert(uint64_t* out, const uint64_t* in, size_t mx, size_t rdim, size_t cols) {
    if (rdim == 0 || mx % rdim) return fail("mx must be a multiple of rdim");
    Scratch sc;
    uint64_t* di = sc.upload(in, rdim * cols * kN);
    uint64_t* dout = sc.get(mx * cols * kN);
    if (!di || !dout) return fail("device allocation/upload failed");
    launch_gadget_invert(di, dout, (uint32_t)mx, (uint32_t)rdim, (uint32_t)cols, 0);
    HIP_OK(hipMemcpy(out, dout, mx * cols * kN * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return 0;
}

int spiral_gpu_get_rescaled(uint64_t* out, const uint64_t* in, size_t n, uint64_t inp_mod, uint64_t out_mod) {
    Scratch sc;
    uint64_t* di = sc.upload(in, n);
    if (!di) return fail("device allocation/upload failed");
    launch_rescale(di, di, (uint32_t)n, inp_mod, out_mod, 0);
    HIP_OK(hipMemcpy(out, di, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return 0;
}

int spiral_gpu_multiply_query_by_database(uint64_t* output, const uint64_t* reorientedCiphertexts, const uint64_t* database, size_t dim0,
                                          size_t num_per) {
    if (dim0 == 0 || num_per == 0) return fail("empty geometry");
    Scratch sc;
    const size_t db_words = (size_t)kN * dim0 * num_per * 4;
    uint64_t* d_ref = sc.upload(database, db_words);
    uint64_t* d_db = sc.get(db_device_words((uint32_t)(2 * num_per), (uint32_t)dim0));
    uint64_t* d_re = sc.upload(reorientedCiphertexts, (size_t)kN * dim0 * 8);
    uint64_t* d_qs = sc.get((size_t)kN * dim0 * 6);
    uint64_t* d_acc = sc.get(num_per * 6 * kN);
    if (!d_ref || !d_db || !d_re || !d_qs || !d_acc) return fail("device allocation/upload failed");
    launch_db_relayout(d_ref, d_db, (uint32_t)num_per, (uint32_t)dim0, 0, (uint32_t)dim0, 0, kN, 0);
    launch_qs_from_reoriented(d_re, (uint32_t*)d_qs, (uint32_t)(2 * dim0), 0);
    launch_sweep(d_db, (const uint32_t*)d_qs, d_acc, (uint32_t)num_per, (uint32_t)(2 * dim0), 0, 0);
    return download_pk(sc, d_acc, identity_map(), output, num_per * 6);
}

int spiral_gpu_multiply_queries_by_database(uint64_t* outputs, const uint64_t* reorientedCiphertexts, size_t n, const uint64_t* database, size_t dim0,
                                            size_t num_per) {
    if (dim0 == 0 || num_per == 0 || n == 0) return fail("empty geometry");
    if (n > kMaxLanes) return fail("at most %u queries per pass", kMaxLanes);
    Scratch sc;
    const size_t db_words = (size_t)kN * dim0 * num_per * 4, dev_words = db_device_words((uint32_t)(2 * num_per), (uint32_t)dim0);
    const bool mfma = sweep_mfma_ok((uint32_t)num_per, (uint32_t)(2 * dim0));
    uint64_t* d_ref = sc.upload(database, db_words);
    uint64_t* d_db = sc.get(dev_words);
    uint64_t* d_limbs = mfma ? sc.get(dev_words) : nullptr;
    uint64_t* d_re = sc.upload(reorientedCiphertexts, n * (size_t)kN * dim0 * 8);
    uint64_t* d_qs = sc.get(n * (size_t)kN * dim0 * 6);
    uint64_t* d_acc = sc.get(n * num_per * 6 * kN);
    if (!d_ref || !d_db || (mfma && !d_limbs) || !d_re || !d_qs || !d_acc) return fail("device allocation/upload failed");
    launch_db_relayout(d_ref, d_db, (uint32_t)num_per, (uint32_t)dim0, 0, (uint32_t)dim0, 0, kN, 0);
    if (mfma) launch_db_limb_planes(d_db, d_limbs, (uint32_t)num_per, (uint32_t)(2 * dim0), 0);
    const uint32_t* qs[kMaxLanes];
    uint64_t* acc[kMaxLanes];
    for (size_t b = 0; b < n; b++) {
        qs[b] = (const uint32_t*)(d_qs + b * (size_t)kN * dim0 * 6);
        acc[b] = d_acc + b * num_per * 6 * kN;
        launch_qs_from_reoriented(d_re + b * (size_t)kN * dim0 * 8, (uint32_t*)qs[b], (uint32_t)(2 * dim0), 0);
    }
    if (mfma) {
        const hipError_t e = launch_sweep_mfma(d_limbs, qs, acc, (uint32_t)n, (uint32_t)num_per, (uint32_t)(2 * dim0), 0, 0);
        if (e != hipSuccess) return fail("the matrix-core sweep could not be launched: %s", hipGetErrorString(e));
    } else
        for (size_t b0 = 0; b0 < n; b0 += 2) {
            if (n - b0 >= 2 && sweep_batch_ok((uint32_t)num_per, (uint32_t)(2 * dim0)))
                launch_sweep_batch(d_db, qs + b0, acc + b0, 2, (uint32_t)num_per, (uint32_t)(2 * dim0), 0, 0);
            else
                for (size_t b = b0; b < n && b < b0 + 2; b++) launch_sweep(d_db, qs[b], acc[b], (uint32_t)num_per, (uint32_t)(2 * dim0), 0, 0);
        }
    return download_pk(sc, d_acc, identity_map(), outputs, n * num_per * 6);
}

int spiral_gpu_split_and_crt(uint64_t* out, const uint64_t* in, size_t num_per, uint32_t t_gsw) {
    DeviceTables tb;
    if (current_tables(&tb)) return -1;
    Scratch sc;
    const uint32_t m2 = 3 * t_gsw;
    uint64_t* di = sc.upload(in, num_per * 6 * kN);
    uint64_t* dd = sc.get(num_per * 2 * m2 * 2 * kN);  // fold operand layout, only the low halves are filled
    if (!di || !dd) return fail("device allocation/upload failed");
    FwdParams fp{};
    fp.src = di;
    fp.dst = dd;
    fp.src_map = identity_map();
    fp.n_digits = t_gsw;
    fp.bits = get_bits_per(t_gsw);
    fp.ell = t_gsw;
    fp.fold_np = (uint32_t)num_per;  // every ct index < num_per -> low half
    launch_ntt_forward(tb, fp, LD_SDIGIT, ST_PK, (uint32_t)(num_per * 6 * t_gsw), 0);
    // D[i][row][c] at (i*2*m2 + row)*2 + c  ->  reference [i][row][c]
    return download_pk(sc, dd, IndexMap{2 * m2, 4 * m2, 0}, out, num_per * m2 * 2);
}

int spiral_gpu_fold_one_further_dimension(uint64_t* cts, size_t num_per, const uint64_t* query_ct, const uint64_t* query_ct_neg,
                                          uint32_t t_gsw) {
    DeviceTables tb;
    if (current_tables(&tb)) return -1;
    Scratch sc;
    const uint32_t m2 = 3 * t_gsw;
    uint64_t* d_cts = sc.upload(cts, 2 * num_per * 6 * kN);
    uint64_t* d_q = sc.upload(query_ct, (size_t)kN * 3 * m2);
    uint64_t* d_qn = sc.upload(query_ct_neg, (size_t)kN * 3 * m2);
    uint64_t* d_key = sc.get((size_t)3 * 2 * m2 * kN);
    uint64_t* d_d = sc.get(num_per * 2 * m2 * 2 * kN);
    uint64_t* d_c = sc.get(num_per * 6 * kN);
    if (!d_cts || !d_q || !d_qn || !d_key || !d_d || !d_c) return fail("device allocation/upload failed");
    launch_fold_key_from_reoriented(d_q, d_qn, d_key, m2, 0);
    FwdParams fp{};
    fp.src = d_cts;
    fp.dst = d_d;
    fp.src_map = identity_map();
    fp.n_digits = t_gsw;
    fp.bits = get_bits_per(t_gsw);
    fp.ell = t_gsw;
    fp.fold_np = (uint32_t)num_per;
    launch_ntt_forward(tb, fp, LD_SDIGIT, ST_PK, (uint32_t)(2 * num_per * 6 * t_gsw), 0);
    launch_fold_mac(d_key, d_d, d_c, 2 * m2, (uint32_t)num_per, 0);
    InvParams ip{};
    ip.src = d_c;
    ip.dst = d_cts;
    ip.src_map = ip.dst_map = identity_map();
    launch_ntt_inverse(tb, ip, IST_CRT, (uint32_t)(num_per * 6), 0);
    HIP_OK(hipMemcpy(cts, d_cts, num_per * 6 * kN * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return 0;
}

int spiral_gpu_expand_improved(uint64_t* cv_v, uint32_t g, uint32_t t_exp, const uint64_t* w_left, uint32_t t_exp_right,
                               const uint64_t* w_right, uint32_t n_right, uint32_t max_bits_to_gen_right, uint32_t stopround) {
    DeviceTables tb;
    if (current_tables(&tb)) return -1;
    if (g == 0 || g > kLogN) return fail("g out of range");
    const uint32_t need_right = stopround ? stopround + 1 : g;
    if (n_right < need_right) return fail("W_exp_right has %u matrices, %u needed", n_right, need_right);
    Scratch sc;
    const size_t ncv = (size_t)1 << g;
    uint64_t* d_cv = upload_pk(sc, cv_v, ncv * 2);
    uint64_t* d_wl = upload_pk(sc, w_left, (size_t)g * 2 * t_exp);
    uint64_t* d_wr = upload_pk(sc, w_right, (size_t)n_right * 2 * t_exp_right);
    ExpandWork wk{sc.get(ncv * 2 * kN), sc.get(expand_g_polys(g, t_exp, t_exp_right) * kN)};
    if (!d_cv || !d_wl || !d_wr || !wk.raw || !wk.g) return fail("device allocation/upload failed");
    run_expand(tb, d_cv, g, t_exp, d_wl, t_exp_right, d_wr, max_bits_to_gen_right, stopround, wk, 0);
    return download_pk(sc, d_cv, identity_map(), cv_v, ncv * 2);
}

int spiral_gpu_scal_to_mat(uint64_t* out, const uint64_t* cv, const uint64_t* w, uint32_t t_conv) {
    DeviceTables tb;
    if (current_tables(&tb)) return -1;
    Scratch sc;
    uint64_t* d_cv = upload_pk(sc, cv, 2);
    uint64_t* d_w = upload_pk(sc, w, (size_t)3 * 2 * t_conv);
    uint64_t* d_raw = sc.get(kN);
    uint64_t* d_g = sc.get((size_t)t_conv * kN);
    uint64_t* d_out = sc.get((size_t)6 * kN);
    if (!d_cv || !d_w || !d_raw || !d_g || !d_out) return fail("device allocation/upload failed");
    InvParams ip{};
    ip.src = d_cv;
    ip.dst = d_raw;
    ip.src_map = ip.dst_map = identity_map();
    launch_ntt_inverse(tb, ip, IST_CRT, 1, 0);
    FwdParams fp{};
    fp.src = d_raw;
    fp.dst = d_g;
    fp.src_map = fp.dst_map = identity_map();
    fp.n_digits = t_conv;
    fp.bits = get_bits_per(t_conv);
    launch_ntt_forward(tb, fp, LD_DIGIT, ST_PK, t_conv, 0);
    Scal2MatParams sp{};
    sp.w = d_w;
    sp.g = d_g;
    sp.cv = d_cv;
    sp.cv_pos = identity_map();
    sp.out = d_out;
    sp.t_conv = t_conv;
    sp.count = 1;
    launch_scal2mat(sp, 0);
    return download_pk(sc, d_out, identity_map(), out, 6);
}

int spiral_gpu_regev_to_gsw(uint64_t* out, const uint64_t* cv_v, const uint64_t* w, const uint64_t* v, uint32_t t_conv, uint32_t ell) {
    DeviceTables tb;
    if (current_tables(&tb)) return -1;
    Scratch sc;
    uint64_t* d_cv = upload_pk(sc, cv_v, (size_t)ell * 2);
    uint64_t* d_w = upload_pk(sc, w, (size_t)3 * 2 * t_conv);
    uint64_t* d_v = upload_pk(sc, v, (size_t)3 * 2 * t_conv);
    uint64_t* d_raw = sc.get((size_t)ell * 2 * kN);
    uint64_t* d_chat = sc.get((size_t)ell * 2 * t_conv * kN);
    uint64_t* d_gsw = sc.get((size_t)3 * 3 * ell * kN);
    if (!d_cv || !d_w || !d_v || !d_raw || !d_chat || !d_gsw) return fail("device allocation/upload failed");
    InvParams ip{};
    ip.src = d_cv;
    ip.dst = d_raw;
    ip.src_map = ip.dst_map = identity_map();
    launch_ntt_inverse(tb, ip, IST_CRT, 2 * ell, 0);
    FwdParams fp{};
    fp.src = d_raw;
    fp.dst = d_chat;
    fp.src_map = fp.dst_map = identity_map();
    fp.n_digits = t_conv;
    fp.bits = get_bits_per(t_conv);
    launch_ntt_forward(tb, fp, LD_DIGIT, ST_PK, 2 * ell * t_conv, 0);
    GswParams gp{};
    gp.w = d_w;
    gp.v = d_v;
    gp.chat = d_chat;
    gp.cv = d_cv;
    gp.cv_pos = identity_map();
    gp.gsw = d_gsw;
    gp.t_conv = t_conv;
    gp.ell = ell;
    gp.dims = 1;
    launch_regev_to_gsw(gp, 0);
    return download_pk(sc, d_gsw, identity_map(), out, (size_t)9 * ell);
}

// ------------------------------------------------------------------------------------------------
// resident server
// ------------------------------------------------------------------------------------------------
static int srv_create(const spiral_gpu_params* p, int device, uint32_t j_begin, uint32_t j_end, const spiral_gpu_server* db_owner, spiral_gpu_server** out) {
    if (!p || !out) return fail("null argument");
    spiral_gpu_shape s;
    if (shape_of(p, &s)) return -1;
    if (j_end == 0 && j_begin == 0) j_end = s.dim0;
    if (j_begin >= j_end || j_end > s.dim0) return fail("bad first-dimension shard [%u, %u) of %u", j_begin, j_end, s.dim0);
    HIP_OK(hipSetDevice(device));
    spiral_gpu_server* S = new spiral_gpu_server();
    S->p = *p;
    S->s = s;
    S->device = device;
    S->j0 = j_begin;
    S->j1 = j_end;
    S->dim0_shard = j_end - j_begin;
    {  // the process-wide options as they stand now (spiral_gpu_set_option; SPIRAL_FOLD_PAIR / SPIRAL_SWEEP_MFMA give their initial values)
        const Options& o = options();
        S->fold_chain = o.fold_chain != 0;
        S->fold_pair = o.fold_pair != 0;
        S->fold_blocks = o.fold_blocks;
        S->sweep_mfma_min = o.sweep_mfma_min;
    }
    if (p->direct_upload || s.stopround == 0) {
        S->pos_stride = 1;
        S->pos_first = 0;
        S->pos_rest = s.dim0;
    } else {  // reorderFromStopround (src/spiral.cpp:2027-2036): even slots, then odd slots
        S->pos_stride = 2;
        S->pos_first = 0;
        S->pos_rest = 1;
    }
    if (tables_get(device, &S->tb) != 0) {
        delete S;
        return fail("twiddle table setup failed on device %d", device);
    }
    if (hipStreamCreate(&S->own_stream) != hipSuccess) {
        delete S;
        return fail("hipStreamCreate failed");
    }
    S->stream = S->own_stream;
    if (hipStreamCreateWithFlags(&S->side_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&S->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&S->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&S->ev_batch, hipEventDisableTiming) != hipSuccess) {
        srv_free(S);
        delete S;
        return fail("side stream setup failed");
    }
    for (auto& e : S->ev)
        if (hipEventCreate(&e) != hipSuccess) {
            srv_free(S);
            delete S;
            return fail("hipEventCreate failed");
        }
    if (srv_alloc(S, db_owner)) {
        lane_detach(S);
        srv_free(S);
        delete S;
        return -1;
    }
    *out = S;
    return 0;
}

int spiral_gpu_server_create(const spiral_gpu_params* p, int device, uint32_t j_begin, uint32_t j_end, spiral_gpu_server** out) {
    return srv_create(p, device, j_begin, j_end, nullptr, out);
}

int spiral_gpu_server_create_lane(spiral_gpu_server* owner, spiral_gpu_server** out) {
    if (!owner || !out) return fail("null argument");
    if (owner->db_shared || owner->zombie) return fail("the owner does not own its database image");
    if (!owner->have_db) return fail("the owner has no database loaded");
    return srv_create(&owner->p, owner->device, owner->j0, owner->j1, owner, out);
}

void spiral_gpu_server_destroy(spiral_gpu_server* S) {
    if (!S || S->zombie) return;
    (void)hipSetDevice(S->device);
    (void)hipDeviceSynchronize();
    if (S->n_lanes > 0) {  // lanes still sweep this server's image: keep the image (only), the last lane frees it
        srv_free(S, true);
        S->zombie = true;
        S->have_pp = S->have_query = false;
        return;
    }
    lane_detach(S);
    srv_free(S);
    delete S;
}

int spiral_gpu_server_set_stream(spiral_gpu_server* S, void* hip_stream) {
    if (!S) return fail("null server");
    S->stream = hip_stream ? (hipStream_t)hip_stream : S->own_stream;
    return 0;
}

void* spiral_gpu_server_get_stream(spiral_gpu_server* S) { return S ? (void*)S->stream : nullptr; }

int spiral_gpu_server_use_graphs(spiral_gpu_server* S, int on) {
    if (!S) return fail("null server");
    S->use_graphs = on != 0;
    if (!on) srv_drop_graphs(S);
    return 0;
}

int spiral_gpu_server_load_db(spiral_gpu_server* S, const uint64_t* database) {
    if (!S || !database) return fail("null argument");
    if (S->db_shared) return fail("this server sweeps another server's database image (share_db): load it through the owner");
    HIP_OK(hipSetDevice(S->device));
    // stage the reference-layout database one z-slab group at a time and re-lay the shard
    const size_t per_z_ref = (size_t)S->s.num_per * 2 * S->s.dim0 * 2;
    const size_t stage_bytes = options().db_stage_bytes;  // (tests force several staging passes)
    const uint32_t zchunk = (uint32_t)std::max<size_t>(1, std::min<size_t>(kN, stage_bytes / (per_z_ref * sizeof(uint64_t))));
    DevBuf st;
    if (st.alloc(per_z_ref * zchunk)) return -1;
    for (uint32_t z = 0; z < kN; z += zchunk) {
        const uint32_t nz = std::min(zchunk, kN - z);
        if (hipMemcpyAsync(st.p, database + (size_t)z * per_z_ref, (size_t)nz * per_z_ref * sizeof(uint64_t), hipMemcpyHostToDevice, S->stream) !=
            hipSuccess) {
            st.release();
            return fail("database upload failed");
        }
        launch_db_relayout(st.p, S->db.p, S->s.num_per, S->s.dim0, S->j0, S->dim0_shard, z, nz, S->stream);
        if (hipStreamSynchronize(S->stream) != hipSuccess) {
            st.release();
            return fail("database relayout failed");
        }
    }
    st.release();
    srv_db_loaded(S);
    return 0;
}

int spiral_gpu_server_gen_db(spiral_gpu_server* S, uint64_t seed) {
    if (!S) return fail("null server");
    if (S->db_shared) return fail("this server sweeps another server's database image (share_db): load it through the owner");
    HIP_OK(hipSetDevice(S->device));
    FwdParams fp{};
    fp.dst = S->db.p;
    fp.src_map = fp.dst_map = identity_map();
    fp.n_digits = 1;
    fp.seed = seed;
    fp.p_db = S->p.p_db;
    fp.num_per = S->s.num_per;
    fp.dim0_shard = S->dim0_shard;
    fp.j0 = S->j0;
    const uint64_t items = (uint64_t)S->dim0_shard * S->s.num_per, first = (uint64_t)S->j0 * S->s.num_per;
    const uint64_t chunk = 1u << 16;  // items per launch (4 polynomials each)
    for (uint64_t done = 0; done < items; done += chunk) {
        fp.item_base = first + done;
        launch_ntt_forward(S->tb, fp, LD_DBGEN, ST_DB, (uint32_t)(std::min(chunk, items - done) * 4), S->stream);
    }
    HIP_OK(hipStreamSynchronize(S->stream));
    srv_db_loaded(S);
    return 0;
}

int spiral_gpu_server_load_db_items(spiral_gpu_server* S, const void* items, uint32_t coeff_bits, uint64_t first_item, uint64_t n_items) {
    if (!S) return fail("null server");
    if (S->db_shared) return fail("this server sweeps another server's database image (share_db): load it through the owner");
    HIP_OK(hipSetDevice(S->device));
    const uint64_t total = (uint64_t)S->s.dim0 * S->s.num_per;
    if (first_item > total || n_items > total - first_item) return fail("items [%llu, +%llu) outside the database of %llu", (unsigned long long)first_item,
                                                                         (unsigned long long)n_items, (unsigned long long)total);
    // item i lives at (ii = i % num_per, j = i / num_per): this shard holds the items of j in [j0, j1)
    const uint64_t lo = std::max<uint64_t>(first_item, (uint64_t)S->j0 * S->s.num_per), hi = std::min<uint64_t>(first_item + n_items, (uint64_t)S->j1 * S->s.num_per);
    FwdParams fp{};
    fp.dst = S->db.p;
    fp.src_map = fp.dst_map = identity_map();
    fp.n_digits = 1;
    fp.p_db = S->p.p_db;
    fp.num_per = S->s.num_per;
    fp.dim0_shard = S->dim0_shard;
    fp.j0 = S->j0;
    fp.coeff_bits = coeff_bits;
    // a partial load scatters packed words into the image: an image a batch has converted to limb planes goes back to the packed form first
    if (S->db_format != SPIRAL_GPU_DB_PACKED && srv_db_set_format(S, SPIRAL_GPU_DB_PACKED, S->stream)) return -1;
    if (ingest_items(items, coeff_bits, first_item, lo, hi, 4, S->p.p_db, S->stream, [&](const uint8_t* d_items, uint32_t* d_err, uint64_t first, uint64_t n) {
            fp.items = d_items;
            fp.err = d_err;
            fp.items_first = fp.item_base = first;
            launch_ntt_forward(S->tb, fp, LD_DBGEN, ST_DB, (uint32_t)(n * 4), S->stream);
        }))
        return -1;
    srv_db_loaded(S);
    return 0;
}

int spiral_gpu_server_read_db_item(spiral_gpu_server* S, uint64_t item, uint64_t* out) {
    if (!S || !out) return fail("null argument");
    HIP_OK(hipSetDevice(S->device));
    const uint64_t j = item / S->s.num_per;
    if (j < S->j0 || j >= S->j1) return fail("item %llu is not in this shard", (unsigned long long)item);
    Scratch sc;
    uint64_t* d = sc.get(4 * kRefNtt);
    if (!d) return fail("device allocation failed");
    HIP_OK(hipStreamSynchronize(S->stream));
    launch_db_read_item(S->db.p, d, S->s.num_per, S->dim0_shard, (uint32_t)(j - S->j0), (uint32_t)(item % S->s.num_per), S->stream,
                        holder_of(S)->db_format == SPIRAL_GPU_DB_LIMBS);
    HIP_OK(hipMemcpyAsync(out, d, 4 * kRefNtt * sizeof(uint64_t), hipMemcpyDeviceToHost, S->stream));
    HIP_OK(hipStreamSynchronize(S->stream));
    return 0;
}

static int read_db_region(spiral_gpu_server* S, uint32_t z_begin, uint32_t nz, uint32_t ii0, uint32_t n_ii, uint64_t* out) {
    if (!S || !out) return fail("null argument");
    HIP_OK(hipSetDevice(S->device));
    if (z_begin >= kN || nz == 0 || nz > kN - z_begin) return fail("slot range out of bounds");
    if (n_ii == 0 || ii0 >= S->s.num_per || n_ii > S->s.num_per - ii0) return fail("column range out of bounds");
    const size_t per_z = (size_t)n_ii * 2 * S->dim0_shard * 2;
    const uint32_t zchunk = (uint32_t)std::max<size_t>(1, std::min<size_t>(nz, ((size_t)256 << 20) / (per_z * sizeof(uint64_t))));
    DevBuf st;
    if (st.alloc(per_z * zchunk)) return -1;
    hipError_t e = hipSuccess;
    for (uint32_t z = 0; z < nz && e == hipSuccess; z += zchunk) {
        const uint32_t n = std::min(zchunk, nz - z);
        launch_db_read_slots(S->db.p, st.p, S->s.num_per, S->dim0_shard, z_begin + z, n, ii0, n_ii, S->stream, holder_of(S)->db_format == SPIRAL_GPU_DB_LIMBS);
        e = hipMemcpyAsync(out + (size_t)z * per_z, st.p, (size_t)n * per_z * sizeof(uint64_t), hipMemcpyDeviceToHost, S->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(S->stream);
    }
    st.release();
    if (e != hipSuccess) return fail("database read-back failed: %s", hipGetErrorString(e));
    return 0;
}
int spiral_gpu_server_read_db_slots(spiral_gpu_server* S, uint32_t z_begin, uint32_t nz, uint64_t* out) {
    if (!S) return fail("null argument");
    return read_db_region(S, z_begin, nz, 0, S->s.num_per, out);
}
int spiral_gpu_server_read_db_columns(spiral_gpu_server* S, uint32_t ii_begin, uint32_t n_ii, uint64_t* out) {
    return read_db_region(S, 0, kN, ii_begin, n_ii, out);
}

int spiral_gpu_server_set_db_format(spiral_gpu_server* S, int format) {
    if (!S) return fail("null server");
    if (format != SPIRAL_GPU_DB_PACKED && format != SPIRAL_GPU_DB_LIMBS) return fail("unknown database image format %d", format);
    if (S->db_shared) return fail("this server sweeps another server's database image: convert it through the owner");
    if (!S->have_db) return fail("no database loaded");
    HIP_OK(hipSetDevice(S->device));
    return srv_db_set_format(S, (uint32_t)format, S->stream);
}
int spiral_gpu_server_db_format(spiral_gpu_server* S) { return S ? (int)holder_of(S)->db_format : -1; }
uint64_t spiral_gpu_server_db_device_bytes(spiral_gpu_server* S) {
    if (!S) return 0;
    const spiral_gpu_server* H = holder_of(S);
    return (uint64_t)(H->db.p ? H->db.words : 0) * 8u + (uint64_t)(H->db_limbs.p ? H->db_limbs.words : 0) * 8u;
}

int spiral_gpu_server_fill_db_random(spiral_gpu_server* S, uint64_t seed) {
    if (!S) return fail("null server");
    if (S->db_shared) return fail("this server sweeps another server's database image (share_db): load it through the owner");
    HIP_OK(hipSetDevice(S->device));
    launch_fill_db_random(S->db.p, S->s.num_per, S->dim0_shard, seed, S->stream);
    HIP_OK(hipStreamSynchronize(S->stream));
    srv_db_loaded(S);
    return 0;
}

// A second in-flight query on the same database: `S` gives up its own image and sweeps `owner`'s (the image is read-only on
// the answer path).  Same parameters, shard and device; `owner` must stay alive and must not reload its database while `S`
// answers.  With one handle per query lane, each on its own stream, the latency-bound expansion / fold of one query runs
// under the HBM-bound sweep of the other.
int spiral_gpu_server_share_db(spiral_gpu_server* S, spiral_gpu_server* owner) {
    if (!S || !owner || S == owner) return fail("share_db needs two different servers");
    if (owner->db_shared || owner->zombie) return fail("the owner does not own its database image");
    if (S->device != owner->device || S->j0 != owner->j0 || S->dim0_shard != owner->dim0_shard || S->p.nu1 != owner->p.nu1 || S->p.nu2 != owner->p.nu2 ||
        S->s.num_per != owner->s.num_per)
        return fail("share_db: the servers differ in device, shard or database geometry");
    // the image encodes plaintexts mod p_db (centred lift) and the lane's response switch uses ITS p_db: they must agree
    if (S->p.p_db != owner->p.p_db || S->p.direct_upload != owner->p.direct_upload)
        return fail("share_db: the servers differ in plaintext modulus or query form");
    if (!owner->have_db) return fail("the owner has no database loaded");
    HIP_OK(hipSetDevice(S->device));
    HIP_OK(hipStreamSynchronize(S->stream));
    srv_drop_graphs(S);  // captured sweeps hold the old image's address
    if (S->n_lanes > 0) return fail("share_db: this server's own image is swept by %u lanes", S->n_lanes);
    if (!S->db_shared) S->db.release();
    S->db_limbs.release();  // (a second image of the database this server gives up)
    S->limbs_valid = S->limbs_refused = false;
    S->db_format = SPIRAL_GPU_DB_PACKED;
    if (S->db_owner != owner) {
        lane_detach(S);
        lane_attach(S, owner);
    }
    S->db.p = owner->db.p;
    S->db.words = owner->db.words;
    S->db_shared = true;
    S->have_db = true;
    return 0;
}

int spiral_gpu_server_set_pub_params(spiral_gpu_server* S, const uint64_t* w_left, const uint64_t* w_right, const uint64_t* w,
                                     const uint64_t* v) {
    if (!S) return fail("null server");
    HIP_OK(hipSetDevice(S->device));
    const spiral_gpu_params& p = S->p;
    if (upload_ref_ntt(S, w_left, S->w_left.p, (size_t)S->s.n_left * 2 * p.t_exp)) return -1;
    if (upload_ref_ntt(S, w_right, S->w_right.p, (size_t)S->s.n_right * 2 * p.t_exp_right)) return -1;
    if (upload_ref_ntt(S, w, S->w.p, (size_t)3 * 2 * p.t_conv)) return -1;
    if (upload_ref_ntt(S, v, S->v.p, (size_t)3 * 2 * p.t_conv)) return -1;
    S->have_pp = true;
    return 0;
}

int spiral_gpu_server_set_query(spiral_gpu_server* S, const uint64_t* query) {
    if (!S) return fail("null server");
    HIP_OK(hipSetDevice(S->device));
    if (upload_ref_ntt(S, query, S->query.p, (size_t)S->s.n_query_cts * 2)) return -1;
    S->have_query = true;
    S->have_records = false;
    return 0;
}

}  // extern "C"

namespace {
// expandImproved for the query lanes `lanes` of S (lane 0 = S itself); rounds [r_begin, r_end) of it
int expand_lanes(spiral_gpu_server* S, const Lanes& lanes, uint32_t r_begin = 0, uint32_t r_end = 0xffffffffu) {
    const spiral_gpu_params& p = S->p;
    if (p.direct_upload || S->s.g == 0) {  // nothing to expand: the query ciphertexts are the expanded ones
        const size_t n = p.direct_upload ? (size_t)S->s.n_bits * 2 : 2;
        for (uint32_t q = 0; q < lanes.n && r_begin == 0; q++)
            HIP_OK(hipMemcpyAsync(S->cv.p + lanes.off[q], S->query.p + lanes.off[q], n * kPolyBytes, hipMemcpyDeviceToDevice, S->stream));
        return 0;
    }
    ExpandWork wk{S->ex_raw.p, S->ex_g.p};
    run_expand(S->tb, S->cv.p, S->s.g, p.t_exp, S->w_left.p, p.t_exp_right, S->w_right.p, S->s.ell * p.nu2, S->s.stopround, wk, S->stream, S->query.p, r_begin, r_end,
               S->ex_shard, 3, lanes);
    return 0;
}
}  // namespace

extern "C" {

int spiral_gpu_server_expand(spiral_gpu_server* S) {
    if (!S) return fail("null server");
    HIP_OK(hipSetDevice(S->device));
    if (!S->have_query || !S->have_pp) return fail("query and public parameters must be set before expand");
    return expand_lanes(S, Lanes{});
}

}  // extern "C"

namespace {
// The conversion is: lift (INTT + CRT) of the source rows, t_conv gadget digits of each + forward transforms, then the two
// products.  WHAT selects the ScalToMat part (this shard's first-dimension ciphertexts, src/spiral.cpp:2230-2253), the
// Regev->GSW part (the nu2 further dimensions + fold keys, src/spiral.cpp:2315-2331, 2361-2386), or both with the lifts and
// the digit transforms of the two parts merged into one launch each (their scratch is contiguous).
enum ConvertWhat : uint32_t { CONV_S2M = 1, CONV_GSW = 2, CONV_BOTH = 3 };
int convert_part(spiral_gpu_server* S, uint32_t what, hipStream_t st, bool mark_split = false, const Lanes& lanes = Lanes{}) {
    const spiral_gpu_params& p = S->p;
    const spiral_gpu_shape& s = S->s;
    const uint32_t ps = S->pos_stride, ngs = p.nu2 * s.ell;
    if (ngs == 0) what &= ~CONV_GSW;
    if (what & CONV_S2M) S->have_records = true;  // (a replayed graph sets it in run_group)
    const uint32_t n1 = (what & CONV_S2M) ? S->dim0_shard : 0, n2 = (what & CONV_GSW) ? 2 * ngs : 0;
    const IndexMap map1{1, 2 * ps, 2 * (S->j0 * ps + S->pos_first)};  // row 0 of ct pos(j0 + a)
    const IndexMap map2{2, 2 * ps, 2 * S->pos_rest};                   // rows 0, 1 of the nu2*ell GSW-bit cts
    InvParams ip{};
    ip.src = S->cv.p;
    ip.dst = n1 ? S->cv_raw.p : S->gs_raw_p;
    ip.src_map = n1 ? map1 : map2;
    ip.split = (n1 && n2) ? n1 : 0;
    ip.src_map2 = map2;
    ip.dst_map = identity_map();
    ip.lanes = lanes;
    launch_ntt_inverse(S->tb, ip, IST_CRT, n1 + n2, st);
    FwdParams fp{};
    fp.src = ip.dst;
    fp.dst = n1 ? S->cv_g.p : S->gs_chat_p;
    fp.src_map = fp.dst_map = identity_map();
    fp.n_digits = p.t_conv;
    fp.bits = get_bits_per(p.t_conv);
    fp.lazy_out = lazy_ok(2 * p.t_conv) ? 1 : 0;  // read only by the conversion products, which sum at most 2 * t_conv terms per accumulator
    fp.lanes = lanes;
    launch_ntt_forward(S->tb, fp, LD_DIGIT, ST_PK, (n1 + n2) * p.t_conv, st);
    Scal2MatParams sp{};
    sp.w = S->w.p;
    sp.g = S->cv_g.p;
    sp.cv = S->cv.p;
    sp.cv_pos = IndexMap{1, ps, S->j0 * ps + S->pos_first};
    sp.out = S->keep_cts ? S->cts_keep.p : nullptr;
    sp.qs = (uint32_t*)S->qs.p;
    sp.t_conv = p.t_conv;
    sp.count = S->dim0_shard;
    sp.jm_total = 2 * S->dim0_shard;
    sp.j_base = 0;
    sp.lanes = lanes;
    GswParams gp{};
    gp.w = S->w.p;
    gp.v = S->v.p;
    gp.chat = S->gs_chat_p;
    gp.cv = S->cv.p;
    gp.cv_pos = IndexMap{1, ps, S->pos_rest};
    gp.gsw = S->key.p;  // the GSW matrices ARE the fold key: written once (the reference also keeps Q_neg = G2 - Q, src/spiral.cpp:2361-2379: derived here where a round needs it)
    gp.t_conv = p.t_conv;
    gp.ell = s.ell;
    gp.dims = p.nu2;
    gp.key = nullptr;
    gp.lanes = lanes;
    if (what == CONV_BOTH && !mark_split) {  // the two products are independent: one launch
        launch_convert_products(sp, gp, st);
        return 0;
    }
    if (what & CONV_S2M) launch_scal2mat(sp, st);
    if (mark_split) HIP_OK(hipEventRecord(S->ev[7], st));  // ScalToMat | RegevToGSW split of the reference summary
    if (what & CONV_GSW) launch_regev_to_gsw(gp, st);
    return 0;
}
int convert_scal2mat(spiral_gpu_server* S, hipStream_t st) { return convert_part(S, CONV_S2M, st); }
int convert_gsw(spiral_gpu_server* S, hipStream_t st) { return convert_part(S, CONV_GSW, st); }
}  // namespace

extern "C" {

int spiral_gpu_server_expand(spiral_gpu_server* S);
int spiral_gpu_server_convert(spiral_gpu_server* S);

}  // extern "C"

namespace {
// expand + convert as the launch groups run them.  (Tried: forking the Regev->GSW conversion onto the side stream after
// round `stopround`, where the odd-index side of the expansion is complete, to run beside the remaining even-only rounds.
// Inside a hipGraph the second branch costs far more than the 26 us it hides -- expand + convert went from 340 to 415 us --
// so the group stays one chain.)
int expand_convert(spiral_gpu_server* S) {
    if (spiral_gpu_server_expand(S)) return -1;
    return spiral_gpu_server_convert(S);
}
}  // namespace

extern "C" {

int spiral_gpu_server_convert(spiral_gpu_server* S) {
    if (!S) return fail("null server");
    HIP_OK(hipSetDevice(S->device));
    return convert_part(S, CONV_BOTH, S->stream, !S->use_graphs);
}

int spiral_gpu_server_set_overlap(spiral_gpu_server* S, int on) {
    if (!S) return fail("null server");
    HIP_OK(hipSetDevice(S->device));
    if (on != 0 && on != 2) return fail("overlap mode %d: only 0 (one stream) and 2 (split: the GSW side on a side stream) exist", on);
    if (srv_join_side(S)) return -1;
    if (on == 2) {  // the odd tree needs its own work buffers, and evens / odds must be first-dimension / GSW ciphertexts
        if (S->p.direct_upload || S->s.g == 0 || S->s.stopround == 0 || S->ex_shard.g_log) return fail("split overlap needs query compression with stopround > 0 on an unsharded expansion");
        if (!S->ex_raw2.p && S->ex_raw2.alloc((size_t)S->n_cv * 2 * kN)) return -1;
        if (!S->ex_g2.p && S->ex_g2.alloc(expand_g_polys(S->s.g, S->p.t_exp, S->p.t_exp_right) * kN)) return -1;
    }
    S->overlap = on;
    srv_drop_graphs(S);
    return 0;
}

int spiral_gpu_server_first_dim(spiral_gpu_server* S) {
    if (!S) return fail("null server");
    HIP_OK(hipSetDevice(S->device));
    if (!S->have_db) return fail("no database loaded");
    S->raw_from_acc = false;
    return sweep_one(S, -1);
}

// Pipelined sweep (N > 1): the output columns are independent (src/spiral.cpp:628-999 loops over i and c outside j), so a rank may
// sweep them in K stages of num_per / K ciphertexts and hand every stage's accumulators to its own reduce-scatter while the next
// stage streams the database.  set_sweep_stages lays the accumulators out [stage][rank][ct] (every stage one contiguous block of
// 1/K of the buffer; a rank's reduce-scattered rows still come out in the order fold_local expects); first_dim_stage launches one
// stage; first_dim launches all of them at once into the same layout.
int spiral_gpu_server_set_sweep_stages(spiral_gpu_server* S, uint32_t n_stages) {
    if (!S) return fail("null server");
    HIP_OK(hipSetDevice(S->device));
    if (n_stages == 0 || (n_stages & (n_stages - 1))) return fail("sweep stages must be a power of two");
    const uint32_t k_log = ceil_log2(n_stages);
    if (!sweep_stages_ok(S->s.num_per, 2 * S->dim0_shard, S->fold_g_log, k_log))
        return fail("%u sweep stages: needs the packed layout, whole 64-column blocks per stage (at most %u stages here) and a ciphertext per rank and stage", n_stages,
                    S->s.num_per / 32);
    // stage-by-stage launches exist for the packed image only: an image a batch has converted to limb planes goes back (its holder's lanes re-capture)
    if (k_log && holder_of(S)->db_format != SPIRAL_GPU_DB_PACKED && srv_db_set_format(holder_of(S), SPIRAL_GPU_DB_PACKED, S->stream)) return -1;
    S->sweep_k_log = k_log;
    srv_drop_graphs(S);
    return 0;
}

uint32_t spiral_gpu_server_max_sweep_stages(spiral_gpu_server* S) {
    if (!S) return 0;
    uint32_t k = 0;
    while (k < 16 && sweep_stages_ok(S->s.num_per, 2 * S->dim0_shard, S->fold_g_log, k + 1)) k++;
    return 1u << k;
}

int spiral_gpu_server_first_dim_stage(spiral_gpu_server* S, uint32_t stage) {
    if (!S) return fail("null server");
    HIP_OK(hipSetDevice(S->device));
    if (!S->have_db) return fail("no database loaded");
    if (stage >= (1u << S->sweep_k_log)) return fail("stage %u of %u", stage, 1u << S->sweep_k_log);
    if (S->sweep_k_log == 0) return spiral_gpu_server_first_dim(S);
    S->raw_from_acc = false;
    return sweep_one(S, (int)stage);
}

// One pass over the database for the queries of n servers that sweep the SAME image (an owner and its lanes, create_lane /
// share_db): server b's query records against the database into server b's accumulators.  The launch goes on servers[0]'s stream;
// every other lane's stream is made to wait for it and it for theirs (events), so each lane's run_pre / run_post on its own stream
// stay correctly ordered around it.  Geometries the batched kernel does not cover fall back to one sweep per lane.
int spiral_gpu_server_first_dim_batch(spiral_gpu_server* const* servers, uint32_t n) {
    if (!servers || n == 0) return fail("no servers");
    for (uint32_t b = 0; b < n; b++)
        if (!servers[b]) return fail("null server");
    spiral_gpu_server* S0 = servers[0];
    if (n == 1) return spiral_gpu_server_first_dim(S0);
    if (n > kMaxLanes) return fail("at most %u queries per batched sweep", kMaxLanes);
    HIP_OK(hipSetDevice(S0->device));
    if (!S0->have_db) return fail("no database loaded");
    const uint32_t* qs[kMaxLanes];
    uint64_t* acc[kMaxLanes];
    for (uint32_t b = 0; b < n; b++) {  // every lane is validated before anything is launched: a failure leaves no lane swept
        spiral_gpu_server* S = servers[b];
        if (!S->have_db) return fail("first_dim_batch: server %u has no database", b);
        if (!S->have_records) return fail("first_dim_batch: server %u has not converted its query (run_pre / convert first)", b);
        if (S->db.p != S0->db.p || S->device != S0->device || S->dim0_shard != S0->dim0_shard || S->s.num_per != S0->s.num_per || S->fold_g_log != S0->fold_g_log || S->sweep_k_log != 0)
            return fail("first_dim_batch: server %u does not sweep the same database image with the same layout as server 0", b);
        for (uint32_t c = 0; c < b; c++)
            if (servers[c] == S) return fail("first_dim_batch: server %u listed twice", b);
        qs[b] = (const uint32_t*)S->qs.p;
        acc[b] = S->acc;
    }
    int rc = 0;
    const uint64_t* limbs = limb_image(S0, n, &rc);
    if (rc) return rc;
    if (!limbs && !sweep_batch_ok(S0->s.num_per, 2 * S0->dim0_shard)) {  // (a packed image: limb planes always come back as `limbs`)
        for (uint32_t b = 0; b < n; b++)
            if (spiral_gpu_server_first_dim(servers[b])) return -1;
        return 0;
    }
    for (uint32_t b = 1; b < n; b++) {  // the lanes' records must be complete
        if (servers[b]->stream == S0->stream) continue;  // (lanes on the batch's own stream are ordered by it)
        HIP_OK(hipEventRecord(servers[b]->ev_batch, servers[b]->stream));
        HIP_OK(hipStreamWaitEvent(S0->stream, servers[b]->ev_batch, 0));
    }
    if (sweep_queries(S0, limbs, qs, acc, n, S0->fold_g_log, S0->stream)) return -1;
    for (uint32_t b = 0; b < n; b++) servers[b]->raw_from_acc = false;
    HIP_OK(hipEventRecord(S0->ev_batch, S0->stream));
    for (uint32_t b = 1; b < n; b++)
        if (servers[b]->stream != S0->stream) HIP_OK(hipStreamWaitEvent(servers[b]->stream, S0->ev_batch, 0));
    return 0;
}

int spiral_gpu_server_lift(spiral_gpu_server* S, int reduce_first) {
    if (!S) return fail("null server");
    HIP_OK(hipSetDevice(S->device));
    InvParams ip{};
    ip.src = S->acc;
    ip.dst = S->raw.p;
    ip.src_map = ip.dst_map = identity_map();
    ip.pre_reduce = reduce_first ? 1 : 0;
    launch_ntt_inverse(S->tb, ip, IST_CRT, S->s.num_per * 6, S->stream);
    S->raw_from_acc = true;
    return 0;
}

}  // extern "C"

namespace {
// foldOneFurtherDimension rounds [d0, d0 + rounds) on np0 ciphertexts (src/spiral.cpp:1349-1410); the result is left
// CRT-lifted at the head of S->raw.  src_pk == nullptr: the ciphertexts are already lifted in S->raw.  Otherwise they
// are the PK polynomials [np0][3][2] at src_pk (accumulators, lazy sums when pre_reduce) and the lift is chained into
// the digit transforms (fold_chain_kernel); later rounds chain from the previous round's product the same way.
// finish: the folded ciphertext is the answer; follow with the response modulus switch (spiral_gpu_server_finish).
// raw_addend: with src_pk == nullptr, the transform-domain words of the ciphertexts lifted in S->raw, when the caller still has them
// (the stage API's fold after lift: the accumulators) -- the first round can then take the pair form too (LD_SDIFF on S->raw).
// lanes: the same rounds for every query lane in the same launches (all pointers are lane 0's, kernels.h Lanes).
int finish_lanes(spiral_gpu_server* S, const Lanes& lanes) {
    // row 0 -> q', rows 1.. -> 4*p_db (src/spiral.cpp:1441-1447)
    launch_rescale2(S->raw.p, S->resp.p, 2 * kN, 6 * kN, kQ, S->s.qprime, 4 * S->p.p_db, S->stream, lanes);
    return 0;
}
int run_fold_rounds(spiral_gpu_server* S, uint32_t np0, uint32_t d0, uint32_t rounds, const uint64_t* src_pk, bool pre_reduce, bool finish = false,
                    const uint64_t* raw_addend = nullptr, const Lanes& lanes = Lanes{}) {
    const spiral_gpu_shape& s = S->s;
    S->raw_from_acc = false;  // S->raw ends up holding the folded ciphertext
    uint32_t np = np0;
    auto lift = [&](uint32_t npolys) {
        InvParams ip{};
        ip.src = src_pk;
        ip.dst = S->raw.p;
        ip.src_map = ip.dst_map = identity_map();
        ip.pre_reduce = pre_reduce ? 1 : 0;
        ip.lanes = lanes;
        launch_ntt_inverse(S->tb, ip, IST_CRT, npolys, S->stream);
        src_pk = nullptr;
    };
    uint64_t* out_pk = S->fold_c.p;
    for (uint32_t d = d0; d < d0 + rounds; d++) {
        np /= 2;
        const uint32_t n_src = 2 * np * 6;
        const uint64_t* key = S->key.p + (size_t)d * 3 * s.m2 * kN;
        if (src_pk == out_pk) out_pk = out_pk == S->fold_c.p ? S->fold_c2.p : S->fold_c.p;  // the pair form's product reads its source
        const bool from_raw = !src_pk && raw_addend && S->fold_pair && fold_pair_exact(s.ell);  // lifted already, transform-domain words at hand
        if (from_raw || (src_pk && S->fold_chain && S->fold_pair && fold_pair_exact(s.ell))) {
            // wide round: the lift of all 2 np ciphertexts as one full-occupancy launch, then one digit-difference transform per
            // workgroup (LD_SDIFF) -- no inverse transform is repeated, both kernels run 8 workgroups per CU
            const uint64_t* low = from_raw ? raw_addend : src_pk;
            if (!from_raw) {
                InvParams ip{};
                ip.src = src_pk;
                ip.dst = S->raw.p;
                ip.src_map = ip.dst_map = identity_map();
                ip.pre_reduce = pre_reduce ? 1 : 0;
                ip.lanes = lanes;
                launch_ntt_inverse(S->tb, ip, IST_CRT, n_src, S->stream);
            }
            raw_addend = nullptr;
            FwdParams fp{};
            fp.src = S->raw.p;
            fp.dst = S->fold_d.p;
            fp.src_map = identity_map();
            fp.n_digits = s.ell;
            fp.bits = get_bits_per(s.ell);
            fp.ell = s.ell;
            fp.fold_np = np;
            fp.lazy_out = lazy_ok(3 * s.ell + 1) ? 1 : 0;
            fp.lanes = lanes;
            launch_ntt_forward(S->tb, fp, LD_SDIFF, ST_PK, (n_src / 2) * s.ell, S->stream);
            launch_fold_mac(key, S->fold_d.p, out_pk, s.m2, np, S->stream, s.m2, low, lanes);
            src_pk = out_pk;
            pre_reduce = false;
            continue;
        }
        if (src_pk && S->fold_chain) {
            FoldChainParams cp{};
            cp.src = src_pk;
            cp.dst = S->fold_d.p;
            cp.ell = s.ell;
            cp.bits = get_bits_per(s.ell);
            cp.fold_np = np;
            cp.pre_reduce = pre_reduce ? 1 : 0;
            cp.dpb = fold_dpb(S, n_src);
            cp.lazy_out = lazy_ok(6 * s.ell) ? 1 : 0;  // fold_mac sums 2 * m2 = 6 ell products per accumulator
            cp.lanes = lanes;
            launch_fold_chain(S->tb, cp, n_src, S->stream);
        } else {
            if (src_pk) lift(n_src);
            FwdParams fp{};
            fp.src = S->raw.p;
            fp.dst = S->fold_d.p;
            fp.src_map = identity_map();
            fp.n_digits = s.ell;
            fp.bits = get_bits_per(s.ell);
            fp.ell = s.ell;
            fp.fold_np = np;
            fp.lanes = lanes;
            launch_ntt_forward(S->tb, fp, LD_SDIGIT, ST_PK, n_src * s.ell, S->stream);
        }
        launch_fold_mac_two(key, S->fold_d.p, out_pk, s.m2, s.ell, get_bits_per(s.ell), np, S->stream, lanes);  // the reference's two products, Q_neg derived
        src_pk = out_pk;
        pre_reduce = false;
    }
    if (src_pk) lift(np * 6);
    // (the switch is its own launch: fused into the 6-workgroup lift it serialises 8 coefficients per thread and is slower)
    return finish ? finish_lanes(S, lanes) : 0;
}
}  // namespace

extern "C" {

int spiral_gpu_server_finish(spiral_gpu_server* S);

int spiral_gpu_server_fold(spiral_gpu_server* S) {
    if (!S) return fail("null server");
    HIP_OK(hipSetDevice(S->device));
    if (srv_join_side(S)) return -1;  // no-op inside run_post's capture: run_post joined before capturing
    return run_fold_rounds(S, S->s.num_per, 0, S->p.nu2, nullptr, false, false, S->raw_from_acc ? S->acc : nullptr);  // src/spiral.cpp:1622-1626
}

int spiral_gpu_server_set_fold_ranks(spiral_gpu_server* S, uint32_t n_ranks) {
    if (!S) return fail("null server");
    HIP_OK(hipSetDevice(S->device));
    if (n_ranks == 0 || (n_ranks & (n_ranks - 1)) || n_ranks > S->s.num_per) return fail("fold ranks must be a power of two <= num_per");
    S->fold_g_log = ceil_log2(n_ranks);
    S->sweep_k_log = 0;  // the stage layout depends on the rank count: set_sweep_stages comes after
    srv_drop_graphs(S);
    return 0;
}

// Sharded expansion for one rank of an n_ranks-GPU answer.  After it, expand() leaves this rank with
// the first-dimension ciphertexts of its own j-range (which must be the contiguous block `rank` of n_ranks equal ones) and
// with the GSW bits i = rank mod n_ranks; gsw_bits_pack / one all-gather / gsw_bits_unpack give every rank all of them
// before convert().  Needs the reordered layout (stopround > 0, src/spiral.cpp:2027-2036) and query compression.
int spiral_gpu_server_set_expand_shard(spiral_gpu_server* S, uint32_t rank, uint32_t n_ranks) {
    if (!S) return fail("null server");
    HIP_OK(hipSetDevice(S->device));
    srv_drop_graphs(S);
    if (n_ranks <= 1) {
        S->ex_shard = ExpandShard{};
        return 0;
    }
    if ((n_ranks & (n_ranks - 1)) || n_ranks > S->s.dim0 || rank >= n_ranks) return fail("expansion shard %u of %u: the rank count must be a power of two <= dim0", rank, n_ranks);
    if (S->p.direct_upload || S->s.stopround == 0) return fail("sharded expansion needs query compression with stopround > 0");
    if (S->overlap) return fail("sharded expansion and the split overlap schedule exclude each other");
    const uint32_t per = S->s.dim0 / n_ranks;
    if (S->j0 != rank * per || S->j1 != (rank + 1) * per) return fail("sharded expansion: this server must hold first-dimension block %u of %u, it holds [%u, %u)", rank, n_ranks, S->j0, S->j1);
    S->ex_shard = ExpandShard{rank, ceil_log2(n_ranks), S->p.nu1 - ceil_log2(n_ranks)};
    return 0;
}

size_t spiral_gpu_server_gsw_bits_words(spiral_gpu_server* S) {
    if (!S) return 0;
    const uint32_t G = 1u << S->ex_shard.g_log, n_bits = S->s.ell * S->p.nu2;
    return (size_t)((n_bits + G - 1) / G) * 2 * kN;
}

int spiral_gpu_server_gsw_bits_pack(spiral_gpu_server* S, void* block_out) {
    if (!S || !block_out) return fail("null argument");
    HIP_OK(hipSetDevice(S->device));
    launch_gsw_bits_pack(S->cv.p, (uint64_t*)block_out, S->ex_shard.rank, 1u << S->ex_shard.g_log, S->s.ell * S->p.nu2, S->stream);
    return 0;
}

int spiral_gpu_server_gsw_bits_unpack(spiral_gpu_server* S, const void* gathered) {
    if (!S || !gathered) return fail("null argument");
    HIP_OK(hipSetDevice(S->device));
    launch_gsw_bits_unpack(S->cv.p, (const uint64_t*)gathered, 1u << S->ex_shard.g_log, S->s.ell * S->p.nu2, S->stream);
    return 0;
}

int spiral_gpu_server_finish(spiral_gpu_server* S) {
    if (!S) return fail("null server");
    HIP_OK(hipSetDevice(S->device));
    return finish_lanes(S, Lanes{});
}

int spiral_gpu_server_sync(spiral_gpu_server* S) {
    if (!S) return fail("null server");
    HIP_OK(hipSetDevice(S->device));
    if (srv_join_side(S)) return -1;
    HIP_OK(hipStreamSynchronize(S->stream));
    HIP_OK(hipGetLastError());
    return 0;
}

void* spiral_gpu_server_acc(spiral_gpu_server* S, size_t* bytes) {
    if (!S) return nullptr;
    if (bytes) *bytes = (size_t)S->s.num_per * 6 * kPolyBytes;
    return S->acc;
}

int spiral_gpu_server_set_acc(spiral_gpu_server* S, void* device_ptr) {
    if (!S) return fail("null server");
    S->acc = device_ptr ? (uint64_t*)device_ptr : S->acc_own.p;
    S->raw_from_acc = false;
    srv_drop_graphs(S);  // the accumulator pointer is baked into the captured lift
    return 0;
}

}  // extern "C"

namespace {
// run `body` (kernel launches on S->stream) directly, or capture it once into a hipGraph and replay it
template <class F>
int run_group(spiral_gpu_server* S, int slot, hipStream_t st, F body) {
    if (!S->use_graphs) return body();
    srv_check_epoch(S);
    if (!S->graph[slot]) {
        if (st == nullptr) return fail("graph capture needs a non-default stream");
        HIP_OK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
        int rc = body();
        hipGraph_t g = nullptr;
        hipError_t e = hipStreamEndCapture(st, &g);
        if (rc) {
            if (g) (void)hipGraphDestroy(g);
            return rc;
        }
        if (e != hipSuccess) return fail("hipStreamEndCapture failed: %s", hipGetErrorString(e));
        if (const char* dot = tuning_env("SPIRAL_GRAPH_DOT")) {  // debugging aid: <prefix>.<slot>.dot
            const std::string path = std::string(dot) + "." + std::to_string(slot) + ".dot";
            (void)hipGraphDebugDotPrint(g, path.c_str(), 0);
        }
        e = hipGraphInstantiate(&S->graph[slot], g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (e != hipSuccess) return fail("hipGraphInstantiate failed: %s", hipGetErrorString(e));
    }
    HIP_OK(hipGraphLaunch(S->graph[slot], st));
    if (slot == 0 || slot == 4 || slot == 7 || slot == 9 || slot == 10 || slot == 12) S->have_records = true;  // the groups that hold ScalToMat
    if (slot == 4 || slot == 7 || slot == 9 || slot == 10) S->raw_from_acc = false;  // the groups that hold the sweep: a replay overwrites the accumulators (the eager calls clear it themselves)
    return 0;
}
}  // namespace

extern "C" {

int spiral_gpu_server_run_pre(spiral_gpu_server* S) {
    if (!S) return fail("null server");
    HIP_OK(hipSetDevice(S->device));
    if (!S->have_query || !S->have_pp) return fail("query and public parameters must be set first");
    if (!S->overlap) return run_group(S, 0, S->stream, [&]() { return expand_convert(S); });
    // split: [odd tree of the expansion + Regev->GSW + fold keys] on the side stream, [even tree + ScalToMat] on the main stream, forked
    // HERE (the side depends on the query only: it runs beside the even tree); the fold joins the side stream
    if (srv_join_side(S)) return -1;
    const spiral_gpu_params& p = S->p;
    HIP_OK(hipEventRecord(S->ev_fork, S->stream));  // the previous query's fold has read its keys; the new query is uploaded
    HIP_OK(hipStreamWaitEvent(S->side_stream, S->ev_fork, 0));
    auto half = [&](uint32_t parity, hipStream_t st, uint64_t* raw, uint64_t* g) {
        ExpandWork wk{raw, g};
        run_expand(S->tb, S->cv.p, S->s.g, p.t_exp, S->w_left.p, p.t_exp_right, S->w_right.p, S->s.ell * p.nu2, S->s.stopround, wk, st, S->query.p, 0, 0xffffffffu,
                   ExpandShard{}, parity);
    };
    if (run_group(S, 3, S->side_stream, [&]() {
            half(2u, S->side_stream, S->ex_raw2.p, S->ex_g2.p);
            return convert_gsw(S, S->side_stream);
        }))
        return -1;
    HIP_OK(hipEventRecord(S->ev_join, S->side_stream));
    S->side_pending = true;
    return run_group(S, 0, S->stream, [&]() {
        half(1u, S->stream, S->ex_raw.p, S->ex_g.p);
        return convert_scal2mat(S, S->stream);
    });
}

int spiral_gpu_server_run_query(spiral_gpu_server* S) {
    if (!S) return fail("null server");
    HIP_OK(hipSetDevice(S->device));
    if (!S->have_query || !S->have_pp) return fail("query and public parameters must be set first");
    if (!S->have_db) return fail("no database loaded");
    if (S->overlap) {  // the split schedule is three launch groups on two streams, not one graph
        if (spiral_gpu_server_run_pre(S)) return -1;
        if (spiral_gpu_server_first_dim(S)) return -1;
        return spiral_gpu_server_run_post(S, 0);
    }
    if (srv_join_side(S)) return -1;
    return run_group(S, 4, S->stream, [&]() {
        if (expand_convert(S)) return -1;
        if (spiral_gpu_server_first_dim(S)) return -1;
        return run_fold_rounds(S, S->s.num_per, 0, S->p.nu2, S->acc, false, true);
    });
}

// B <= kMaxLanes whole queries -- one per server: an owner and its lanes (create_lane), all with the same parameters, each with its own
// client's keys and query -- as ONE launch sequence: every launch of expansion, conversion, lift, folding and the switch carries all B queries
// (gridDim.z = B, kernels.h Lanes), and the sweep makes one pass over the database for all of them (sweep_mfma_kernel; sweep_queries).  The reference
// answers one query per process_crtd_query (src/spiral.cpp:2337-2406); this is throughput, not latency: a query's ~50 launch-bound launches
// cost the same ~5 us whether they carry one query or four.  Every lane's buffers end up exactly as after its own run_query.
// The sequence runs on servers[0]'s stream (captured once per lane set into a hipGraph when servers[0] has use_graphs on); the other
// lanes' streams are ordered before and after it with events, as in first_dim_batch.
int spiral_gpu_server_run_query_batch(spiral_gpu_server* const* servers, uint32_t n) {
    if (!servers || n == 0) return fail("no servers");
    for (uint32_t b = 0; b < n; b++)
        if (!servers[b]) return fail("null server");
    spiral_gpu_server* S = servers[0];
    if (n == 1) return spiral_gpu_server_run_query(S);
    if (n > kMaxLanes) return fail("at most %u queries per batch", kMaxLanes);
    HIP_OK(hipSetDevice(S->device));
    Lanes lanes;
    lanes.n = n;
    for (uint32_t b = 0; b < n; b++) {  // every lane is validated before anything is launched
        spiral_gpu_server* L = servers[b];
        if (!L->have_query || !L->have_pp) return fail("run_query_batch: server %u needs its query and public parameters set first", b);
        if (!L->have_db) return fail("run_query_batch: server %u has no database", b);
        if (memcmp(&L->p, &S->p, sizeof(S->p)) != 0 || L->device != S->device || L->j0 != S->j0 || L->dim0_shard != S->dim0_shard || L->cv.words != S->cv.words)
            return fail("run_query_batch: server %u differs from server 0 in parameters, device or shard", b);
        if (L->db.p != S->db.p) return fail("run_query_batch: server %u does not sweep server 0's database image (create_lane / share_db)", b);
        if (L->acc != L->acc_own.p || L->keep_cts || L->overlap || L->fold_g_log || L->sweep_k_log || L->ex_shard.g_log || L->side_pending || L->fold_pair != S->fold_pair ||
            L->fold_chain != S->fold_chain)
            return fail("run_query_batch: server %u has an external accumulator, keep_cts, a split / sharded / staged schedule or other fold options set", b);
        for (uint32_t c = 0; c < b; c++)
            if (servers[c] == L) return fail("run_query_batch: server %u listed twice", b);
        lanes.off[b] = L->w_left.p - S->w_left.p;  // (the first piece of the arena)
    }
    for (uint32_t b = 1; b < n; b++) {  // the lanes' uploads (and whatever else their streams still hold) come first
        if (servers[b]->stream == S->stream) continue;  // (lanes on the batch's own stream are ordered by it: the cheapest arrangement, each other stream costs ~20 us per batch)
        HIP_OK(hipEventRecord(servers[b]->ev_batch, servers[b]->stream));
        HIP_OK(hipStreamWaitEvent(S->stream, servers[b]->ev_batch, 0));
    }
    int rc_l = 0;
    const uint64_t* limbs = limb_image(S, n, &rc_l);  // (not inside the capture below: it may build the image)
    if (rc_l) return rc_l;
    // Batches of four or more: the Regev->GSW conversion runs as soon as the odd (GSW-bit) tree of the expansion is complete, after round `stopround`
    // (src/spiral.cpp:1700-1702: no odd ciphertext is touched later), and ScalToMat after the last round.  The same launches' work in another order -- at
    // these sizes none of them is launch-bound -- but the 24 MiB of GSW matrices and keys per query are then written ~0.3 ms before the sweep instead of
    // right in front of it: dirty lines draining into the database stream cost the matrix-core sweep 30-70 us (profiles/r06_sweep_in_situ_batch.txt).
    const bool gsw_early = n >= 4 && !S->p.direct_upload && S->s.stopround > 0 && S->s.stopround + 1 < S->s.g && S->p.nu2 > 0;
    auto body = [&]() {
        if (gsw_early && tuning_env("SPIRAL_GSW_ORDER") && atoi(tuning_env("SPIRAL_GSW_ORDER")) == 2) {  // (tuning builds only) ScalToMat first, the GSW side last
            if (expand_lanes(S, lanes)) return -1;
            if (convert_part(S, CONV_S2M, S->stream, false, lanes)) return -1;
            if (convert_part(S, CONV_GSW, S->stream, false, lanes)) return -1;
        } else if (gsw_early) {
            if (expand_lanes(S, lanes, 0, S->s.stopround + 1)) return -1;
            if (convert_part(S, CONV_GSW, S->stream, false, lanes)) return -1;
            if (expand_lanes(S, lanes, S->s.stopround + 1)) return -1;
            if (convert_part(S, CONV_S2M, S->stream, false, lanes)) return -1;
        } else {
            if (expand_lanes(S, lanes)) return -1;
            if (convert_part(S, CONV_BOTH, S->stream, false, lanes)) return -1;
        }
        const uint32_t* qs[kMaxLanes];
        uint64_t* acc[kMaxLanes];
        for (uint32_t b = 0; b < n; b++) {
            qs[b] = (const uint32_t*)(S->qs.p + lanes.off[b]);
            acc[b] = S->acc + lanes.off[b];
        }
        if (sweep_queries(S, limbs, qs, acc, n, 0, S->stream)) return -1;  // one pass on the matrix cores where the limb-plane image exists
        return run_fold_rounds(S, S->s.num_per, 0, S->p.nu2, S->acc, false, true, nullptr, lanes);
    };
    int rc = 0;
    if (!S->use_graphs) {
        rc = body();
    } else {
        srv_check_epoch(S);  // (limb_image above may just have changed the image's form)
        // the capture bakes in the lanes' arenas, the image the sweep reads and its kernel (limbs or not)
        bool same = S->graph_batch && S->batch_n == n && S->batch_limbs == limbs;
        for (uint32_t b = 0; same && b < n; b++) same = S->batch_key[b] == servers[b]->w_left.p;
        if (!same) {
            if (S->graph_batch) (void)hipGraphExecDestroy(S->graph_batch);
            S->graph_batch = nullptr;
            HIP_OK(hipStreamBeginCapture(S->stream, hipStreamCaptureModeRelaxed));
            rc = body();
            hipGraph_t g = nullptr;
            const hipError_t e = hipStreamEndCapture(S->stream, &g);
            if (rc || e != hipSuccess) {
                if (g) (void)hipGraphDestroy(g);
                return rc ? rc : fail("hipStreamEndCapture failed: %s", hipGetErrorString(e));
            }
            const hipError_t e2 = hipGraphInstantiate(&S->graph_batch, g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            if (e2 != hipSuccess) return fail("hipGraphInstantiate failed: %s", hipGetErrorString(e2));
            S->batch_n = n;
            S->batch_limbs = limbs;
            for (uint32_t b = 0; b < n; b++) S->batch_key[b] = servers[b]->w_left.p;
        }
        HIP_OK(hipGraphLaunch(S->graph_batch, S->stream));
    }
    if (rc) return rc;
    for (uint32_t b = 0; b < n; b++) {
        servers[b]->have_records = true;
        servers[b]->raw_from_acc = false;
    }
    bool other = false;
    for (uint32_t b = 1; b < n; b++) other |= servers[b]->stream != S->stream;
    if (other) HIP_OK(hipEventRecord(S->ev_batch, S->stream));
    for (uint32_t b = 1; b < n; b++)
        if (servers[b]->stream != S->stream) HIP_OK(hipStreamWaitEvent(servers[b]->stream, S->ev_batch, 0));
    return 0;
}

// One query against n INSTANCES of the database.  An item larger than one plaintext (configs[3]: 100 KB items, 15 360-byte plaintexts) is
// factor = ceil(item / plaintext) database instances (select_params.py:297-298); the client sends ONE query, the server expands and converts it once and
// answers it against every instance: first dimension + folding + response switch per instance, `factor` responses (the reference runs one instance and
// multiplies fdim_us, fold_us and the response size by the factor, select_params.py:409-418).  S holds the query (its public parameters, query, records,
// keys, accumulators); instances[k] hold the images (servers with the same geometry on the same device, each with its own database; S may be one of them).
// pre != 0: expansion + conversion first (run_pre's work), else S must have converted its query already.  Instance k's switched response goes to
// responses + k * 6 * 2048 words and, when finals != null, its folded ciphertext to finals + k * 6 * 2048 (device pointers).  One launch sequence on S's
// stream, sweeps back to back; a hipGraph per (instance set, output buffers) when S has use_graphs on.
int spiral_gpu_server_run_query_instances(spiral_gpu_server* S, spiral_gpu_server* const* instances, uint32_t n, int pre, void* responses, void* finals) {
    if (!S || !instances || n == 0 || !responses) return fail("null argument");
    HIP_OK(hipSetDevice(S->device));
    if (!S->have_query || !S->have_pp) return fail("query and public parameters must be set first");
    if (!pre && !S->have_records) return fail("run_query_instances: the query has not been converted (run_pre first, or pass pre = 1)");
    if (S->overlap || S->fold_g_log || S->sweep_k_log || S->ex_shard.g_log || S->keep_cts || S->acc != S->acc_own.p)
        return fail("run_query_instances needs the default schedule on the query's server (own accumulators, no split / sharded / staged options)");
    std::vector<uint64_t> key{(uint64_t)(uintptr_t)responses, (uint64_t)(uintptr_t)finals, (uint64_t)(pre != 0)};
    for (uint32_t k = 0; k < n; k++) {
        const spiral_gpu_server* I = instances[k];
        if (!I) return fail("null instance %u", k);
        const spiral_gpu_server* H = holder_of(I);
        if (!I->have_db) return fail("run_query_instances: instance %u has no database", k);
        if (I->device != S->device || I->j0 != S->j0 || I->dim0_shard != S->dim0_shard || I->p.nu1 != S->p.nu1 || I->p.nu2 != S->p.nu2 || I->p.p_db != S->p.p_db ||
            I->p.direct_upload != S->p.direct_upload)
            return fail("run_query_instances: instance %u differs from the query's server in device, shard, database geometry or plaintext modulus", k);
        key.push_back((uint64_t)(uintptr_t)I->db.p);
        key.push_back(H->db_epoch);
    }
    if (srv_join_side(S)) return -1;
    auto body = [&]() {
        if (pre && expand_convert(S)) return -1;
        for (uint32_t k = 0; k < n; k++) {
            if (sweep_with(S, holder_of(instances[k]), -1)) return -1;
            if (run_fold_rounds(S, S->s.num_per, 0, S->p.nu2, S->acc, false, true)) return -1;
            HIP_OK(hipMemcpyAsync((uint64_t*)responses + (size_t)k * 6 * kN, S->resp.p, 6 * kPolyBytes, hipMemcpyDeviceToDevice, S->stream));
            if (finals) HIP_OK(hipMemcpyAsync((uint64_t*)finals + (size_t)k * 6 * kN, S->raw.p, 6 * kPolyBytes, hipMemcpyDeviceToDevice, S->stream));
        }
        return 0;
    };
    int rc = 0;
    if (!S->use_graphs) {
        rc = body();
    } else {
        srv_check_epoch(S);
        if (!S->graph_inst || S->inst_key != key) {
            if (S->graph_inst) (void)hipGraphExecDestroy(S->graph_inst);
            S->graph_inst = nullptr;
            HIP_OK(hipStreamBeginCapture(S->stream, hipStreamCaptureModeRelaxed));
            rc = body();
            hipGraph_t g = nullptr;
            const hipError_t e = hipStreamEndCapture(S->stream, &g);
            if (rc || e != hipSuccess) {
                if (g) (void)hipGraphDestroy(g);
                return rc ? rc : fail("hipStreamEndCapture failed: %s", hipGetErrorString(e));
            }
            const hipError_t e2 = hipGraphInstantiate(&S->graph_inst, g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            if (e2 != hipSuccess) return fail("hipGraphInstantiate failed: %s", hipGetErrorString(e2));
            S->inst_key = key;
        }
        HIP_OK(hipGraphLaunch(S->graph_inst, S->stream));
    }
    if (rc) return rc;
    if (pre) S->have_records = true;
    S->raw_from_acc = false;
    return 0;
}

// the same from host buffers, as spiral_gpu_server_answer is to the stages: upload the query, answer it against the n instances, download the n
// responses (n x 6 x 2048 words) and, when finals != null, the folded ciphertexts; total_us (optional): device time of the whole item query
int spiral_gpu_server_answer_instances(spiral_gpu_server* S, spiral_gpu_server* const* instances, uint32_t n, const uint64_t* query, uint64_t* responses,
                                       uint64_t* finals, double* total_us) {
    if (!S || !instances || n == 0 || !query || !responses) return fail("null argument");
    HIP_OK(hipSetDevice(S->device));
    if (spiral_gpu_server_set_query(S, query)) return -1;
    Scratch sc;
    uint64_t* d_resp = sc.get((size_t)n * 6 * kN);
    uint64_t* d_fin = finals ? sc.get((size_t)n * 6 * kN) : nullptr;
    if (!d_resp || (finals && !d_fin)) return fail("device allocation failed");
    HIP_OK(hipEventRecord(S->ev[0], S->stream));
    if (spiral_gpu_server_run_query_instances(S, instances, n, 1, d_resp, d_fin)) return -1;
    HIP_OK(hipEventRecord(S->ev[1], S->stream));
    HIP_OK(hipMemcpyAsync(responses, d_resp, (size_t)n * 6 * kPolyBytes, hipMemcpyDeviceToHost, S->stream));
    if (finals) HIP_OK(hipMemcpyAsync(finals, d_fin, (size_t)n * 6 * kPolyBytes, hipMemcpyDeviceToHost, S->stream));
    HIP_OK(hipStreamSynchronize(S->stream));
    if (total_us) {
        float ms = 0;
        HIP_OK(hipEventElapsedTime(&ms, S->ev[0], S->ev[1]));
        *total_us = ms * 1e3;
    }
    return 0;
}

// The two halves of a distributed fold.  With use_graphs on they replay as hipGraphs too; the buffers the caller hands
// in are baked into the capture, so a graph is dropped when a different pointer arrives.
int spiral_gpu_server_fold_local(spiral_gpu_server* S, const void* acc_chunk, void* out_ct) {
    if (!S || !acc_chunk || !out_ct) return fail("null argument");
    HIP_OK(hipSetDevice(S->device));
    const uint32_t L = S->s.num_per >> S->fold_g_log;
    if (srv_join_side(S)) return -1;
    if (S->graph[5] && (S->cap_chunk != acc_chunk || S->cap_ct != out_ct)) {
        (void)hipGraphExecDestroy(S->graph[5]);
        S->graph[5] = nullptr;
    }
    S->cap_chunk = acc_chunk;
    S->cap_ct = out_ct;
    return run_group(S, 5, S->stream, [&]() {
        if (run_fold_rounds(S, L, 0, S->p.nu2 - S->fold_g_log, (const uint64_t*)acc_chunk, true)) return -1;
        HIP_OK(hipMemcpyAsync(out_ct, S->raw.p, 6 * kPolyBytes, hipMemcpyDeviceToDevice, S->stream));
        return 0;
    });
}

int spiral_gpu_server_fold_root(spiral_gpu_server* S, const void* gathered_cts) {
    if (!S || !gathered_cts) return fail("null argument");
    HIP_OK(hipSetDevice(S->device));
    const uint32_t G = 1u << S->fold_g_log;
    if (srv_join_side(S)) return -1;
    if (S->graph[6] && S->cap_gathered != gathered_cts) {
        (void)hipGraphExecDestroy(S->graph[6]);
        S->graph[6] = nullptr;
    }
    S->cap_gathered = gathered_cts;
    return run_group(S, 6, S->stream, [&]() {
        HIP_OK(hipMemcpyAsync(S->raw.p, gathered_cts, (size_t)G * 6 * kPolyBytes, hipMemcpyDeviceToDevice, S->stream));
        return run_fold_rounds(S, G, S->p.nu2 - S->fold_g_log, S->fold_g_log, nullptr, false, true);
    });
}

// run_pre + first_dim as one group: what a rank does before the collective
int spiral_gpu_server_run_pre_sweep(spiral_gpu_server* S) {
    if (!S) return fail("null server");
    HIP_OK(hipSetDevice(S->device));
    if (!S->have_query || !S->have_pp) return fail("query and public parameters must be set first");
    if (!S->have_db) return fail("no database loaded");
    if (S->overlap) {
        if (spiral_gpu_server_run_pre(S)) return -1;
        return spiral_gpu_server_first_dim(S);
    }
    return run_group(S, 7, S->stream, [&]() {
        if (expand_convert(S)) return -1;
        return spiral_gpu_server_first_dim(S);
    });
}

// the two halves of everything before the accumulator reduce when the expansion is sharded: run_expand_pack = expand + pack of
// this rank's GSW bits into bits_out; [all-gather]; run_unpack_convert_sweep = unpack of the gathered blocks + convert + sweep
int spiral_gpu_server_run_expand_pack(spiral_gpu_server* S, void* bits_out) {
    if (!S || !bits_out) return fail("null argument");
    HIP_OK(hipSetDevice(S->device));
    if (!S->have_query || !S->have_pp) return fail("query and public parameters must be set first");
    if (S->graph[8] && S->cap_bits_out != bits_out) {
        (void)hipGraphExecDestroy(S->graph[8]);
        S->graph[8] = nullptr;
    }
    S->cap_bits_out = bits_out;
    return run_group(S, 8, S->stream, [&]() {
        if (spiral_gpu_server_expand(S)) return -1;
        return spiral_gpu_server_gsw_bits_pack(S, bits_out);
    });
}

int spiral_gpu_server_run_unpack_convert_sweep(spiral_gpu_server* S, const void* gathered) {
    if (!S || !gathered) return fail("null argument");
    HIP_OK(hipSetDevice(S->device));
    if (!S->have_db) return fail("no database loaded");
    if (S->cap_bits_in != gathered)
        for (int g : {9, 11})
            if (S->graph[g]) {
                (void)hipGraphExecDestroy(S->graph[g]);
                S->graph[g] = nullptr;
            }
    S->cap_bits_in = gathered;
    return run_group(S, 9, S->stream, [&]() {
        if (spiral_gpu_server_gsw_bits_unpack(S, gathered)) return -1;
        if (spiral_gpu_server_convert(S)) return -1;
        return spiral_gpu_server_first_dim(S);
    });
}

// The same split so that the all-gather of the GSW bits can run UNDER the database-dependent work: the sweep needs only the
// ScalToMat outputs (this rank's own first-dimension ciphertexts), the GSW bits only feed the folding keys.
// run_scal2mat_sweep = ScalToMat + sweep (after run_expand_pack, while the all-gather is in flight);
// run_unpack_gsw = unpack of the gathered blocks + Regev->GSW conversion (after the all-gather, e.g. under the reduce-scatter).
int spiral_gpu_server_run_scal2mat_sweep(spiral_gpu_server* S) {
    if (!S) return fail("null server");
    HIP_OK(hipSetDevice(S->device));
    if (!S->have_db) return fail("no database loaded");
    return run_group(S, 10, S->stream, [&]() {
        if (convert_scal2mat(S, S->stream)) return -1;
        return spiral_gpu_server_first_dim(S);
    });
}

// ScalToMat alone (the pipelined schedule issues the sweep stage by stage after it)
int spiral_gpu_server_run_scal2mat(spiral_gpu_server* S) {
    if (!S) return fail("null server");
    HIP_OK(hipSetDevice(S->device));
    return run_group(S, 12, S->stream, [&]() { return convert_scal2mat(S, S->stream); });
}

int spiral_gpu_server_run_unpack_gsw(spiral_gpu_server* S, const void* gathered) {
    if (!S || !gathered) return fail("null argument");
    HIP_OK(hipSetDevice(S->device));
    if (S->cap_bits_in != gathered)
        for (int g : {9, 11})
            if (S->graph[g]) {
                (void)hipGraphExecDestroy(S->graph[g]);
                S->graph[g] = nullptr;
            }
    S->cap_bits_in = gathered;
    return run_group(S, 11, S->stream, [&]() {
        if (spiral_gpu_server_gsw_bits_unpack(S, gathered)) return -1;
        return convert_gsw(S, S->stream);
    });
}

int spiral_gpu_server_run_post(spiral_gpu_server* S, int reduce_first) {
    if (!S) return fail("null server");
    HIP_OK(hipSetDevice(S->device));
    if (srv_join_side(S)) return -1;
    return run_group(S, reduce_first ? 2 : 1, S->stream, [&]() {
        return run_fold_rounds(S, S->s.num_per, 0, S->p.nu2, S->acc, reduce_first != 0, true);  // lift chained into round 0, switch into the last
    });
}

int spiral_gpu_server_answer_resident(spiral_gpu_server* S, double stage_us[8]) {
    if (!S) return fail("null server");
    HIP_OK(hipSetDevice(S->device));
    hipStream_t st = S->stream;
    const bool g = S->use_graphs;
    HIP_OK(hipEventRecord(S->ev[0], st));
    if (g) {
        if (spiral_gpu_server_run_pre(S)) return -1;
        HIP_OK(hipEventRecord(S->ev[1], st));
    } else {
        if (spiral_gpu_server_expand(S)) return -1;
        HIP_OK(hipEventRecord(S->ev[1], st));
        if (spiral_gpu_server_convert(S)) return -1;
    }
    HIP_OK(hipEventRecord(S->ev[2], st));
    if (spiral_gpu_server_first_dim(S)) return -1;
    HIP_OK(hipEventRecord(S->ev[3], st));
    if (g) {
        if (spiral_gpu_server_run_post(S, 0)) return -1;
        HIP_OK(hipEventRecord(S->ev[4], st));
        HIP_OK(hipEventRecord(S->ev[5], st));
    } else {
        if (spiral_gpu_server_lift(S, 0)) return -1;
        HIP_OK(hipEventRecord(S->ev[4], st));
        if (spiral_gpu_server_fold(S)) return -1;
        HIP_OK(hipEventRecord(S->ev[5], st));
        if (spiral_gpu_server_finish(S)) return -1;
    }
    HIP_OK(hipEventRecord(S->ev[6], st));
    HIP_OK(hipStreamSynchronize(st));
    HIP_OK(hipGetLastError());
    if (stage_us) {
        float ms[6];
        for (int i = 0; i < 6; i++) HIP_OK(hipEventElapsedTime(&ms[i], S->ev[i], S->ev[i + 1]));
        float total = 0;
        HIP_OK(hipEventElapsedTime(&total, S->ev[0], S->ev[6]));
        // with graphs on, expansion+conversion land in [0] and lift+fold+switch in [2]/[3] as one group
        stage_us[0] = ms[0] * 1e3;
        stage_us[1] = ms[1] * 1e3;
        stage_us[2] = (ms[2] + ms[3]) * 1e3;
        stage_us[3] = ms[4] * 1e3;
        stage_us[4] = ms[5] * 1e3;
        stage_us[5] = ms[2] * 1e3;
        stage_us[6] = total * 1e3;
        stage_us[7] = 0;
        if (!g) {  // ScalToMat share of the conversion bucket (src/spiral.cpp:2254-2256)
            float s2m = 0;
            HIP_OK(hipEventElapsedTime(&s2m, S->ev[1], S->ev[7]));
            stage_us[7] = s2m * 1e3;
        }
    }
    return 0;
}

int spiral_gpu_server_answer(spiral_gpu_server* S, const uint64_t* query, uint64_t* final_ct, uint64_t* response, double stage_us[8]) {
    if (!S) return fail("null server");
    HIP_OK(hipSetDevice(S->device));
    if (spiral_gpu_server_set_query(S, query)) return -1;
    if (spiral_gpu_server_answer_resident(S, stage_us)) return -1;
    if (final_ct) HIP_OK(hipMemcpy(final_ct, S->raw.p, 6 * kPolyBytes, hipMemcpyDeviceToHost));
    if (response) HIP_OK(hipMemcpy(response, S->resp.p, 6 * kPolyBytes, hipMemcpyDeviceToHost));
    return 0;
}

int spiral_gpu_server_keep_cts(spiral_gpu_server* S, int on) {
    if (!S) return fail("null server");
    HIP_OK(hipSetDevice(S->device));
    if (on && !S->cts_keep.p) {
        HIP_OK(hipSetDevice(S->device));
        if (S->cts_keep.alloc((size_t)S->dim0_shard * 6 * kN)) return -1;
    }
    if (S->keep_cts != (on != 0)) srv_drop_graphs(S);  // ScalToMat's output pointer is baked into the captured conversion
    S->keep_cts = on != 0;
    return 0;
}

size_t spiral_gpu_server_buffer_words(spiral_gpu_server* S, int which) {
    if (!S) return 0;
    const spiral_gpu_shape& s = S->s;
    switch (which) {
        case SPIRAL_GPU_BUF_EXPANDED: return (size_t)s.n_bits * 2 * kRefNtt;
        case SPIRAL_GPU_BUF_CTS: return (size_t)S->dim0_shard * 6 * kRefNtt;
        case SPIRAL_GPU_BUF_GSW: return (size_t)S->p.nu2 * 3 * s.m2 * kRefNtt;
        case SPIRAL_GPU_BUF_ACC: return (size_t)s.num_per * 6 * kRefNtt;
        case SPIRAL_GPU_BUF_RAW: return (size_t)s.num_per * 6 * kN;
        case SPIRAL_GPU_BUF_FINAL: return (size_t)6 * kN;
        case SPIRAL_GPU_BUF_RESPONSE: return (size_t)6 * kN;
        default: return 0;
    }
}

int spiral_gpu_server_read(spiral_gpu_server* S, int which, uint64_t* out) {
    if (!S || !out) return fail("null argument");
    HIP_OK(hipSetDevice(S->device));
    HIP_OK(hipStreamSynchronize(S->stream));
    const spiral_gpu_shape& s = S->s;
    const uint32_t ps = S->pos_stride;
    switch (which) {
        case SPIRAL_GPU_BUF_EXPANDED: {
            // first dim0 cts, then the ell*nu2 conversion inputs, each 2 polys
            if (download_pk_as_ref(S, S->cv.p, IndexMap{2, 2 * ps, 2 * S->pos_first}, out, (size_t)s.dim0 * 2)) return -1;
            return download_pk_as_ref(S, S->cv.p, IndexMap{2, 2 * ps, 2 * S->pos_rest}, out + (size_t)s.dim0 * 2 * kRefNtt,
                                      (size_t)s.ell * S->p.nu2 * 2);
        }
        case SPIRAL_GPU_BUF_CTS:
            if (!S->keep_cts) return fail("keep_cts is off");
            return download_pk_as_ref(S, S->cts_keep.p, identity_map(), out, (size_t)S->dim0_shard * 6);
        case SPIRAL_GPU_BUF_GSW: return download_pk_as_ref(S, S->key.p, identity_map(), out, (size_t)S->p.nu2 * 3 * s.m2);
        case SPIRAL_GPU_BUF_ACC: return download_pk_as_ref(S, S->acc, identity_map(), out, (size_t)s.num_per * 6);
        case SPIRAL_GPU_BUF_RAW: HIP_OK(hipMemcpy(out, S->raw.p, (size_t)s.num_per * 6 * kPolyBytes, hipMemcpyDeviceToHost)); return 0;
        case SPIRAL_GPU_BUF_FINAL: HIP_OK(hipMemcpy(out, S->raw.p, 6 * kPolyBytes, hipMemcpyDeviceToHost)); return 0;
        case SPIRAL_GPU_BUF_RESPONSE: HIP_OK(hipMemcpy(out, S->resp.p, 6 * kPolyBytes, hipMemcpyDeviceToHost)); return 0;
        default: return fail("unknown buffer %d", which);
    }
}

size_t spiral_gpu_response_wire_bytes(const spiral_gpu_params* p, uint32_t out_n) {
    if (!p || out_n < 1 || out_n > 16 || p->qprime_bits < 1 || p->qprime_bits > 36 || p->p_db < 2 || p->p_db > (1ull << 40)) return 0;
    return wire_bytes(p, out_n);
}

int spiral_gpu_server_read_response_wire(spiral_gpu_server* S, void* out, size_t capacity) {
    if (!S || !out) return fail("null argument");
    HIP_OK(hipSetDevice(S->device));
    const size_t nbytes = wire_bytes(&S->p, 2);
    if (capacity < nbytes) return fail("response buffer of %zu bytes, the wire form needs %zu", capacity, nbytes);
    if (S->wire.words * 8 < nbytes && (S->wire.release(), S->wire.alloc(nbytes / 8))) return -1;
    launch_response_wire(S->resp.p, S->wire.p, 2 * kN, S->p.qprime_bits, 4 * kN, wire_bits_rest(&S->p), S->stream);
    HIP_OK(hipMemcpyAsync(out, S->wire.p, nbytes, hipMemcpyDeviceToHost, S->stream));
    HIP_OK(hipStreamSynchronize(S->stream));
    return 0;
}

// client side of the wire form (load_modswitched_into_ct, src/client.cpp:90-110): plain host code, no device involved
int spiral_gpu_response_from_wire(const spiral_gpu_params* p, uint32_t out_n, const void* wire, uint64_t* response) {
    if (!p || !wire || !response) return fail("null argument");
    if (spiral_gpu_response_wire_bytes(p, out_n) == 0) return fail("unsupported parameters for the wire form");
    const uint8_t* b = (const uint8_t*)wire;
    const size_t total = wire_bytes(p, out_n);
    size_t bit = 0;
    for (uint32_t r = 0; r <= out_n; r++) {
        const uint32_t w = r == 0 ? p->qprime_bits : wire_bits_rest(p);
        for (size_t i = 0; i < (size_t)out_n * kN; i++, bit += w) {
            unsigned __int128 acc = 0;  // up to 42 + 7 bits starting at a byte boundary
            const size_t first = bit / 8;
            for (size_t k = 0; k < 8 && first + k < total; k++) acc |= (unsigned __int128)b[first + k] << (8 * k);
            response[(size_t)r * out_n * kN + i] = (uint64_t)(acc >> (bit % 8)) & ((1ull << w) - 1);
        }
    }
    return 0;
}

int spiral_gpu_server_write_raw(spiral_gpu_server* S, const uint64_t* raw_cts) {
    if (!S || !raw_cts) return fail("null argument");
    HIP_OK(hipSetDevice(S->device));
    HIP_OK(hipMemcpy(S->raw.p, raw_cts, (size_t)S->s.num_per * 6 * kPolyBytes, hipMemcpyHostToDevice));
    S->raw_from_acc = false;
    return 0;
}

int spiral_gpu_server_time_sweep(spiral_gpu_server* S, int iters, float* avg_ms) {
    if (!S || !avg_ms || iters <= 0) return fail("bad argument");
    HIP_OK(hipSetDevice(S->device));
    if (!S->have_db) return fail("no database loaded");
    HIP_OK(hipSetDevice(S->device));
    HIP_OK(hipEventRecord(S->ev[0], S->stream));
    for (int i = 0; i < iters; i++)
        if (sweep_one(S, -1)) return -1;
    S->raw_from_acc = false;
    HIP_OK(hipEventRecord(S->ev[1], S->stream));
    HIP_OK(hipStreamSynchronize(S->stream));
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, S->ev[0], S->ev[1]));
    *avg_ms = ms / iters;
    return 0;
}

int spiral_gpu_server_time_sweep_batch(spiral_gpu_server* const* servers, uint32_t n, int iters, float* avg_ms) {
    if (!servers || !avg_ms || iters <= 0 || n == 0 || n > kMaxLanes) return fail("bad argument");
    spiral_gpu_server* S = servers[0];
    if (!S) return fail("null server");
    HIP_OK(hipSetDevice(S->device));
    const uint32_t* qs[kMaxLanes];
    uint64_t* acc[kMaxLanes];
    for (uint32_t b = 0; b < n; b++) {
        spiral_gpu_server* L = servers[b];
        if (!L || !L->have_db || !L->have_records || L->db.p != S->db.p || L->device != S->device || L->dim0_shard != S->dim0_shard || L->fold_g_log != S->fold_g_log || L->sweep_k_log)
            return fail("time_sweep_batch: server %u is not a converted lane of server 0's database image", b);
        qs[b] = (const uint32_t*)L->qs.p;
        acc[b] = L->acc;
        L->raw_from_acc = false;
    }
    int rc = 0;
    const uint64_t* limbs = limb_image(S, n, &rc);
    if (rc) return rc;
    HIP_OK(hipDeviceSynchronize());  // (the lanes' streams: their records are complete)
    HIP_OK(hipEventRecord(S->ev[0], S->stream));
    for (int i = 0; i < iters; i++)
        if (sweep_queries(S, limbs, qs, acc, n, S->fold_g_log, S->stream)) return -1;
    HIP_OK(hipEventRecord(S->ev[1], S->stream));
    HIP_OK(hipStreamSynchronize(S->stream));
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, S->ev[0], S->ev[1]));
    *avg_ms = ms / iters;
    return 0;
}

uint64_t spiral_gpu_server_sweep_device_bytes(spiral_gpu_server* S) {
    if (!S) return 0;
    // what one launch has to move on this device: the database in its device layout (7 bytes per word when packed),
    // the query records (48 B per (z, j)) and the accumulators
    const uint64_t n = kN;
    return (uint64_t)db_device_words(2 * S->s.num_per, S->dim0_shard) * 8 + (uint64_t)S->dim0_shard * 48 * n + (uint64_t)S->s.num_per * 6 * n * 8;
}

uint64_t spiral_gpu_server_sweep_bytes(spiral_gpu_server* S) {
    if (!S) return 0;
    // SURVEY.md 8d at 8 B per packed word: database + packed query (3 rows x 2 x dim0) + output (2 limbs as u64)
    const uint64_t n = kN;
    return (uint64_t)S->dim0_shard * S->s.num_per * 4 * n * 8 + (uint64_t)S->dim0_shard * 3 * 2 * n * 8 + (uint64_t)S->s.num_per * 6 * 2 * n * 8;
}

}  // extern "C"
