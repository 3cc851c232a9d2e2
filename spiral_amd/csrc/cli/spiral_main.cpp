// ./spiral -- drop-in for the reference executable's command line and text summary (src/spiral.cpp:1228-1346,
// 209-265), with the server-answer path running on an MI355X through libspiral_gpu.so.
//
//   ./spiral <nu1> <nu2> <IDX_TARGET> <dbfile|"a"> [--random-data] [--direct-upload] [--nonoise] [--show-diff] [--seed N] [--batch B] [--instances F]
//
// The reference fixes its scheme parameters at compile time (-DTEXP ... -DOUTN, include/values.h:78-93,
// select_params.py:337); here the same nine values are read at run time from the environment variables or
// flags of the same names (TEXP, TEXPRIGHT, TCONV, TGSW, QPBITS, PVALUE, QNUMFIRST, QNUMREST, OUTN), defaulting
// to the paper's (20, 256) set.  The stdout lines select_params.py scrapes (:386-401) keep their wording.
// --high-rate selects SpiralPack / SpiralStreamPack (testHighRate, src/testing.cpp:777) with OUTN as out_n.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <random>
#include <string>

#include "client.hpp"

using namespace spiral_cli;
using std::cout;
using std::endl;

static uint64_t now_us() {
    return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static uint64_t param(int argc, char** argv, const char* name, uint64_t dflt) {
    std::string flag = std::string("--") + name;
    for (auto& ch : flag) ch = (char)tolower(ch);
    for (int i = 5; i + 1 < argc; i++)
        if (flag == argv[i]) return strtoull(argv[i + 1], nullptr, 10);
    if (const char* e = getenv(name)) return strtoull(e, nullptr, 10);
    return dflt;
}

#define GPU_OK(x)                                                              \
    do {                                                                       \
        if ((x) != 0) {                                                        \
            fprintf(stderr, "spiral: %s\n", spiral_gpu_last_error());          \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

static size_t bits_to_bytes(size_t bits) { return (size_t)std::llround((double)bits / 8.0); }

// testHighRate (src/testing.cpp:777-1154): SpiralPack / SpiralStreamPack end to end, summary of :626-733
static int run_high_rate(spiral_gpu_params p, uint32_t out_n, uint64_t idx_target, uint64_t seed, bool nonoise, bool show_diff, uint64_t qnum_first) {
    cout << "Using n=" << out_n << endl;
    spiral_gpu_pack_shape s;
    GPU_OK(spiral_gpu_pack_get_shape(&p, out_n, &s));
    const uint64_t total_n = (uint64_t)s.dim0 * s.num_per, db_seed = 1234;
    spiral_gpu_pack_server* srv = nullptr;
    GPU_OK(spiral_gpu_pack_server_create(&p, out_n, 0, &srv));
    GPU_OK(spiral_gpu_pack_server_gen_db(srv, db_seed));
    PackClient cl(p, out_n, seed, nonoise);
    uint64_t t0 = now_us();
    cl.keygen();
    cl.gen_pub_params();
    const double time_key_gen = (double)(now_us() - t0);
    cout << "query: (" << idx_target / s.num_per << " ";
    for (uint32_t i = 0; i < p.nu2; i++) cout << (((idx_target % s.num_per) >> i) & 1) << " ";
    cout << ")" << endl;
    t0 = now_us();
    Poly query = cl.query(idx_target);
    const double time_query_gen = (double)(now_us() - t0);
    GPU_OK(spiral_gpu_pack_server_set_pub_params(srv, cl.w_left.data(), cl.w_right.data(), cl.v.data(), cl.v_w.data()));
    Poly resp((size_t)(out_n + 1) * out_n * N);
    double us[8];
    GPU_OK(spiral_gpu_pack_server_answer(srv, query.data(), resp.data(), nullptr, us));  // warm-up
    GPU_OK(spiral_gpu_pack_server_answer(srv, query.data(), resp.data(), nullptr, us));
    // the response travels in its wire form (bit-packed on the device, include/spiral_gpu.h); the client unpacks and decodes it
    std::vector<uint8_t> wire(spiral_gpu_response_wire_bytes(&p, out_n));
    GPU_OK(spiral_gpu_pack_server_read_response_wire(srv, wire.data(), wire.size()));
    t0 = now_us();
    GPU_OK(spiral_gpu_response_from_wire(&p, out_n, wire.data(), resp.data()));
    Poly pt = cl.decode(resp.data());
    const double time_decoding = (double)(now_us() - t0);
    Poly corr = pack_db_item(db_seed, idx_target, total_n, out_n, p.p_db);
    const bool is_corr = pt == corr;
    cout << "Is correct? : " << (is_corr ? 1 : 0) << endl;
    if (show_diff) {
        size_t shown = 0;
        for (size_t i = 0; i < pt.size() && shown < 10; i++)
            if (pt[i] != corr[i]) {
                cout << i << ": " << pt[i] << " " << corr[i] << endl;
                shown++;
            }
    }
    // print_summary_testing (src/testing.cpp:626-733)
    const size_t logp = (size_t)std::ceil(std::log2((double)p.p_db));
    size_t total_query_size_b = bits_to_bytes((size_t)(((1ull << p.nu1) + 2ull * p.nu2 * p.t_gsw) * N * 56));
    if (qnum_first == 1) total_query_size_b = bits_to_bytes((size_t)N * 56);
    const size_t total_resp_size_b = bits_to_bytes((size_t)out_n * out_n * N * (logp + 2) + (size_t)out_n * N * p.qprime_bits);
    const size_t item_size_b = bits_to_bytes((size_t)out_n * out_n * N * logp);
    const double t_exp = us[0], t_conv = us[1], t_fdim = us[2], t_fold = us[3], t_pack = us[4];
    const double total_time = t_fdim + t_fold + t_pack + t_exp + t_conv;
    cout << "ScalToMat took (CPU·us): 0" << endl;
    cout << "RegevToGSW took (CPU·us): 0" << endl;
    cout << "Expansion took (CPU·us): 0" << endl;
    cout << std::fixed << std::setprecision(0);
    cout << "Database" << endl << endl;
    cout << "                       Number of items: " << total_n << endl;
    cout << "                             Item size: " << item_size_b << endl;
    cout << "Communication" << endl << endl;
    cout << "         Total offline query size (b): " << cl.offline_bytes << endl;
    cout << "          Total online query size (b): " << total_query_size_b << endl;
    cout << "                    Response size (b): " << total_resp_size_b << endl;
    cout << std::fixed << std::setprecision(4);
    cout << "                                Rate : " << ((double)item_size_b / (double)total_resp_size_b) << endl;
    cout << std::fixed << std::setprecision(0);
    cout << endl << endl;
    cout << "Database-independent computation" << endl << endl;
    cout << "              Main expansion  (CPU·us): " << t_exp << endl;
    cout << "                   Conversion (CPU·us): " << t_conv << endl;
    cout << "                        Total (CPU·us): " << (t_exp + t_conv) << endl << endl;
    cout << "Database-dependent computation" << endl << endl;
    cout << "     First dimension multiply (CPU·us): " << t_fdim << endl;
    cout << "                      Folding (CPU·us): " << t_fold << endl;
    cout << "                      Packing (CPU·us): " << t_pack << endl;
    cout << "                        Total (CPU·us): " << (t_fdim + t_fold + t_pack) << endl;
    cout << "                   Throughput (MB / s): " << ((double)total_n * item_size_b / total_time) << endl << endl;
    cout << "Client computation" << endl << endl;
    cout << "               Key generation (CPU·us): " << time_key_gen << endl;
    cout << "             Query generation (CPU·us): " << time_query_gen << endl;
    cout << "                     Decoding (CPU·us): " << time_decoding << endl << endl;
    cout << "GPU extras" << endl << endl;
    cout << "      Sweep kernels alone (GPU·us): " << us[5] << "  (" << (double)out_n * out_n * spiral_gpu_pack_server_sweep_bytes(srv) / us[5] / 1e3 << " GB/s)" << endl;
    cout << "      Whole answer, device (GPU·us): " << us[6] << endl;
    spiral_gpu_pack_server_destroy(srv);
    return is_corr ? 0 : 2;
}

int main(int argc, char** argv) {
    if (argc < 4) {
        fprintf(stderr, "usage: %s <nu1> <nu2> <IDX_TARGET> [dbfile|a] [--random-data] [--direct-upload] [--nonoise] [--show-diff] [--output-err F] [--seed N]\n", argv[0]);
        return 1;
    }
    const uint32_t nu1 = (uint32_t)strtol(argv[1], nullptr, 10), nu2 = (uint32_t)strtol(argv[2], nullptr, 10);
    const uint64_t total_n = (1ull << nu1) * (1ull << nu2);
    const uint64_t idx_target = strtoull(argv[3], nullptr, 10);
    bool nonoise = false, random_data = false, show_diff = false, direct_flag = false, high_rate = false;
    uint32_t batch = 0, instances = 0;
    // as the reference (random_device, src/core.cpp:202; it labels its own generator NOT SECURE): two words of it.
    // This client is a test harness for the server path, not a hardened client.
    std::random_device rd;
    uint64_t seed = ((uint64_t)rd() << 32) ^ (uint64_t)rd();
    for (int i = 5; i < argc; i++) {  // flags are only parsed after the db filename (src/spiral.cpp:1250-1303)
        if (!strcmp(argv[i], "--nonoise")) { cout << "Using no noise" << endl; nonoise = true; }
        if (!strcmp(argv[i], "--high-rate")) { cout << "Using high rate variant..." << endl; high_rate = true; }
        if (!strcmp(argv[i], "--random-data")) { cout << "Using random data..." << endl; random_data = true; }
        if (!strcmp(argv[i], "--show-diff")) { cout << "Showing diff..." << endl; show_diff = true; }
        if (!strcmp(argv[i], "--direct-upload")) { cout << "Direct uploading of query (no compression)" << endl; direct_flag = true; }
        if (!strcmp(argv[i], "--seed") && i + 1 < argc) seed = strtoull(argv[++i], nullptr, 10);
        // --batch B (2 .. 8; not a flag of the reference, which answers one query per process): after the reference's own single-query run, B clients --
        // own keys, own indices -- are answered by ONE call of spiral_gpu_server_run_query_batch and each is decoded and checked
        if (!strcmp(argv[i], "--batch") && i + 1 < argc) batch = (uint32_t)strtoul(argv[++i], nullptr, 10);
        // --instances F (2 .. 16; not a flag of the reference either: select_params.py:297-298 runs ONE instance and multiplies by factor = ceil(item size /
        // plaintext size)): an item of F plaintexts = F instances of the database; the one query is converted once and answered against all of them by ONE
        // call of spiral_gpu_server_answer_instances, every plaintext of the item is decoded and checked
        if (!strcmp(argv[i], "--instances") && i + 1 < argc) instances = (uint32_t)strtoul(argv[++i], nullptr, 10);
        // --output-err F (src/spiral.cpp:1287-1291) asks the reference to dump its empirical noise statistics (analyze_err.py's
        // input): those are outside this path.  The flag and its file name are consumed so that a driver's
        // command line parses the same way, and the file is not written.
        if (!strcmp(argv[i], "--output-err") && i + 1 < argc) { cout << "--output-err " << argv[++i] << ": noise statistics are not produced by this build (ignored)" << endl; }
    }
    if (idx_target >= total_n) {
        fprintf(stderr, "spiral: IDX_TARGET %llu out of range (n = %llu)\n", (unsigned long long)idx_target, (unsigned long long)total_n);
        return 1;
    }

    spiral_gpu_params p{};
    p.nu1 = nu1;
    p.nu2 = nu2;
    p.t_exp = (uint32_t)param(argc, argv, "TEXP", 8);
    p.t_exp_right = (uint32_t)param(argc, argv, "TEXPRIGHT", 56);
    p.t_conv = (uint32_t)param(argc, argv, "TCONV", 4);
    p.t_gsw = (uint32_t)param(argc, argv, "TGSW", 8);
    p.qprime_bits = (uint32_t)param(argc, argv, "QPBITS", 20);
    p.p_db = param(argc, argv, "PVALUE", 256);
    const uint64_t qnum_first = param(argc, argv, "QNUMFIRST", direct_flag ? (1ull << nu1) : 1);
    const uint64_t qnum_rest = param(argc, argv, "QNUMREST", direct_flag ? (uint64_t)p.t_gsw * nu2 : 0);
    const bool du_first = qnum_first >= (1ull << nu1), du_rest = qnum_rest >= (uint64_t)nu2 * p.t_gsw;  // src/spiral.cpp:2060-2061
    if (du_first != du_rest || (!du_first && (qnum_first != 1 || qnum_rest != 0))) {
        fprintf(stderr, "spiral: unsupported QNUMFIRST/QNUMREST combination (supported: 1/0 and 2^nu1 / t_GSW*nu2)\n");
        return 1;
    }
    p.direct_upload = du_first ? 1 : 0;
    if (du_rest) cout << "directly uploading Regev -> GSW ciphertexts" << endl;

    if (spiral_gpu_device_count() <= 0) {
        fprintf(stderr, "spiral: no ROCm device found; this build has no CPU path\n");
        return 1;
    }
    if (high_rate) return run_high_rate(p, (uint32_t)param(argc, argv, "OUTN", 2), idx_target, seed, nonoise, show_diff, qnum_first);
    spiral_gpu_shape s;
    GPU_OK(spiral_gpu_get_shape(&p, &s));
    cout << "dim0: " << s.dim0 << endl;
    cout << "num_per: " << s.num_per << endl;

    // ---- database (load_db, src/spiral.cpp:1028-1172): explicit seeded database generated on the device; with
    // --random-data the same (a full-size database is cheap on the GPU, so the result is still checkable)
    const uint64_t db_seed = 1234;
    spiral_gpu_server* srv = nullptr;
    GPU_OK(spiral_gpu_server_create(&p, 0, 0, 0, &srv));
    cout << "starting generation of db" << endl;
    GPU_OK(spiral_gpu_server_gen_db(srv, db_seed));
    cout << "done loading/generating db." << endl;
    (void)random_data;

    // ---- client: keys, public parameters, query
    double time_key_gen = 0, time_query_gen = 0, time_decoding = 0;
    Client cl(p, seed, nonoise);
    uint64_t t0 = now_us();
    cl.keygen();
    cl.gen_pub_params();
    time_key_gen = (double)(now_us() - t0);
    if (!p.direct_upload) {
        cout << "g = " << s.g << endl;
        cout << "stopround = " << s.stopround << endl;
    }
    t0 = now_us();
    Poly query = cl.query(idx_target);
    time_query_gen = (double)(now_us() - t0);

    // ---- server
    cout << "Beginning query processing..." << endl;
    GPU_OK(spiral_gpu_server_set_pub_params(srv, cl.w_left.data(), cl.w_right.data(), cl.w.data(), cl.v.data()));
    Poly final_ct(6 * N), resp(6 * N);
    double us[8];
    GPU_OK(spiral_gpu_server_answer(srv, query.data(), final_ct.data(), resp.data(), us));  // warm-up (table upload, first launches)
    GPU_OK(spiral_gpu_server_answer(srv, query.data(), final_ct.data(), resp.data(), us));
    const double time_expansion_main = us[0], time_conversion = us[1], time_first_multiply = us[2], time_folding = us[3];
    cout << std::fixed << std::setprecision(0);
    cout << "Expansion took (CPU·us): " << time_expansion_main << endl;
    if (p.direct_upload) cout << "directly uploading Regev ciphertexts" << endl;
    cout << "ScalToMat took (CPU·us): " << us[7] << endl;
    cout << "RegevToGSW took (CPU·us): " << (us[1] - us[7]) << endl;
    cout << "done folding" << endl;
    cout << "Done with query processing!" << endl;

    // ---- client decode + check_final (src/spiral.cpp:1412-1494)
    // the response travels in its wire form (bit-packed on the device, include/spiral_gpu.h); the client unpacks and decodes it
    std::vector<uint8_t> wire(spiral_gpu_response_wire_bytes(&p, 2));
    GPU_OK(spiral_gpu_server_read_response_wire(srv, wire.data(), wire.size()));
    t0 = now_us();
    GPU_OK(spiral_gpu_response_from_wire(&p, 2, wire.data(), resp.data()));
    Poly pt = cl.decode(resp.data());
    time_decoding = (double)(now_us() - t0);
    Poly corr = db_item(db_seed, idx_target, p.p_db);
    const bool is_corr = pt == corr;
    cout << "Is correct?: " << (is_corr ? 1 : 0) << endl;
    if (show_diff)
        for (size_t i = 0; i < pt.size(); i++)
            if (pt[i] != corr[i]) cout << i << " " << corr[i] << ", " << pt[i] << endl;

    // ---- --batch B: the same server, B queries of B clients in one launch sequence (include/spiral_gpu.h, spiral_gpu_server_run_query_batch)
    double batch_us = 0;
    bool batch_corr = true;
    if (batch >= 2 && batch <= 8) {
        std::vector<spiral_gpu_server*> lanes{srv};
        std::vector<Client> clients;
        std::vector<uint64_t> idxs;
        clients.reserve(batch);
        for (uint32_t b = 0; b < batch; b++) {
            if (b) {
                spiral_gpu_server* lane = nullptr;
                GPU_OK(spiral_gpu_server_create_lane(srv, &lane));
                GPU_OK(spiral_gpu_server_set_stream(lane, spiral_gpu_server_get_stream(srv)));  // the lanes of a batch on one stream: no event ordering around the launch sequence
                lanes.push_back(lane);
            }
            clients.emplace_back(p, seed + 1 + b, nonoise);
            clients[b].keygen();
            clients[b].gen_pub_params();
            idxs.push_back((idx_target + 1 + 7919ull * b) % total_n);
            GPU_OK(spiral_gpu_server_set_pub_params(lanes[b], clients[b].w_left.data(), clients[b].w_right.data(), clients[b].w.data(), clients[b].v.data()));
            Poly qb = clients[b].query(idxs[b]);
            GPU_OK(spiral_gpu_server_set_query(lanes[b], qb.data()));
        }
        GPU_OK(spiral_gpu_server_use_graphs(srv, 1));
        const int reps = 10;
        for (int it = 0; it < 2 + reps; it++) {  // two untimed passes (graph capture, first replay), then `reps` timed ones
            if (it == 2) {
                GPU_OK(spiral_gpu_server_sync(srv));
                t0 = now_us();
            }
            GPU_OK(spiral_gpu_server_run_query_batch(lanes.data(), batch));
        }
        GPU_OK(spiral_gpu_server_sync(srv));
        batch_us = (double)(now_us() - t0) / reps;
        cout << "Batch of " << batch << " queries, Is correct?:";
        for (uint32_t b = 0; b < batch; b++) {
            GPU_OK(spiral_gpu_server_sync(lanes[b]));
            GPU_OK(spiral_gpu_server_read_response_wire(lanes[b], wire.data(), wire.size()));
            GPU_OK(spiral_gpu_response_from_wire(&p, 2, wire.data(), resp.data()));
            const bool ok = clients[b].decode(resp.data()) == db_item(db_seed, idxs[b], p.p_db);
            batch_corr = batch_corr && ok;
            cout << " " << (ok ? 1 : 0);
        }
        cout << endl;
        GPU_OK(spiral_gpu_server_use_graphs(srv, 0));
        for (uint32_t b = 1; b < batch; b++) spiral_gpu_server_destroy(lanes[b]);
    } else if (batch) {
        fprintf(stderr, "spiral: --batch takes 2 .. 8\n");
        return 1;
    }

    // ---- --instances F: the item at idx_target = plaintext idx_target of F databases (instance k seeded db_seed + k; instance 0 is the server above)
    double item_us = 0;
    bool item_corr = true;
    if (instances >= 2 && instances <= 16) {
        std::vector<spiral_gpu_server*> inst{srv};
        for (uint32_t k = 1; k < instances; k++) {
            spiral_gpu_server* sv = nullptr;
            GPU_OK(spiral_gpu_server_create(&p, 0, 0, 0, &sv));
            GPU_OK(spiral_gpu_server_gen_db(sv, db_seed + k));
            inst.push_back(sv);
        }
        std::vector<uint64_t> resps((size_t)instances * 6 * N);
        GPU_OK(spiral_gpu_server_use_graphs(srv, 1));
        for (int it = 0; it < 3; it++)  // capture, a replay, the timed replay
            GPU_OK(spiral_gpu_server_answer_instances(srv, inst.data(), instances, query.data(), resps.data(), nullptr, &item_us));
        cout << "Item of " << instances << " plaintexts, Is correct?:";
        for (uint32_t k = 0; k < instances; k++) {
            const bool ok = cl.decode(resps.data() + (size_t)k * 6 * N) == db_item(db_seed + k, idx_target, p.p_db);
            item_corr = item_corr && ok;
            cout << " " << (ok ? 1 : 0);
        }
        cout << endl;
        GPU_OK(spiral_gpu_server_use_graphs(srv, 0));
        for (uint32_t k = 1; k < instances; k++) spiral_gpu_server_destroy(inst[k]);
    } else if (instances) {
        fprintf(stderr, "spiral: --instances takes 2 .. 16\n");
        return 1;
    }

    // ---- print_summary (src/spiral.cpp:209-265)
    const double pt_mod = std::log2((double)p.p_db);
    const size_t pt_elem_size = (size_t)((2.0 * 2 * N * pt_mod) / 8.0);
    const size_t b_per_elem = (size_t)((double)N * 56 / 8.0);
    const size_t dim0_query_size = (size_t)(qnum_first + qnum_rest) * 2 * b_per_elem;
    const size_t total_resp_size = wire.size();  // = ((n0 n0 N (log2 p + 2)) + (n0 N q'bits)) / 8, src/spiral.cpp:231-233
    cout << endl;
    cout << "PIR over n=" << total_n << " elements of size " << pt_elem_size << " bytes each." << endl;
    cout << "The database is structured as " << (1 << nu1) << " x 2^" << nu2 << "." << endl;
    cout << endl;
    cout << "Communication" << endl;
    cout << endl;
    cout << "         Total offline query size (b): " << cl.offline_bytes << endl;
    cout << "                  First dimension (b): " << dim0_query_size << endl;
    cout << "       Total for other dimensions (b): " << 0 << endl;
    cout << "          Total online query size (b): " << dim0_query_size << endl;
    cout << "                    Response size (b) : " << total_resp_size << endl;
    cout << endl;
    cout << endl;
    cout << "Database-independent computation" << endl;
    cout << endl;
    cout << "              Main expansion  (CPU·us): " << time_expansion_main << endl;
    cout << "  Further dimension expansion (CPU·us): " << 0 << endl;
    cout << "                   Conversion (CPU·us): " << time_conversion << endl;
    cout << "                        Total (CPU·us): " << (time_expansion_main + time_conversion) << endl;
    cout << endl;
    cout << "Database-dependent computation" << endl;
    cout << endl;
    cout << "     First dimension multiply (CPU·us): " << time_first_multiply << endl;
    cout << "                      Folding (CPU·us): " << time_folding << endl;
    cout << "                        Total (CPU·us): " << (time_first_multiply + time_folding) << endl;
    cout << endl;
    cout << "Client computation" << endl;
    cout << endl;
    cout << "               Key generation (CPU·us): " << time_key_gen << endl;
    cout << "             Query generation (CPU·us): " << time_query_gen << endl;
    cout << "                     Decoding (CPU·us): " << time_decoding << endl;
    cout << endl;
    // MI355X extras (not scraped by select_params.py)
    cout << "GPU extras" << endl;
    cout << endl;
    cout << "        Sweep kernel alone (GPU·us): " << us[5] << endl;
    cout << "      Response switch kernel (GPU·us): " << us[4] << endl;
    cout << "        Whole answer, device (GPU·us): " << us[6] << endl;
    if (batch_us > 0) cout << "   Batch of " << batch << " queries, wall (GPU·us): " << batch_us << endl;
    if (item_us > 0) cout << "   Item of " << instances << " plaintexts (one query, " << instances << " database instances), device (GPU·us): " << item_us << endl;
    spiral_gpu_server_destroy(srv);
    return (is_corr && batch_corr && item_corr) ? 0 : 2;
}
