// Host client for the ./spiral command line: key generation, public parameters, query encoding and response
// decoding (the CLIENT half of the reference: src/client.cpp and the client parts of src/spiral.cpp:2040-2331,
// 1451-1494).  Not part of the server hot path.  All ring arithmetic goes through libspiral_gpu.so's C ABI
// (to_ntt / multiply / add / automorph ...), so this program contains no CPU NTT; the only CPU polynomial
// arithmetic is the final decode product modulo q' (the reference uses HEXL there, src/util.cpp:213-274).
#pragma once
#include <cstdint>
#include <random>
#include <vector>

#include "../../../include/spiral_gpu.h"

namespace spiral_cli {

constexpr uint32_t N = 2048;
constexpr uint64_t P = 268369921ull, B = 249561089ull, Q = P * B;
using Poly = std::vector<uint64_t>;  // raw: N words per polynomial; NTT form: 2N words per polynomial

struct Client {
    spiral_gpu_params p;
    spiral_gpu_shape s;
    bool nonoise = false;
    std::mt19937_64 rng;
    std::vector<double> cdf;
    Poly sr;  // 1 x 1 raw: Regev secret
    Poly sp;  // n0 x 1 raw: matrix-Regev secret
    Poly w_left, w_right, w, v;  // public parameters, reference NTT layout
    uint64_t offline_bytes = 0;

    Client(const spiral_gpu_params& params, uint64_t seed, bool nonoise_);
    void keygen();
    void gen_pub_params();
    Poly query(uint64_t idx_target);
    Poly decode(const uint64_t* response) const;  // -> n0 x n2 raw plaintext in [0, p_db)

  private:
    uint64_t sample_noise();
    Poly noise_polys(size_t n);
    Poly uniform_polys(size_t n);
    Poly regev_samples(size_t m);                                  // n0 x m NTT
    Poly fresh_public_key(size_t m);                               // n1 x m NTT
    Poly public_encryptions(uint32_t count, uint32_t t_dim);       // count x (n0 x t_dim) NTT
    Poly encrypt_simple_regev(const Poly& sigma_raw);              // n0 x 1 NTT
};

// SpiralPack / SpiralStreamPack client (src/testing.cpp:905-1006, 1086-1122): base_dim x 1 Regev secret sr, out_n x 1
// matrix secret Sp, packing keys v_W, expansion keys, conversion key V.
struct PackClient {
    spiral_gpu_params p;
    spiral_gpu_pack_shape s;
    uint32_t out_n;
    bool nonoise = false;
    std::mt19937_64 rng;
    std::vector<double> cdf;
    Poly sr, sp;
    Poly w_left, w_right, v, v_w;
    uint64_t offline_bytes = 0;

    PackClient(const spiral_gpu_params& params, uint32_t out_n_, uint64_t seed, bool nonoise_);
    void keygen();
    void gen_pub_params();
    Poly query(uint64_t idx_target);
    Poly decode(const uint64_t* response) const;  // -> out_n x out_n raw plaintexts

  private:
    uint64_t sample_noise();
    Poly noise_polys(size_t n);
    Poly uniform_polys(size_t n);
    Poly regev_samples(size_t m);
    Poly expansion_keys(uint32_t count, uint32_t t_dim);
    Poly encrypt_simple_regev(const Poly& sigma_raw);
};
Poly pack_db_item(uint64_t seed, uint64_t item, uint64_t total_n, uint32_t out_n, uint64_t p_db);

// plaintext item of the seeded explicit database (same generator as spiral_gpu_server_gen_db)
Poly db_item(uint64_t seed, uint64_t item, uint64_t p_db);

}  // namespace spiral_cli
