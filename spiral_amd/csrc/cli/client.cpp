#include "client.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>

namespace spiral_cli {

namespace {
typedef unsigned __int128 u128;

void ok(int rc, const char* what) {
    if (rc != 0) throw std::runtime_error(std::string(what) + ": " + spiral_gpu_last_error());
}
Poly to_ntt(const Poly& raw) {
    Poly out(raw.size() * 2);
    ok(spiral_gpu_to_ntt(out.data(), raw.data(), raw.size() / N, 1), "to_ntt");
    return out;
}
Poly add(const Poly& a, const Poly& b) {
    Poly out(a.size());
    ok(spiral_gpu_add(out.data(), a.data(), b.data(), a.size() / (2 * N)), "add");
    return out;
}
Poly mul_by_const(const Poly& single, const Poly& a) {
    Poly out(a.size());
    ok(spiral_gpu_mul_by_const(out.data(), single.data(), a.data(), a.size() / (2 * N)), "mul_by_const");
    return out;
}
Poly multiply(const Poly& a, const Poly& b, size_t rs, size_t ms, size_t cs) {
    Poly out(rs * cs * 2 * N);
    ok(spiral_gpu_multiply(out.data(), a.data(), b.data(), rs, ms, cs), "multiply");
    return out;
}
Poly automorph(const Poly& raw, uint64_t t) {
    Poly out(raw.size());
    ok(spiral_gpu_automorph(out.data(), raw.data(), raw.size() / N, t), "automorph");
    return out;
}
Poly invert(const Poly& raw) {  // Q - a (src/poly.cpp:269)
    Poly out(raw.size());
    for (size_t i = 0; i < raw.size(); i++) out[i] = Q - raw[i];
    return out;
}
uint32_t bits_per(uint32_t dim) { return dim == 56 ? 1u : 56u / dim + 1u; }  // include/util.h:34
// buildGadget (src/util.cpp:89-106): raw rows x cols constant polynomials
Poly build_gadget(size_t rows, size_t cols) {
    Poly g(rows * cols * N, 0);
    size_t ne = cols / rows;
    uint32_t bits = bits_per((uint32_t)ne);
    for (size_t i = 0; i < rows; i++)
        for (size_t j = 0; j < ne; j++)
            if ((uint64_t)bits * j < 64) g[(i * cols + (i + j * rows)) * N] = 1ull << (bits * j);
    return g;
}
uint64_t inv_mod_q(uint64_t a) {
    __int128 t = 0, nt = 1, r = Q, nr = a % Q;
    while (nr != 0) {
        __int128 q = r / nr, tmp = t - q * nt;
        t = nt;
        nt = tmp;
        tmp = r - q * nr;
        r = nr;
        nr = tmp;
    }
    if (t < 0) t += Q;
    return (uint64_t)t;
}
uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
}  // namespace

Poly db_item(uint64_t seed, uint64_t item, uint64_t p_db) {
    Poly pt(4 * N);
    for (uint64_t k = 0; k < 4ull * N; k++) pt[k] = splitmix64(seed ^ (item * (4ull * N) + k)) % p_db;
    return pt;
}

Client::Client(const spiral_gpu_params& params, uint64_t seed, bool nonoise_) : p(params), nonoise(nonoise_), rng(seed) {
    ok(spiral_gpu_get_shape(&p, &s), "get_shape");
    double acc = 0;  // discrete Gaussian of width 6.4 over [-64, 64] (src/core.cpp:182-207)
    for (int i = -64; i <= 64; i++) {
        acc += std::exp(-M_PI * (double)i * i / (6.4 * 6.4));
        cdf.push_back(acc);
    }
}

uint64_t Client::sample_noise() {
    if (nonoise) return 0;
    double u = std::uniform_real_distribution<double>(0.0, cdf.back())(rng);
    int64_t v = (int64_t)(std::lower_bound(cdf.begin(), cdf.end(), u) - cdf.begin()) - 64;
    if (v > 64) v = 64;
    return (uint64_t)((v + (int64_t)Q) % (int64_t)Q);
}
Poly Client::noise_polys(size_t n) {
    Poly a(n * N);
    for (auto& x : a) x = sample_noise();
    return a;
}
Poly Client::uniform_polys(size_t n) {
    Poly a(n * N);
    for (auto& x : a) x = rng() % Q;
    return a;
}

void Client::keygen() {  // src/client.cpp:21-46
    sr = noise_polys(1);
    sp = noise_polys(2);
}

// getRegevSample x m (src/client.cpp:147-174): column i = ( -a_i ; a_i * s + e_i )
Poly Client::regev_samples(size_t m) {
    Poly a = uniform_polys(m), e = noise_polys(m);
    Poly a_ntt = to_ntt(a), e_ntt = to_ntt(e), s_ntt = to_ntt(sr), ainv_ntt = to_ntt(invert(a));
    Poly b = add(mul_by_const(s_ntt, a_ntt), e_ntt);
    Poly out(2 * m * 2 * N);
    std::copy(ainv_ntt.begin(), ainv_ntt.end(), out.begin());
    std::copy(b.begin(), b.end(), out.begin() + m * 2 * N);
    return out;
}

// to_ntt(get_fresh_public_key_raw(Sp, m)) (src/client.cpp:48-68): ( -A ; Sp*A + E ), n1 x m
Poly Client::fresh_public_key(size_t m) {
    Poly a = uniform_polys(m), e = noise_polys(2 * m);
    Poly a_ntt = to_ntt(a), e_ntt = to_ntt(e), sp_ntt = to_ntt(sp), ainv_ntt = to_ntt(invert(a));
    Poly bp = multiply(sp_ntt, a_ntt, 2, 1, m);
    Poly b = add(e_ntt, bp);
    Poly out(3 * m * 2 * N);
    std::copy(ainv_ntt.begin(), ainv_ntt.end(), out.begin());
    std::copy(b.begin(), b.end(), out.begin() + m * 2 * N);
    return out;
}

// getPublicEncryptions (src/client.cpp:270-293): W_exp_i = Enc_s0( tau_i(s0) * G_exp ), tau_i: x -> x^(N/2^i + 1)
Poly Client::public_encryptions(uint32_t count, uint32_t t_dim) {
    Poly g_ntt = to_ntt(build_gadget(1, t_dim));
    Poly out;
    out.reserve((size_t)count * 2 * t_dim * 2 * N);
    for (uint32_t i = 0; i < count; i++) {
        uint64_t t = (N >> i) + 1;
        Poly mat = mul_by_const(to_ntt(automorph(sr, t)), g_ntt);  // 1 x t_dim
        Poly enc = regev_samples(t_dim);                            // encryptSimpleRegevMatrix, src/client.cpp:214-233
        Poly row1(enc.begin() + (size_t)t_dim * 2 * N, enc.end());
        row1 = add(row1, mat);
        std::copy(row1.begin(), row1.end(), enc.begin() + (size_t)t_dim * 2 * N);
        out.insert(out.end(), enc.begin(), enc.end());
    }
    return out;
}

void Client::gen_pub_params() {
    const uint32_t tc = p.t_conv;
    offline_bytes = 0;
    auto account = [&](size_t rows, size_t cols, size_t count) { offline_bytes += (uint64_t)count * rows * cols * N * 56 / 8; };  // add_pub_param :199
    if (s.n_right) {  // src/spiral.cpp:2091-2092
        w_right = public_encryptions(s.n_right, p.t_exp_right);
        account(2, p.t_exp_right, s.n_right);
    }
    if (s.n_left) {
        w_left = public_encryptions(s.n_left, p.t_exp);
        account(2, p.t_exp, s.n_left);
    }
    Poly s0_ntt = to_ntt(sr);
    {  // W = P + pad(s0 * G_scale) (src/spiral.cpp:2205-2219)
        size_t m = 2 * (size_t)tc;
        Poly s0g = mul_by_const(s0_ntt, to_ntt(build_gadget(2, m)));
        w = fresh_public_key(m);
        Poly rows(w.begin() + m * 2 * N, w.end());
        rows = add(rows, s0g);
        std::copy(rows.begin(), rows.end(), w.begin() + m * 2 * N);
        account(3, m, 1);
    }
    {  // V = P + pad(Sp * [s0*gv | gv]) (src/spiral.cpp:2279-2296)
        size_t m = 2 * (size_t)tc;
        Poly gv = to_ntt(build_gadget(1, tc));
        Poly scaled = mul_by_const(s0_ntt, gv);
        Poly together = scaled;
        together.insert(together.end(), gv.begin(), gv.end());
        Poly res = multiply(to_ntt(sp), together, 2, 1, m);
        v = fresh_public_key(m);
        Poly rows(v.begin() + m * 2 * N, v.end());
        rows = add(rows, res);
        std::copy(rows.begin(), rows.end(), v.begin() + m * 2 * N);
        if (!p.direct_upload) account(3, m, 1);
    }
    if (w_left.empty()) w_left.assign(1, 0);
    if (w_right.empty()) w_right.assign(1, 0);
}

Poly Client::encrypt_simple_regev(const Poly& sigma_raw) {  // src/client.cpp:176-192
    Poly c = regev_samples(1);
    Poly row1(c.begin() + 2 * N, c.end());
    row1 = add(row1, to_ntt(sigma_raw));
    std::copy(row1.begin(), row1.end(), c.begin() + 2 * N);
    return c;
}

Poly Client::query(uint64_t idx_target) {
    const uint64_t idx_dim0 = idx_target / s.num_per, idx_further = idx_target % s.num_per;
    const uint64_t scale_k = Q / p.p_db;  // include/values.h:93
    const uint32_t bits = bits_per(s.ell);
    Poly out;
    if (p.direct_upload) {  // src/spiral.cpp:2177-2188, 2298-2310
        for (uint32_t i = 0; i < s.dim0; i++) {
            Poly sigma(N, 0);
            if (i == idx_dim0) sigma[0] = scale_k % Q;
            Poly c = encrypt_simple_regev(sigma);
            out.insert(out.end(), c.begin(), c.end());
        }
        for (uint32_t i = 0; i < p.nu2; i++) {
            uint64_t bit = (idx_further >> i) & 1;
            for (uint32_t j = 0; j < s.ell; j++) {
                Poly sigma(N, 0);
                sigma[0] = bit ? (1ull << (j * bits)) : 0;
                Poly c = encrypt_simple_regev(sigma);
                out.insert(out.end(), c.begin(), c.end());
            }
        }
        return out;
    }
    Poly sigma(N, 0);
    if (s.stopround != 0) {  // src/spiral.cpp:2104-2116, 2141-2147
        sigma[2 * idx_dim0] = scale_k % Q;
        for (uint32_t i = 0; i < p.nu2; i++) {
            uint64_t bit = (idx_further >> i) & 1;
            for (uint32_t j = 0; j < s.ell; j++) sigma[2 * (i * s.ell + j) + 1] = ((1ull << (bits * j)) * bit) % Q;
        }
        uint64_t inv_first = inv_mod_q(1ull << s.g), inv_rest = inv_mod_q(1ull << (s.stopround + 1));
        for (uint32_t i = 0; i < N / 2; i++) {
            sigma[2 * i] = (uint64_t)((u128)sigma[2 * i] * inv_first % Q);
            sigma[2 * i + 1] = (uint64_t)((u128)sigma[2 * i + 1] * inv_rest % Q);
        }
    } else {  // src/spiral.cpp:2117-2140, 2148-2152
        sigma[idx_dim0] = scale_k % Q;
        uint32_t ctr = 0;
        for (uint32_t i = 0; i < p.nu2; i++) {
            uint64_t bit = (idx_further >> i) & 1;
            for (uint32_t j = 0; j < s.ell; j++) sigma[s.dim0 + ctr++] = ((1ull << (bits * j)) * bit) % Q;
        }
        uint64_t inv = inv_mod_q(1ull << s.g);
        for (auto& x : sigma) x = (uint64_t)((u128)x * inv % Q);
    }
    return encrypt_simple_regev(sigma);
}

// check_final's client half (src/spiral.cpp:1451-1491)
Poly Client::decode(const uint64_t* resp) const {
    const uint64_t qp = s.qprime, p_db = p.p_db, q1 = 4 * p_db;
    Poly spq(2 * N), out(4 * N), prod(N);
    for (size_t i = 0; i < 2 * (size_t)N; i++) {  // to_ntt_qprime's centring (src/util.cpp:218-223)
        __int128 a = (__int128)sp[i];
        if (a >= (__int128)(Q / 2)) a -= Q;
        spq[i] = (uint64_t)((a + (__int128)((Q / qp) * qp) + (__int128)(2 * qp)) % (__int128)qp);
    }
    for (uint32_t r = 0; r < 2; r++)
        for (uint32_t col = 0; col < 2; col++) {
            std::fill(prod.begin(), prod.end(), 0);
            const uint64_t* a = &spq[(size_t)r * N];
            const uint64_t* b = resp + (size_t)col * N;
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
                int64_t vr = (int64_t)resp[((size_t)(1 + r) * 2 + col) * N + z];
                if (vr >= (int64_t)(q1 / 2)) vr -= (int64_t)q1;
                uint64_t denom = qp * (q1 / p_db);
                int64_t rr = vf * (int64_t)q1 + vr * (int64_t)qp;
                int64_t sign = rr >= 0 ? 1 : -1;
                __int128 res = ((__int128)rr + sign * (int64_t)(denom / 2)) / (__int128)denom;
                res = (res + (__int128)((denom / p_db) * p_db) + (__int128)(2 * p_db)) % (__int128)p_db;
                out[((size_t)r * 2 + col) * N + z] = (uint64_t)res;
            }
        }
    return out;
}

// =====================================================================================================
// SpiralPack client
// =====================================================================================================
namespace {
Poly from_ntt(const Poly& a) {
    Poly out(a.size() / 2);
    ok(spiral_gpu_from_ntt(out.data(), a.data(), a.size() / (2 * N)), "from_ntt");
    return out;
}
Poly const_poly(uint64_t v) {
    Poly p(N, 0);
    p[0] = v;
    return p;
}
}  // namespace

Poly pack_db_item(uint64_t seed, uint64_t item, uint64_t total_n, uint32_t out_n, uint64_t p_db) {
    Poly pt((size_t)out_n * out_n * N);
    for (uint32_t t = 0; t < out_n * out_n; t++)
        for (uint32_t z = 0; z < N; z++) pt[(size_t)t * N + z] = splitmix64(seed ^ (((uint64_t)t * total_n + item) * N + z)) % p_db;
    return pt;
}

PackClient::PackClient(const spiral_gpu_params& params, uint32_t out_n_, uint64_t seed, bool nonoise_)
    : p(params), out_n(out_n_), nonoise(nonoise_), rng(seed) {
    ok(spiral_gpu_pack_get_shape(&p, out_n, &s), "pack_get_shape");
    double acc = 0;
    for (int i = -64; i <= 64; i++) {
        acc += std::exp(-M_PI * (double)i * i / (6.4 * 6.4));
        cdf.push_back(acc);
    }
}
uint64_t PackClient::sample_noise() {
    if (nonoise) return 0;
    double u = std::uniform_real_distribution<double>(0.0, cdf.back())(rng);
    int64_t v = (int64_t)(std::lower_bound(cdf.begin(), cdf.end(), u) - cdf.begin()) - 64;
    if (v > 64) v = 64;
    return (uint64_t)((v + (int64_t)Q) % (int64_t)Q);
}
Poly PackClient::noise_polys(size_t n) {
    Poly a(n * N);
    for (auto& x : a) x = sample_noise();
    return a;
}
Poly PackClient::uniform_polys(size_t n) {
    Poly a(n * N);
    for (auto& x : a) x = rng() % Q;
    return a;
}
void PackClient::keygen() {  // keygen(S, Sp, sr, out_n), src/testing.cpp:907-910
    sr = noise_polys(1);
    sp = noise_polys(out_n);
}
Poly PackClient::regev_samples(size_t m) {
    Poly a = uniform_polys(m), e = noise_polys(m);
    Poly b = add(mul_by_const(to_ntt(sr), to_ntt(a)), to_ntt(e));
    Poly out = to_ntt(invert(a));
    out.insert(out.end(), b.begin(), b.end());
    return out;
}
Poly PackClient::encrypt_simple_regev(const Poly& sigma_raw) {
    Poly c = regev_samples(1);
    Poly row1(c.begin() + 2 * N, c.end());
    row1 = add(row1, to_ntt(sigma_raw));
    std::copy(row1.begin(), row1.end(), c.begin() + 2 * N);
    return c;
}
Poly PackClient::expansion_keys(uint32_t count, uint32_t t_dim) {  // getExpansionKeySwitchingMatrices, src/testing.cpp:21-38
    Poly g_ntt = to_ntt(build_gadget(1, t_dim)), out;
    for (uint32_t i = 0; i < count; i++) {
        Poly mat = mul_by_const(to_ntt(automorph(sr, (N >> i) + 1)), g_ntt);
        Poly enc = regev_samples(t_dim);
        Poly row1(enc.begin() + (size_t)t_dim * 2 * N, enc.end());
        row1 = add(row1, mat);
        std::copy(row1.begin(), row1.end(), enc.begin() + (size_t)t_dim * 2 * N);
        out.insert(out.end(), enc.begin(), enc.end());
    }
    return out;
}
void PackClient::gen_pub_params() {
    const uint32_t tc = p.t_conv, rows = out_n + 1;
    offline_bytes = 0;
    auto account = [&](size_t r, size_t c, size_t count) { offline_bytes += (uint64_t)count * r * c * N * 56 / 8; };
    Poly s0_ntt = to_ntt(sr), sp_ntt = to_ntt(sp);
    Poly s0g = mul_by_const(s0_ntt, to_ntt(build_gadget(1, tc)));  // 1 x t_conv
    v_w.clear();
    for (uint32_t i = 0; i < out_n; i++) {  // v_W[i] = encryptMatrixArbitrary(AG_i), src/testing.cpp:918-925
        Poly a = uniform_polys(tc), e = noise_polys((size_t)out_n * tc);
        Poly a_ntt = to_ntt(a);
        Poly b = add(to_ntt(e), multiply(sp_ntt, a_ntt, out_n, 1, tc));  // out_n x t_conv
        Poly row(b.begin() + (size_t)i * tc * 2 * N, b.begin() + (size_t)(i + 1) * tc * 2 * N);
        row = add(row, s0g);
        std::copy(row.begin(), row.end(), b.begin() + (size_t)i * tc * 2 * N);
        Poly w = to_ntt(invert(a));
        w.insert(w.end(), b.begin(), b.end());
        v_w.insert(v_w.end(), w.begin(), w.end());
    }
    account(rows, tc, out_n);
    if (!p.direct_upload) {  // src/testing.cpp:926-949
        w_left = expansion_keys(s.n_left, p.t_exp);
        w_right = expansion_keys(s.n_right, p.t_exp_right);
        Poly s0sq = multiply(s0_ntt, s0_ntt, 1, 1, 1);
        const uint32_t bits = bits_per(tc), cols = 2 * tc;
        v.assign((size_t)2 * cols * 2 * N, 0);
        for (uint32_t i = 0; i < cols; i++) {
            uint64_t sh = (uint64_t)bits * (i / 2);
            Poly val_ntt = to_ntt(const_poly(sh >= 64 ? 0 : (1ull << sh)));
            Poly sigma = from_ntt(multiply((i % 2 == 0) ? s0sq : s0_ntt, val_ntt, 1, 1, 1));
            Poly ct = encrypt_simple_regev(sigma);
            for (uint32_t r = 0; r < 2; r++) std::copy(ct.begin() + (size_t)r * 2 * N, ct.begin() + (size_t)(r + 1) * 2 * N, v.begin() + ((size_t)r * cols + i) * 2 * N);
        }
        account(2, p.t_exp, s.n_left);
        account(2, p.t_exp_right, s.n_right);
        account(2, cols, 1);
    }
    if (w_left.empty()) w_left.assign(1, 0);
    if (w_right.empty()) w_right.assign(1, 0);
    if (v.empty()) v.assign(1, 0);
}
Poly PackClient::query(uint64_t idx_target) {
    const uint64_t idx_dim0 = idx_target / s.num_per, idx_further = idx_target % s.num_per, scale_k = Q / p.p_db;
    const uint32_t bits = bits_per(s.ell);
    Poly out;
    if (p.direct_upload) {  // src/testing.cpp:966-989
        for (uint32_t i = 0; i < s.dim0; i++) {
            Poly c = encrypt_simple_regev(const_poly(i == idx_dim0 ? scale_k : 0));
            out.insert(out.end(), c.begin(), c.end());
        }
        Poly s0_ntt = to_ntt(sr);
        for (uint32_t i = 0; i < p.nu2; i++) {
            const uint64_t bit = (idx_further >> i) & 1;
            for (uint32_t j = 0; j < s.ell; j++) {
                Poly val = const_poly((1ull << (bits * j)) * bit);
                Poly c_odd = encrypt_simple_regev(val);                                                      // column 2j+1
                Poly c_even = encrypt_simple_regev(from_ntt(multiply(s0_ntt, to_ntt(val), 1, 1, 1)));        // column 2j
                out.insert(out.end(), c_even.begin(), c_even.end());
                out.insert(out.end(), c_odd.begin(), c_odd.end());
            }
        }
        return out;
    }
    Poly sigma(N, 0);  // src/testing.cpp:991-1006
    sigma[2 * idx_dim0] = scale_k;
    for (uint32_t i = 0; i < p.nu2; i++) {
        const uint64_t bit = (idx_further >> i) & 1;
        for (uint32_t j = 0; j < s.ell; j++) sigma[2 * (i * s.ell + j) + 1] = (1ull << (bits * j)) * bit;
    }
    const uint64_t inv_first = inv_mod_q(1ull << s.g), inv_rest = inv_mod_q(1ull << (s.stopround + 1));
    for (uint32_t i = 0; i < N / 2; i++) {
        sigma[2 * i] = (uint64_t)((u128)sigma[2 * i] * inv_first % Q);
        sigma[2 * i + 1] = (uint64_t)((u128)sigma[2 * i + 1] * inv_rest % Q);
    }
    return encrypt_simple_regev(sigma);
}
Poly PackClient::decode(const uint64_t* resp) const {  // src/testing.cpp:1086-1122
    const uint64_t qp = s.qprime, p_db = p.p_db, q1 = 4 * p_db;
    Poly spq((size_t)out_n * N), out((size_t)out_n * out_n * N), prod(N);
    for (size_t i = 0; i < spq.size(); i++) {
        __int128 a = (__int128)sp[i];
        if (a >= (__int128)(Q / 2)) a -= Q;
        spq[i] = (uint64_t)((a + (__int128)((Q / qp) * qp) + (__int128)(2 * qp)) % (__int128)qp);
    }
    for (uint32_t r = 0; r < out_n; r++)
        for (uint32_t col = 0; col < out_n; col++) {
            std::fill(prod.begin(), prod.end(), 0);
            const uint64_t* a = &spq[(size_t)r * N];
            const uint64_t* b = resp + (size_t)col * N;
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
                out[((size_t)r * out_n + col) * N + z] = (uint64_t)res;
            }
        }
    return out;
}

}  // namespace spiral_cli
