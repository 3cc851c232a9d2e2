// Twiddle tables for the two CRT primes, regenerated from the minimal primitive 4096-th roots of unity
// (psi_p = 66687, psi_b = 158221) instead of copying the reference's 436 KB blob (src/constants.cpp:16).
// Layout and scaling of the host rows follow src/core.cpp:6-17: fwd[bitrev11(i)] = psi^i, inv[bitrev11(i)] = psi^-i / 2,
// scaled companion W' = floor(W * 2^32 / m).  tests/test_oracle_tables.py + test_capi_cpu.py pin the
// result to the reference's data via tests/golden/ntt_tables.json.
#include <mutex>
#include <vector>

#include "common.h"
#include "kernels.h"

namespace spiral {

static uint64_t powmod(uint64_t b, uint64_t e, uint64_t m) {
    unsigned __int128 r = 1, x = b % m;
    while (e) {
        if (e & 1) r = r * x % m;
        x = x * x % m;
        e >>= 1;
    }
    return (uint64_t)r;
}
static uint32_t bitrev11(uint32_t x) {
    uint32_t r = 0;
    for (uint32_t i = 0; i < kLogN; i++) r |= ((x >> i) & 1u) << (kLogN - 1 - i);
    return r;
}

// rows in the reference's order: [inv p W][inv p W'][inv b W][inv b W'][fwd p W][fwd p W'][fwd b W][fwd b W']
void tables_host_rows(uint64_t* out) {
    const uint64_t mods[2] = {kP, kB}, psis[2] = {66687, 158221};
    for (int n = 0; n < 2; n++) {
        uint64_t m = mods[n], psi = psis[n], ipsi = powmod(psi, 2 * kN - 1, m);
        uint64_t f = 1, v = (m + 1) / 2;
        for (uint32_t i = 0; i < kN; i++) {
            uint32_t k = bitrev11(i);
            out[(4 + 2 * n) * kN + k] = f;
            out[(5 + 2 * n) * kN + k] = (f << 32) / m;
            out[(0 + 2 * n) * kN + k] = v;
            out[(1 + 2 * n) * kN + k] = (v << 32) / m;
            f = (uint64_t)((unsigned __int128)f * psi % m);
            v = (uint64_t)((unsigned __int128)v * ipsi % m);
        }
    }
}

static std::mutex g_mu;
static DeviceTables g_tables[16];
static bool g_ready[16];

int tables_get(int device, DeviceTables* out) {
    if (device < 0 || device >= 16) return -1;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_ready[device]) {
        std::vector<uint64_t> rows(8 * kN);
        tables_host_rows(rows.data());
        std::vector<uint4> fwd(kN), inv(kN);
        for (uint32_t i = 0; i < kN; i++) {
            fwd[i] = make_uint4((uint32_t)rows[4 * kN + i], (uint32_t)rows[5 * kN + i], (uint32_t)rows[6 * kN + i], (uint32_t)rows[7 * kN + i]);
        }
        // Device inverse table (ntt_device.h ntt_inverse_block): the reference's rows carry the 1/2 of its per-stage halving
        // (src/core.cpp:6-17, 445-472); the kernels run the stages unscaled and apply N^-1 once, in the last stage, so here
        // rows >= 2 are 2 * (reference row) = psi^-i, row 1 (the last stage's only twiddle) is psi^-(N/2) * N^-1 and the
        // unused row 0 holds N^-1 itself.  spiral_gpu_get_tables still returns the reference's rows (tables_host_rows).
        {
            const uint64_t mods[2] = {kP, kB};
            uint32_t w[2][kN];
            for (int n = 0; n < 2; n++) {
                const uint64_t m = mods[n], ninv = powmod(kN, m - 2, m);
                for (uint32_t i = 0; i < kN; i++) w[n][i] = (uint32_t)(rows[(0 + 2 * n) * kN + i] * 2 % m);
                w[n][1] = (uint32_t)((unsigned __int128)w[n][1] * ninv % m);
                w[n][0] = (uint32_t)ninv;
            }
            for (uint32_t i = 0; i < kN; i++)
                inv[i] = make_uint4(w[0][i], (uint32_t)(((uint64_t)w[0][i] << 32) / kP), w[1][i], (uint32_t)(((uint64_t)w[1][i] << 32) / kB));
        }
        DeviceTables t;
        if (hipSetDevice(device) != hipSuccess) return -1;
        if (hipMalloc(&t.fwd, kN * sizeof(uint4)) != hipSuccess) return -1;
        if (hipMalloc(&t.inv, kN * sizeof(uint4)) != hipSuccess) return -1;
        if (hipMalloc(&t.neg1, (size_t)kLogN * kN * sizeof(uint64_t)) != hipSuccess) return -1;
        if (hipMemcpy(t.fwd, fwd.data(), kN * sizeof(uint4), hipMemcpyHostToDevice) != hipSuccess) return -1;
        if (hipMemcpy(t.inv, inv.data(), kN * sizeof(uint4), hipMemcpyHostToDevice) != hipSuccess) return -1;
        // neg1[r] = NTT(-x^(N - 2^r)) (src/spiral.cpp:171-190): raw polynomial with Q-1 at N-2^r
        std::vector<uint64_t> raw((size_t)kLogN * kN, 0);
        for (uint32_t r = 0; r < kLogN; r++) raw[(size_t)r * kN + (kN - (1u << r))] = kQ - 1;
        uint64_t* d_raw = nullptr;
        if (hipMalloc(&d_raw, raw.size() * sizeof(uint64_t)) != hipSuccess) return -1;
        if (hipMemcpy(d_raw, raw.data(), raw.size() * sizeof(uint64_t), hipMemcpyHostToDevice) != hipSuccess) return -1;
        FwdParams p{};
        p.src = d_raw;
        p.dst = t.neg1;
        p.src_map = identity_map();
        p.dst_map = identity_map();
        p.n_digits = 1;
        launch_ntt_forward(t, p, LD_RAW, ST_PK, kLogN, 0);
        if (hipDeviceSynchronize() != hipSuccess) return -1;
        (void)hipFree(d_raw);
        // Shoup companions floor(w * 2^32 / m) of the same words, so that the expansion multiplies by neg1 without a division
        std::vector<uint64_t> w((size_t)kLogN * kN);
        if (hipMemcpy(w.data(), t.neg1, w.size() * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess) return -1;
        for (uint64_t& x : w) x = (((x & 0xffffffffull) << 32) / kP) | ((((x >> 32) << 32) / kB) << 32);
        if (hipMalloc(&t.neg1s, w.size() * sizeof(uint64_t)) != hipSuccess) return -1;
        if (hipMemcpy(t.neg1s, w.data(), w.size() * sizeof(uint64_t), hipMemcpyHostToDevice) != hipSuccess) return -1;
        g_tables[device] = t;
        g_ready[device] = true;
    }
    *out = g_tables[device];
    return 0;
}

}  // namespace spiral
