// Census of resident workgroups per CU as a function of the LDS a 256-thread workgroup declares:
// every workgroup records (XCC id, HW id, start, end) around a fixed spin; the host counts the largest number of
// workgroups whose intervals overlap on one CU.   hipcc --offload-arch=gfx950 -O3 tools/occupancy_probe.hip -o tools/occupancy_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <map>
#include <vector>

struct Rec { uint32_t xcc, hwid; uint64_t t0, t1; };

__global__ __launch_bounds__(256) void probe(Rec* out, uint32_t spin, uint32_t vgpr_pad) {
    extern __shared__ uint32_t lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        const uint64_t t0 = wall_clock64();
        uint64_t t = t0;
        while (t - t0 < spin) t = wall_clock64();
        out[blockIdx.x] = Rec{xcc & 0xf, hw, t0, t};
    }
    __syncthreads();
    if (lds[(threadIdx.x + 1) & 255] == 12345 + vgpr_pad) out[0].xcc = 99;
}

int main() {
    const int blocks = 256 * 16;
    Rec* d;
    hipMalloc(&d, blocks * sizeof(Rec));
    std::vector<Rec> h(blocks);
    for (uint32_t lds_kib : {1u, 8u, 16u, 18u, 20u, 24u, 32u, 40u, 64u}) {
        hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), lds_kib * 1024, 0, d, 2000u /* 100 MHz ticks = 20 us */, 0u);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, blocks * sizeof(Rec), hipMemcpyDeviceToHost);
        std::map<uint64_t, std::vector<std::pair<uint64_t, int>>> ev;  // per CU: (time, +1/-1)
        for (auto& r : h) {
            const uint64_t cu = ((uint64_t)r.xcc << 32) | (r.hwid & 0xff00u) | ((r.hwid >> 13) & 0x7u);  // cu_id, sh_id, se_id
            ev[cu].push_back({r.t0, +1});
            ev[cu].push_back({r.t1, -1});
        }
        int worst = 0, best = 1 << 30;
        for (auto& kv : ev) {
            auto& v = kv.second;
            std::sort(v.begin(), v.end());
            int cur = 0, mx = 0;
            for (auto& e : v) { cur += e.second; mx = std::max(mx, cur); }
            worst = std::max(worst, mx);
            best = std::min(best, mx);
        }
        printf("LDS %3u KiB per 256-thread workgroup: %zu distinct CUs seen, resident workgroups per CU: max %d, min %d\n", lds_kib, ev.size(), worst, best);
    }
    return 0;
}
