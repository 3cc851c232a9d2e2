// Per-job hand-offs for the narrow phases of the fold: rounds 4-6 + the final lift + the switch are 11 dependent launches
// (~55 us) for < 400 transforms, and the dependencies are per ciphertext pair, not all-to-all:
//     lift(L), lift(H)  ->  ell digit-difference jobs per polynomial  ->  product of the pair (32 slot blocks, each needs the pair's 6 ell digit
//     polynomials)  ->  the next round's lift of that ciphertext (needs its 32 product blocks)
// This probe runs that job graph with the real fan-ins and payload sizes (16 KiB polynomials, 512-byte slot slices) two ways: (a) as ONE persistent
// launch, every job a resident 256-thread workgroup that waits on its producers' counter (agent-scope release -> relaxed counter -> agent-scope
// acquire by one lane + __syncthreads, MI355X_MICROARCH.md "handoff-flag"), (b) as 11 dependent launches replayed as a hipGraph.  A job's arithmetic
// is a stand-in (a spin of `spin` dependent multiply-adds per value: ~the 2-3 us a transform workgroup computes) so that what is compared is the
// hand-off against the kernel boundary; both variants must produce identical words (a stale read shows up as a mismatch).
// hipcc --offload-arch=gfx950 -O3 tools/handoff_probe.hip -o tools/handoff_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned long long u64;
constexpr int N = 2048, ELL = 8, POLYS = 6, SLOTB = 32;  // 32 slot blocks of 64
enum { LIFT = 0, DIGIT = 1, PROD = 2, SWITCH = 3 };
struct Job {
    int type, stage;
    const u64* in0;   // LIFT: polynomial; DIGIT: L polynomial; PROD: the pair's digit polynomials [48][N]; SWITCH: polynomial
    const u64* in1;   // DIGIT: H polynomial
    u64* out;
    int k;            // DIGIT: digit; PROD / SWITCH: slot block
    unsigned* wait;   // counter to wait on (null: ready) and the count that means "complete"
    unsigned need;
    unsigned* done;   // counter to bump when this job's output is visible
};
__device__ __forceinline__ u64 work(u64 x, int spin) {
    for (int i = 0; i < spin; i++) x = x * 6364136223846793005ull + 1442695040888963407ull;
    return x;
}
// SC1: payload through agent-scope relaxed atomics (global_load / global_store ... sc1: write-through, no stale L2 lines), the guide's
// alternative to release / acquire fences, which write back and invalidate the whole L2 per hand-off
template <bool SC1> __device__ __forceinline__ u64 ld(const u64* p) { return SC1 ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p; }
template <bool SC1> __device__ __forceinline__ void st(u64* p, u64 v) { if (SC1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v; }
template <bool SC1>
__device__ void run_job(const Job& j, int spin) {
    const int t = threadIdx.x;
    if (j.type == LIFT) {
        u64 x[8];
        for (int r = 0; r < 8; r++) x[r] = ld<SC1>(&j.in0[t + 256 * r]);
        for (int r = 0; r < 8; r++) st<SC1>(&j.out[t + 256 * r], work(x[r], spin) >> 8);
    } else if (j.type == DIGIT) {
        u64 l[8], h[8];
        for (int r = 0; r < 8; r++) { l[r] = ld<SC1>(&j.in0[t + 256 * r]); h[r] = ld<SC1>(&j.in1[t + 256 * r]); }
        for (int r = 0; r < 8; r++) st<SC1>(&j.out[t + 256 * r], work(((h[r] >> (7 * j.k)) & 255) - ((l[r] >> (7 * j.k)) & 255), spin));
    } else if (j.type == PROD) {  // 64 slots x 4 k-groups, every digit polynomial of the pair contributes
        __shared__ u64 sh[4][64][POLYS];
        const int z = j.k * 64 + (t & 63), kg = t >> 6;
        u64 acc[POLYS] = {};
        for (int m = kg; m < POLYS * ELL; m += 4) {
            const u64 d = ld<SC1>(&j.in0[(size_t)m * N + z]);
            for (int o = 0; o < POLYS; o++) acc[o] += d * (u64)(2 * o + 3 + m);
        }
        for (int o = 0; o < POLYS; o++) sh[kg][t & 63][o] = acc[o];
        __syncthreads();
        if (kg == 0)
            for (int o = 0; o < POLYS; o++) st<SC1>(&j.out[(size_t)o * N + z], work(sh[0][t][o] + sh[1][t][o] + sh[2][t][o] + sh[3][t][o], spin / 4));
    } else {
        const int z = j.k * 256 + t;
        st<SC1>(&j.out[z], work(ld<SC1>(&j.in0[z]), spin / 4) % 1048573ull);
    }
}
template <bool SC1>
__global__ __launch_bounds__(256) void k_persistent(const Job* jobs, int spin) {
    const Job j = jobs[blockIdx.x];
    if (j.wait) {
        if (threadIdx.x == 0) {
            while (__hip_atomic_load(j.wait, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < j.need) __builtin_amdgcn_s_sleep(1);
            if (!SC1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
    run_job<SC1>(j, spin);
    if (j.done) {
        if (SC1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through stores have been acknowledged
        __syncthreads();  // (!SC1: workgroup-scope release of every wave's stores: vmcnt(0) before the barrier)
        if (threadIdx.x == 0) {
            if (!SC1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(j.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
__global__ __launch_bounds__(256) void k_stage(const Job* jobs, int first, int spin) { run_job<false>(jobs[first + blockIdx.x], spin); }

int main(int argc, char** argv) {
    const int spin = argc > 1 ? atoi(argv[1]) : 300;
    const int np0 = argc > 2 ? atoi(argv[2]) : 4;  // pairs of the first round: 4 -> rounds np = 4, 2, 1 (fold rounds 4-6 of config 2)
    // buffers: cts[r] = the 2 np_r ciphertexts entering round r (PK words), raw[r] their lifts, dig[r] digit polynomials per pair
    std::vector<Job> jobs;
    std::vector<int> stage_first;
    std::vector<u64*> bufs;
    auto alloc = [&](size_t words) { u64* p = nullptr; if (hipMalloc(&p, words * 8) != hipSuccess) exit(1); hipMemset(p, 0, words * 8); bufs.push_back(p); return p; };
    unsigned* counters = nullptr;
    OK(hipMalloc(&counters, 4096 * sizeof(unsigned)));
    OK(hipMemset(counters, 0, 4096 * sizeof(unsigned)));
    int nc = 0;
    auto counter = [&]() { return counters + (nc++); };
    u64* cts = alloc((size_t)2 * np0 * POLYS * N);
    {
        std::vector<u64> h((size_t)2 * np0 * POLYS * N);
        for (size_t i = 0; i < h.size(); i++) h[i] = i * 2654435761ull + 12345;
        OK(hipMemcpy(cts, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    }
    std::vector<unsigned*> ct_ready(2 * np0, nullptr);  // counter that reaches SLOTB when ciphertext c of the current round is complete (null: input)
    int stage = 0;
    for (int np = np0; np >= 1; np /= 2) {
        u64* raw = alloc((size_t)2 * np * POLYS * N);
        u64* dig = alloc((size_t)np * POLYS * ELL * N);
        u64* out = alloc((size_t)np * POLYS * N);
        std::vector<unsigned*> lifted(2 * np * POLYS), digits(np), next_ready(np);
        stage_first.push_back((int)jobs.size());
        for (int c = 0; c < 2 * np; c++)
            for (int rc = 0; rc < POLYS; rc++) {
                lifted[c * POLYS + rc] = counter();
                jobs.push_back(Job{LIFT, stage, cts + ((size_t)c * POLYS + rc) * N, nullptr, raw + ((size_t)c * POLYS + rc) * N, 0, ct_ready[c], SLOTB, lifted[c * POLYS + rc]});
            }
        stage++;
        stage_first.push_back((int)jobs.size());
        // a digit job waits on TWO lifts: one counter per (pair, polynomial) that both lifts bump
        std::vector<unsigned*> pair_lift(np * POLYS);
        for (int i = 0; i < np * POLYS; i++) pair_lift[i] = counter();
        for (size_t q = jobs.size() - (size_t)2 * np * POLYS; q < jobs.size(); q++) {
            const int c = (int)(q - (jobs.size() - (size_t)2 * np * POLYS)) / POLYS, rc = (int)(q - (jobs.size() - (size_t)2 * np * POLYS)) % POLYS;
            jobs[q].done = pair_lift[(c % np) * POLYS + rc];
        }
        for (int i = 0; i < np; i++) {
            digits[i] = counter();
            for (int rc = 0; rc < POLYS; rc++)
                for (int k = 0; k < ELL; k++)
                    jobs.push_back(Job{DIGIT, stage, raw + ((size_t)i * POLYS + rc) * N, raw + ((size_t)(np + i) * POLYS + rc) * N, dig + (((size_t)i * POLYS + rc) * ELL + k) * N, k,
                                       pair_lift[i * POLYS + rc], 2, digits[i]});
        }
        stage++;
        stage_first.push_back((int)jobs.size());
        for (int i = 0; i < np; i++) {
            next_ready[i] = counter();
            for (int z = 0; z < SLOTB; z++)
                jobs.push_back(Job{PROD, stage, dig + (size_t)i * POLYS * ELL * N, nullptr, out + (size_t)i * POLYS * N, z, digits[i], (unsigned)(POLYS * ELL), next_ready[i]});
        }
        stage++;
        cts = out;
        ct_ready.assign(next_ready.begin(), next_ready.end());
    }
    u64* raw = alloc((size_t)POLYS * N);
    u64* resp = alloc((size_t)POLYS * N);
    std::vector<unsigned*> lifted(POLYS);
    stage_first.push_back((int)jobs.size());
    for (int rc = 0; rc < POLYS; rc++) {
        lifted[rc] = counter();
        jobs.push_back(Job{LIFT, stage, cts + (size_t)rc * N, nullptr, raw + (size_t)rc * N, 0, ct_ready[0], SLOTB, lifted[rc]});
    }
    stage++;
    stage_first.push_back((int)jobs.size());
    for (int rc = 0; rc < POLYS; rc++)
        for (int z = 0; z < 8; z++) jobs.push_back(Job{SWITCH, stage, raw + (size_t)rc * N, nullptr, resp + (size_t)rc * N, z, lifted[rc], 1, nullptr});
    stage++;
    stage_first.push_back((int)jobs.size());
    Job* d_jobs;
    OK(hipMalloc(&d_jobs, jobs.size() * sizeof(Job)));
    OK(hipMemcpy(d_jobs, jobs.data(), jobs.size() * sizeof(Job), hipMemcpyHostToDevice));
    printf("job graph: %d stages, %zu jobs (workgroups), %d counters; spin %d\n", stage, jobs.size(), nc, spin);
    hipStream_t s;
    OK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    OK(hipEventCreate(&e0));
    OK(hipEventCreate(&e1));
    std::vector<u64> r_launch((size_t)POLYS * N), r_pers((size_t)POLYS * N);
    // (b) dependent launches, as a graph
    hipGraph_t g;
    hipGraphExec_t ge;
    OK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
    for (int st = 0; st < stage; st++) hipLaunchKernelGGL(k_stage, dim3(stage_first[st + 1] - stage_first[st]), dim3(256), 0, s, d_jobs, stage_first[st], spin);
    OK(hipStreamEndCapture(s, &g));
    OK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    const int reps = 50;
    float ms_l = 0;
    for (int w = 0; w < 5; w++) OK(hipGraphLaunch(ge, s));
    OK(hipEventRecord(e0, s));
    for (int w = 0; w < reps; w++) OK(hipGraphLaunch(ge, s));
    OK(hipEventRecord(e1, s));
    OK(hipEventSynchronize(e1));
    OK(hipEventElapsedTime(&ms_l, e0, e1));
    OK(hipMemcpy(r_launch.data(), resp, r_launch.size() * 8, hipMemcpyDeviceToHost));
    OK(hipMemset(resp, 0, r_launch.size() * 8));
    // (a) one persistent launch; counters reset by a memset node between replays.  Twice: fences, then sc1 payload
    float ms_v[2] = {0, 0};
    size_t bad_v[2] = {0, 0};
    for (int v = 0; v < 2; v++) {
        OK(hipMemset(resp, 0, r_launch.size() * 8));
        hipGraph_t g2;
        hipGraphExec_t ge2;
        OK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
        OK(hipMemsetAsync(counters, 0, 4096 * sizeof(unsigned), s));
        if (v == 0)
            hipLaunchKernelGGL(k_persistent<false>, dim3((unsigned)jobs.size()), dim3(256), 0, s, d_jobs, spin);
        else
            hipLaunchKernelGGL(k_persistent<true>, dim3((unsigned)jobs.size()), dim3(256), 0, s, d_jobs, spin);
        OK(hipStreamEndCapture(s, &g2));
        OK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
        for (int w = 0; w < 5; w++) OK(hipGraphLaunch(ge2, s));
        OK(hipEventRecord(e0, s));
        for (int w = 0; w < reps; w++) OK(hipGraphLaunch(ge2, s));
        OK(hipEventRecord(e1, s));
        OK(hipEventSynchronize(e1));
        OK(hipEventElapsedTime(&ms_v[v], e0, e1));
        OK(hipMemcpy(r_pers.data(), resp, r_pers.size() * 8, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < r_pers.size(); i++) bad_v[v] += r_pers[i] != r_launch[i];
    }
    // the memset node alone, to subtract
    hipGraph_t g3;
    hipGraphExec_t ge3;
    OK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
    OK(hipMemsetAsync(counters, 0, 4096 * sizeof(unsigned), s));
    OK(hipStreamEndCapture(s, &g3));
    OK(hipGraphInstantiate(&ge3, g3, nullptr, nullptr, 0));
    float ms_m = 0;
    for (int w = 0; w < 5; w++) OK(hipGraphLaunch(ge3, s));
    OK(hipEventRecord(e0, s));
    for (int w = 0; w < reps; w++) OK(hipGraphLaunch(ge3, s));
    OK(hipEventRecord(e1, s));
    OK(hipEventSynchronize(e1));
    OK(hipEventElapsedTime(&ms_m, e0, e1));
    const double us_l = ms_l * 1e3 / reps, us_m = ms_m * 1e3 / reps;
    printf("%2d dependent launches (hipGraph):                          %7.2f us per pass = %5.2f us per stage\n", stage, us_l, us_l / stage);
    for (int v = 0; v < 2; v++) {
        const double us_p = ms_v[v] * 1e3 / reps;
        printf("one persistent launch, %s: %7.2f us per pass (incl. %.2f us counter memset) = %5.2f us per hop; results %s\n",
               v ? "sc1 payload + drained counter    " : "release / acquire fences + counter", us_p, us_m, (us_p - us_m) / stage, bad_v[v] ? "DIFFER (stale reads)" : "identical");
    }
    return 0;
}
