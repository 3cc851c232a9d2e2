// Micro-benchmark: issue rate of the VALU / cross-lane / LDS instructions the NTT and sweep kernels are built from,
// at 8 / 4 / 2 / 1 waves per SIMD (8 independent chains per wave).
// hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o tools/ubench_valu && tools/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define ITERS 1024

enum Op : int {
    ADD_U32, SUB_U32, ADD3_U32, LSHL_ADD_U32, AND_B32, LSHRREV_B32, ALIGNBIT_B32, BFE_U32, MIN_U32, CMP_CNDMASK,
    MUL_LO_U32, MUL_HI_U32, MAD_U64_U32_S, MAD_U64_U32_V, MUL_U32_U24, MUL_HI_U32_U24, MAD_U32_U24,
    FMA_F64, MUL_F64, ADD_F64, RNDNE_F64, FLOOR_F64, CVT_F64_U32, CVT_U32_F64,
    FMA_F32, PK_FMA_F32, PK_MUL_LO_U16, PK_MAD_U16, DOT4_U32_U8,
    DPP_QUAD, DPP_ROW_SHR, DPP_ROW_ROR, DPP_ROW_MIRROR, DPP_BCAST15, PERMLANE32_SWAP, PERMLANE16_SWAP, DS_BPERMUTE, DS_SWIZZLE,
    DS_WRITE_B64, DS_READ_B64, DS_WRITE_B128, DS_READ_B128, DEP_ADD, DEP_MUL_LO, DEP_MAD_U64, DEP_SHOUP, DEP2_SHOUP, DEP4_SHOUP, MIX_16ADD, MIX_LDS1W1R, MIX_16ADD_LDS, MIX_VMEM1, MIX_16ADD_VMEM, N_OPS
};

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed) {
    __shared__ uint64_t sh[256 * 4 + 64];
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;
    uint32_t b = seed | 1, c = seed * 77 + 5;
    uint64_t w0 = a0, w1 = a1, w2 = a2, w3 = a3, w4 = a4, w5 = a5, w6 = a6, w7 = a7;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7, db = 1.0000001, dc = 1e-9;
    float f0 = a0, f1 = a1, f2 = a2, f3 = a3, f4 = a4, f5 = a5, f6 = a6, f7 = a7, fb = 1.0001f, fc = 1e-3f;
    const uint32_t lds_addr = (uint32_t)(uintptr_t)(sh) + threadIdx.x * 8u, lds_addr16 = (uint32_t)(uintptr_t)(sh) + threadIdx.x * 16u;
    const uint32_t bp = ((threadIdx.x ^ 17u) & 63u) * 4u;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 q0 = {a0, a1, a2, a3}, q1 = {a4, a5, a6, a7};
    sh[threadIdx.x] = a0;
    __syncthreads();
    const uint32_t* gptr = out + (threadIdx.x & 63) * 4;  // 1 KiB per wave, L1-resident
#define ALL8(X) X(a0) X(a1) X(a2) X(a3) X(a4) X(a5) X(a6) X(a7)
#define ALLD(X) X(d0) X(d1) X(d2) X(d3) X(d4) X(d5) X(d6) X(d7)
#define ALLF(X) X(f0) X(f1) X(f2) X(f3) X(f4) X(f5) X(f6) X(f7)
#define ALLW(X) X(w0, a0) X(w1, a1) X(w2, a2) X(w3, a3) X(w4, a4) X(w5, a5) X(w6, a6) X(w7, a7)
    for (int i = 0; i < ITERS; i++) {
        if constexpr (OP == ADD_U32) {
#define X(r) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r) : "v"(b));
            ALL8(X)
#undef X
        } else if constexpr (OP == SUB_U32) {
#define X(r) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(r) : "v"(b));
            ALL8(X)
#undef X
        } else if constexpr (OP == ADD3_U32) {
#define X(r) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(c));
            ALL8(X)
#undef X
        } else if constexpr (OP == LSHL_ADD_U32) {
#define X(r) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(r) : "v"(b));
            ALL8(X)
#undef X
        } else if constexpr (OP == AND_B32) {
#define X(r) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r) : "v"(b));
            ALL8(X)
#undef X
        } else if constexpr (OP == LSHRREV_B32) {
#define X(r) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(r));
            ALL8(X)
#undef X
        } else if constexpr (OP == ALIGNBIT_B32) {
#define X(r) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(r) : "v"(b));
            ALL8(X)
#undef X
        } else if constexpr (OP == BFE_U32) {
#define X(r) asm volatile("v_bfe_u32 %0, %0, 3, 28" : "+v"(r));
            ALL8(X)
#undef X
        } else if constexpr (OP == MIN_U32) {
#define X(r) asm volatile("v_min_u32 %0, %0, %1" : "+v"(r) : "v"(b));
            ALL8(X)
#undef X
        } else if constexpr (OP == CMP_CNDMASK) {
#define X(r) asm volatile("v_cmp_gt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r) : "v"(b) : "vcc");
            ALL8(X)
#undef X
        } else if constexpr (OP == MUL_LO_U32) {
#define X(r) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r) : "v"(b));
            ALL8(X)
#undef X
        } else if constexpr (OP == MUL_HI_U32) {
#define X(r) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(r) : "v"(b));
            ALL8(X)
#undef X
        } else if constexpr (OP == MAD_U64_U32_S) {
#define X(r, s) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(r) : "v"(s), "s"(b) : "s20", "s21");
            ALLW(X)
#undef X
        } else if constexpr (OP == MAD_U64_U32_V) {
#define X(r, s) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(r) : "v"(s), "v"(b) : "s20", "s21");
            ALLW(X)
#undef X
        } else if constexpr (OP == MUL_U32_U24) {
#define X(r) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r) : "v"(b));
            ALL8(X)
#undef X
        } else if constexpr (OP == MUL_HI_U32_U24) {
#define X(r) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(r) : "v"(b));
            ALL8(X)
#undef X
        } else if constexpr (OP == MAD_U32_U24) {
#define X(r) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(c));
            ALL8(X)
#undef X
        } else if constexpr (OP == FMA_F64) {
#define X(r) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(r) : "v"(db), "v"(dc));
            ALLD(X)
#undef X
        } else if constexpr (OP == MUL_F64) {
#define X(r) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(r) : "v"(db));
            ALLD(X)
#undef X
        } else if constexpr (OP == ADD_F64) {
#define X(r) asm volatile("v_add_f64 %0, %0, %1" : "+v"(r) : "v"(dc));
            ALLD(X)
#undef X
        } else if constexpr (OP == RNDNE_F64) {
#define X(r) asm volatile("v_rndne_f64 %0, %0" : "+v"(r));
            ALLD(X)
#undef X
        } else if constexpr (OP == FLOOR_F64) {
#define X(r) asm volatile("v_floor_f64 %0, %0" : "+v"(r));
            ALLD(X)
#undef X
        } else if constexpr (OP == CVT_F64_U32) {
#define X(r, s) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(r) : "v"(s));
            X(d0, a0) X(d1, a1) X(d2, a2) X(d3, a3) X(d4, a4) X(d5, a5) X(d6, a6) X(d7, a7)
#undef X
        } else if constexpr (OP == CVT_U32_F64) {
#define X(r, s) asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(r) : "v"(s));
            X(a0, d0) X(a1, d1) X(a2, d2) X(a3, d3) X(a4, d4) X(a5, d5) X(a6, d6) X(a7, d7)
#undef X
        } else if constexpr (OP == FMA_F32) {
#define X(r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(fb), "v"(fc));
            ALLF(X)
#undef X
        } else if constexpr (OP == PK_FMA_F32) {
#define X(r) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(db), "v"(dc));
            ALLD(X)
#undef X
        } else if constexpr (OP == PK_MUL_LO_U16) {
#define X(r) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(r) : "v"(b));
            ALL8(X)
#undef X
        } else if constexpr (OP == PK_MAD_U16) {
#define X(r) asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(c));
            ALL8(X)
#undef X
        } else if constexpr (OP == DOT4_U32_U8) {
#define X(r) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(r) : "v"(b), "v"(c));
            ALL8(X)
#undef X
        } else if constexpr (OP == DPP_QUAD) {
#define X(r) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r));
            ALL8(X)
#undef X
        } else if constexpr (OP == DPP_ROW_SHR) {
#define X(r) asm volatile("v_add_u32_dpp %0, %0, %1 row_shr:4 row_mask:0xf bank_mask:0xf" : "+v"(r) : "v"(b));
            ALL8(X)
#undef X
        } else if constexpr (OP == DPP_ROW_ROR) {
#define X(r) asm volatile("v_mov_b32_dpp %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(r));
            ALL8(X)
#undef X
        } else if constexpr (OP == DPP_ROW_MIRROR) {
#define X(r) asm volatile("v_mov_b32_dpp %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(r));
            ALL8(X)
#undef X
        } else if constexpr (OP == DPP_BCAST15) {
#define X(r) asm volatile("v_mov_b32_dpp %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(r));
            ALL8(X)
#undef X
        } else if constexpr (OP == PERMLANE32_SWAP) {
            asm volatile("s_nop 1\n v_permlane32_swap_b32 %0, %1" : "+v"(a0), "+v"(a1));
            asm volatile("s_nop 1\n v_permlane32_swap_b32 %0, %1" : "+v"(a2), "+v"(a3));
            asm volatile("s_nop 1\n v_permlane32_swap_b32 %0, %1" : "+v"(a4), "+v"(a5));
            asm volatile("s_nop 1\n v_permlane32_swap_b32 %0, %1" : "+v"(a6), "+v"(a7));
            asm volatile("s_nop 1\n v_permlane32_swap_b32 %0, %1" : "+v"(a0), "+v"(a2));
            asm volatile("s_nop 1\n v_permlane32_swap_b32 %0, %1" : "+v"(a1), "+v"(a3));
            asm volatile("s_nop 1\n v_permlane32_swap_b32 %0, %1" : "+v"(a4), "+v"(a6));
            asm volatile("s_nop 1\n v_permlane32_swap_b32 %0, %1" : "+v"(a5), "+v"(a7));
        } else if constexpr (OP == PERMLANE16_SWAP) {
            asm volatile("s_nop 1\n v_permlane16_swap_b32 %0, %1" : "+v"(a0), "+v"(a1));
            asm volatile("s_nop 1\n v_permlane16_swap_b32 %0, %1" : "+v"(a2), "+v"(a3));
            asm volatile("s_nop 1\n v_permlane16_swap_b32 %0, %1" : "+v"(a4), "+v"(a5));
            asm volatile("s_nop 1\n v_permlane16_swap_b32 %0, %1" : "+v"(a6), "+v"(a7));
            asm volatile("s_nop 1\n v_permlane16_swap_b32 %0, %1" : "+v"(a0), "+v"(a2));
            asm volatile("s_nop 1\n v_permlane16_swap_b32 %0, %1" : "+v"(a1), "+v"(a3));
            asm volatile("s_nop 1\n v_permlane16_swap_b32 %0, %1" : "+v"(a4), "+v"(a6));
            asm volatile("s_nop 1\n v_permlane16_swap_b32 %0, %1" : "+v"(a5), "+v"(a7));
        } else if constexpr (OP == DS_BPERMUTE) {
#define X(r) asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(r) : "v"(bp));
            ALL8(X)
#undef X
        } else if constexpr (OP == DS_SWIZZLE) {
#define X(r) asm volatile("ds_swizzle_b32 %0, %0 offset:swizzle(BITMASK_PERM, \"0000p\")\n s_waitcnt lgkmcnt(0)" : "+v"(r));
            ALL8(X)
#undef X
        } else if constexpr (OP == DS_WRITE_B64) {
#define X(r, s) asm volatile("ds_write_b64 %0, %1" : : "v"(lds_addr), "v"(r) : "memory");
            ALLW(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if constexpr (OP == DS_READ_B64) {
#define X(r, s) asm volatile("ds_read_b64 %0, %1" : "=v"(r) : "v"(lds_addr) : "memory");
            ALLW(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if constexpr (OP == DS_WRITE_B128) {
            asm volatile("ds_write_b128 %0, %1\n ds_write_b128 %0, %2\n ds_write_b128 %0, %1\n ds_write_b128 %0, %2\n ds_write_b128 %0, %1\n ds_write_b128 %0, %2\n ds_write_b128 %0, %1\n ds_write_b128 %0, %2\n s_waitcnt lgkmcnt(0)" : : "v"(lds_addr16), "v"(q0), "v"(q1) : "memory");
        } else if constexpr (OP == DEP_ADD) {  // ONE dependent chain: cycles per instruction = issue-to-issue latency of dependent VALU
#define X(r) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a0) : "v"(b));
            ALL8(X)
#undef X
        } else if constexpr (OP == DEP_MUL_LO) {
#define X(r) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a0) : "v"(b));
            ALL8(X)
#undef X
        } else if constexpr (OP == DEP_MAD_U64) {
#define X(r, s) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(w0) : "v"(a0), "v"(b) : "s20", "s21");
            ALLW(X)
#undef X
        } else if constexpr (OP == DEP_SHOUP) {  // the butterfly's dependent chain: mul_lo + mul_hi -> mad -> add / sub -> add (6 instructions), one butterfly at a time
#define X(x, y) asm volatile("v_mul_lo_u32 v100, %2, %1\n v_mul_hi_u32 v101, %3, %1\n v_mad_u64_u32 v[100:101], s[20:21], v101, %4, v[100:101]\n v_add_u32 %0, v100, %0\n v_sub_u32 %1, %0, v100\n v_add_u32 %1, %5, %1" : "+v"(x), "+v"(y) : "v"(b), "v"(c), "v"(a7), "v"(a6) : "v100", "v101", "s20", "s21");
            X(a0, a1) X(a0, a1) X(a0, a1) X(a0, a1) X(a0, a1) X(a0, a1) X(a0, a1) X(a0, a1)
#undef X
        } else if constexpr (OP == DEP2_SHOUP) {  // two butterflies interleaved instruction by instruction
#define X(x, y, u, v) asm volatile("v_mul_lo_u32 v100, %4, %1\n v_mul_lo_u32 v102, %4, %3\n v_mul_hi_u32 v101, %5, %1\n v_mul_hi_u32 v103, %5, %3\n v_mad_u64_u32 v[100:101], s[20:21], v101, %6, v[100:101]\n v_mad_u64_u32 v[102:103], s[20:21], v103, %6, v[102:103]\n v_add_u32 %0, v100, %0\n v_add_u32 %2, v102, %2\n v_sub_u32 %1, %0, v100\n v_sub_u32 %3, %2, v102\n v_add_u32 %1, %7, %1\n v_add_u32 %3, %7, %3" : "+v"(x), "+v"(y), "+v"(u), "+v"(v) : "v"(b), "v"(c), "v"(a7), "v"(a6) : "v100", "v101", "v102", "v103", "s20", "s21");
            X(a0, a1, a2, a3) X(a0, a1, a2, a3) X(a0, a1, a2, a3) X(a0, a1, a2, a3)
#undef X
        } else if constexpr (OP == DEP4_SHOUP) {  // four butterflies interleaved
#define X(x0, y0, x1, y1, x2, y2, x3, y3) asm volatile( \
    "v_mul_lo_u32 v100, %8, %1\n v_mul_lo_u32 v102, %8, %3\n v_mul_lo_u32 v104, %8, %5\n v_mul_lo_u32 v106, %8, %7\n" \
    "v_mul_hi_u32 v101, %9, %1\n v_mul_hi_u32 v103, %9, %3\n v_mul_hi_u32 v105, %9, %5\n v_mul_hi_u32 v107, %9, %7\n" \
    "v_mad_u64_u32 v[100:101], s[20:21], v101, %10, v[100:101]\n v_mad_u64_u32 v[102:103], s[20:21], v103, %10, v[102:103]\n v_mad_u64_u32 v[104:105], s[20:21], v105, %10, v[104:105]\n v_mad_u64_u32 v[106:107], s[20:21], v107, %10, v[106:107]\n" \
    "v_add_u32 %0, v100, %0\n v_add_u32 %2, v102, %2\n v_add_u32 %4, v104, %4\n v_add_u32 %6, v106, %6\n" \
    "v_sub_u32 %1, %0, v100\n v_sub_u32 %3, %2, v102\n v_sub_u32 %5, %4, v104\n v_sub_u32 %7, %6, v106\n" \
    "v_add_u32 %1, %11, %1\n v_add_u32 %3, %11, %3\n v_add_u32 %5, %11, %5\n v_add_u32 %7, %11, %7" \
    : "+v"(x0), "+v"(y0), "+v"(x1), "+v"(y1), "+v"(x2), "+v"(y2), "+v"(x3), "+v"(y3) : "v"(b), "v"(c), "v"(f0), "v"(f1) : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "s20", "s21");
            X(a0, a1, a2, a3, a4, a5, a6, a7) X(a0, a1, a2, a3, a4, a5, a6, a7)
#undef X
        } else if constexpr (OP == MIX_16ADD || OP == MIX_16ADD_LDS || OP == MIX_LDS1W1R || OP == MIX_VMEM1 || OP == MIX_16ADD_VMEM) {
            // do the VALU, LDS and vector-memory pipes overlap?  16 adds / one ds_write_b64 + one ds_read_b64 / one 16-byte global
            // load, alone and together, per iteration ("cycles" printed = per 1/8 iteration)
            if constexpr (OP == MIX_16ADD || OP == MIX_16ADD_LDS || OP == MIX_16ADD_VMEM) {
#define X(r) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r) : "v"(b));
                ALL8(X) ALL8(X)
#undef X
            }
            if constexpr (OP == MIX_16ADD_LDS || OP == MIX_LDS1W1R) {
                asm volatile("ds_write_b64 %0, %1" : : "v"(lds_addr), "v"(w0) : "memory");
                asm volatile("ds_read_b64 %0, %1" : "=v"(w1) : "v"(lds_addr) : "memory");
                if ((i & 7) == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            if constexpr (OP == MIX_VMEM1 || OP == MIX_16ADD_VMEM) {
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(q1) : "v"(gptr) : "memory");
                if ((i & 7) == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        } else if constexpr (OP == DS_READ_B128) {
            u32x4 t0, t1;
            asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2\n ds_read_b128 %0, %2\n ds_read_b128 %1, %2\n ds_read_b128 %0, %2\n ds_read_b128 %1, %2\n ds_read_b128 %0, %2\n ds_read_b128 %1, %2\n s_waitcnt lgkmcnt(0)" : "=&v"(t0), "=&v"(t1) : "v"(lds_addr16) : "memory");
            q0 ^= t0;
            q1 ^= t1;
        }
    }
    uint32_t r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)(w0 ^ w1 ^ w2 ^ w3 ^ w4 ^ w5 ^ w6 ^ w7) ^ (uint32_t)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) ^
                 (uint32_t)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7) ^ q0.x ^ q1.y;
    if (r == 0x12345678) out[threadIdx.x] = r;
}

template <int OP>
void run(const char* name, uint32_t* d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("%-24s", name);
    for (int wps : {8, 4, 2, 1}) {  // waves per SIMD = blocks of 256 threads per CU
        const int blocks = 256 * wps;
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double inst = (double)blocks * 4 /*waves*/ * ITERS * 8;  // wave-instructions
        const double per_simd_per_s = inst / (256.0 * 4) / (ms * 1e-3);
        printf("  %dw/SIMD %6.2f cyc", wps, 2.4e9 / per_simd_per_s);
    }
    printf("   (cycles per wave-instruction per SIMD @2.4 GHz)\n");
}

int main() {
    uint32_t* d;
    hipMalloc(&d, 1 << 20);
#define R(op) run<op>(#op, d);
    R(ADD_U32) R(SUB_U32) R(ADD3_U32) R(LSHL_ADD_U32) R(AND_B32) R(LSHRREV_B32) R(ALIGNBIT_B32) R(BFE_U32) R(MIN_U32) R(CMP_CNDMASK)
    R(MUL_LO_U32) R(MUL_HI_U32) R(MAD_U64_U32_S) R(MAD_U64_U32_V) R(MUL_U32_U24) R(MUL_HI_U32_U24) R(MAD_U32_U24)
    R(FMA_F64) R(MUL_F64) R(ADD_F64) R(RNDNE_F64) R(FLOOR_F64) R(CVT_F64_U32) R(CVT_U32_F64)
    R(FMA_F32) R(PK_FMA_F32) R(PK_MUL_LO_U16) R(PK_MAD_U16) R(DOT4_U32_U8)
    R(DPP_QUAD) R(DPP_ROW_SHR) R(DPP_ROW_ROR) R(DPP_ROW_MIRROR) R(DPP_BCAST15) R(PERMLANE32_SWAP) R(PERMLANE16_SWAP) R(DS_BPERMUTE) R(DS_SWIZZLE)
    R(DS_WRITE_B64) R(DS_READ_B64) R(DS_WRITE_B128) R(DS_READ_B128)
    printf("dependent chains (8 instructions of ONE chain per iteration: the figure is issue-to-issue latency when one wave runs alone):\n");
    R(DEP_ADD) R(DEP_MUL_LO) R(DEP_MAD_U64)
    printf("Shoup butterflies, 48 instructions per iteration (cycles per INSTRUCTION = figure / 6):\n");
    R(DEP_SHOUP) R(DEP2_SHOUP) R(DEP4_SHOUP)
    printf("pipe overlap (per 1/8 iteration; an iteration = 16 v_add and / or 1 ds_write_b64 + 1 ds_read_b64 and / or 1 global_load_dwordx4 from L1):\n");
    R(MIX_16ADD) R(MIX_LDS1W1R) R(MIX_16ADD_LDS) R(MIX_VMEM1) R(MIX_16ADD_VMEM)
    return 0;
}
