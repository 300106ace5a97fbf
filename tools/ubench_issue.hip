// ubench_issue.hip -- developer micro-benchmark (not part of the product): what one instruction costs the SIMD's issue
// port on gfx950, with 1, 2 and 4 waves per SIMD.  Each test runs REP x 32 copies of one instruction (independent
// destination registers) between two s_memtime stamps; prints cycles per wave-instruction per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_issue.hip -o gpurun_out/ubench && gpurun_out/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP 64

#define X4(s) s s s s
#define X32(s) X4(X4(s)) X4(X4(s))

template <int TEST> __global__ void k(unsigned long long* out, float* sink, const float* src) {
    float a0 = src[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    f4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    h8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)a0; hb[i] = (_Float16)a1; }
    unsigned u0 = __float_as_uint(a0), u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3;
    int sgpr_dst;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REP; ++r) {
        if constexpr (TEST == 0) {  // v_fma_f32, 4 independent chains
            asm volatile(X4(X4("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5));
        } else if constexpr (TEST == 1) {  // v_pk_fma_f32
            asm volatile(X4(X4("v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3\n")) : "+v"(p0), "+v"(p1) : "v"(p2), "v"(p3));
        } else if constexpr (TEST == 2) {  // v_pk_mul_f32
            asm volatile(X4(X4("v_pk_mul_f32 %0, %0, %2\n v_pk_mul_f32 %1, %1, %3\n")) : "+v"(p0), "+v"(p1) : "v"(p2), "v"(p3));
        } else if constexpr (TEST == 3) {  // v_add_f64
            asm volatile(X4(X4("v_add_f64 %0, %0, %2\n v_add_f64 %1, %1, %3\n")) : "+v"(d0), "+v"(d1) : "v"(d2), "v"(d3));
        } else if constexpr (TEST == 4) {  // v_cvt_f32_f64
            asm volatile(X4(X4("v_cvt_f32_f64 %0, %2\n v_cvt_f32_f64 %1, %3\n")) : "+v"(a0), "+v"(a1) : "v"(d2), "v"(d3));
        } else if constexpr (TEST == 5) {  // v_and_or_b32 + v_min_u32 pair
            asm volatile(X4(X4("v_and_or_b32 %0, %2, %3, %0\n v_min_u32 %1, %1, %0\n")) : "+v"(u0), "+v"(u1) : "v"(u2), "v"(u3));
        } else if constexpr (TEST == 6) {  // v_min3_u32
            asm volatile(X4(X4("v_min3_u32 %0, %0, %2, %3\n v_min3_u32 %1, %1, %2, %3\n")) : "+v"(u0), "+v"(u1) : "v"(u2), "v"(u3));
        } else if constexpr (TEST == 7) {  // v_med3_f32
            asm volatile(X4(X4("v_med3_f32 %0, %0, %2, %3\n v_med3_f32 %1, %1, %2, %3\n")) : "+v"(a0), "+v"(a1) : "v"(a2), "v"(a3));
        } else if constexpr (TEST == 8) {  // v_cvt_pk_f16_f32
            asm volatile(X4(X4("v_cvt_pk_f16_f32 %0, %2, %3\n v_cvt_pk_f16_f32 %1, %3, %2\n")) : "+v"(u0), "+v"(u1) : "v"(a2), "v"(a3));
        } else if constexpr (TEST == 9) {  // v_cvt_f32_f16 (plain + sdwa high half)
            asm volatile(X4(X4("v_cvt_f32_f16_e32 %0, %2\n v_cvt_f32_f16_sdwa %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n")) : "+v"(a0), "+v"(a1) : "v"(u2));
        } else if constexpr (TEST == 10) {  // v_rcp_f32
            asm volatile(X4(X4("v_rcp_f32 %0, %2\n v_rcp_f32 %1, %3\n")) : "+v"(a0), "+v"(a1) : "v"(a2), "v"(a3));
        } else if constexpr (TEST == 11) {  // v_exp_f32
            asm volatile(X4(X4("v_exp_f32 %0, %2\n v_exp_f32 %1, %3\n")) : "+v"(a0), "+v"(a1) : "v"(a2), "v"(a3));
        } else if constexpr (TEST == 12) {  // v_readlane_b32
            asm volatile(X4(X4("v_readlane_b32 %0, %1, 3\n v_readlane_b32 %0, %2, 5\n")) : "=s"(sgpr_dst) : "v"(a2), "v"(a3));
        } else if constexpr (TEST == 13) {  // mfma 16x16x32 f16, 4 independent accumulators
            asm volatile(X4("v_mfma_f32_16x16x32_f16 %0, %4, %5, %0\n v_mfma_f32_16x16x32_f16 %1, %4, %5, %1\n v_mfma_f32_16x16x32_f16 %2, %4, %5, %2\n v_mfma_f32_16x16x32_f16 %3, %4, %5, %3\n")
                         X4("v_mfma_f32_16x16x32_f16 %0, %4, %5, %0\n v_mfma_f32_16x16x32_f16 %1, %4, %5, %1\n v_mfma_f32_16x16x32_f16 %2, %4, %5, %2\n v_mfma_f32_16x16x32_f16 %3, %4, %5, %3\n")
                         : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(ha), "v"(hb));
        } else if constexpr (TEST == 14) {  // 1 mfma + 3 v_fma per group (32 instructions: 8 mfma + 24 fma)
            asm volatile(X4("v_mfma_f32_16x16x32_f16 %0, %4, %5, %0\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n v_fma_f32 %6, %6, %8, %9\n"
                            "v_mfma_f32_16x16x32_f16 %1, %4, %5, %1\n v_fma_f32 %7, %7, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
                         : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(ha), "v"(hb), "v"(a0), "v"(a1), "v"(a4), "v"(a5));
        } else if constexpr (TEST == 15) {  // 1 mfma + 7 v_fma per group (32 instructions: 4 mfma + 28 fma)
            asm volatile(X4("v_mfma_f32_16x16x32_f16 %0, %4, %5, %0\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n v_fma_f32 %6, %6, %8, %9\n"
                            "v_fma_f32 %7, %7, %8, %9\n v_fma_f32 %7, %7, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
                         : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(ha), "v"(hb), "v"(a0), "v"(a1), "v"(a4), "v"(a5));
        } else if constexpr (TEST == 16) {  // v_mul_f32 with an SGPR operand
            asm volatile(X4(X4("v_mul_f32 %0, %2, %0\n v_mul_f32 %1, %2, %1\n")) : "+v"(a0), "+v"(a1) : "s"(sgpr_dst = 3));
        } else if constexpr (TEST == 17) {  // dependent v_fma chain (latency)
            asm volatile(X32("v_fma_f32 %0, %0, %1, %2\n") : "+v"(a0) : "v"(a4), "v"(a5));
        } else if constexpr (TEST == 18) {  // dependent v_pk_fma chain
            asm volatile(X32("v_pk_fma_f32 %0, %0, %1, %2\n") : "+v"(p0) : "v"(p2), "v"(p3));
        } else if constexpr (TEST == 19) {  // v_cndmask + v_cmp pair
            asm volatile(X4(X4("v_cmp_lt_f32 vcc, %2, %3\n v_cndmask_b32 %0, %0, %1, vcc\n")) : "+v"(a0) : "v"(a1), "v"(a2), "v"(a3) : "vcc");
        } else if constexpr (TEST == 20) {  // v_perm_b32
            asm volatile(X4(X4("v_perm_b32 %0, %2, %3, %1\n v_perm_b32 %1, %3, %2, %0\n")) : "+v"(u0), "+v"(u1) : "v"(u2), "v"(u3));
        } else if constexpr (TEST == 21) {  // v_mul_f64
            asm volatile(X4(X4("v_mul_f64 %0, %0, %2\n v_mul_f64 %1, %1, %3\n")) : "+v"(d0), "+v"(d1) : "v"(d2), "v"(d3));
        } else if constexpr (TEST == 22) {  // v_mov_b32 DPP row_ror
            asm volatile(X4(X4("v_mov_b32_dpp %0, %2 row_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %3 row_ror:2 row_mask:0xf bank_mask:0xf\n")) : "+v"(a0), "+v"(a1) : "v"(a2), "v"(a3));
        } else if constexpr (TEST == 23) {  // v_mul_f32 vgpr
            asm volatile(X4(X4("v_mul_f32 %0, %2, %0\n v_mul_f32 %1, %3, %1\n")) : "+v"(a0), "+v"(a1) : "v"(a2), "v"(a3));
        } else if constexpr (TEST == 24) {  // v_add_f32 vgpr
            asm volatile(X4(X4("v_add_f32 %0, %2, %0\n v_add_f32 %1, %3, %1\n")) : "+v"(a0), "+v"(a1) : "v"(a2), "v"(a3));
        } else if constexpr (TEST == 25) {  // v_fmac_f32
            asm volatile(X4(X4("v_fmac_f32 %0, %2, %3\n v_fmac_f32 %1, %3, %2\n")) : "+v"(a0), "+v"(a1) : "v"(a2), "v"(a3));
        } else if constexpr (TEST == 26) {  // v_fma_f32 sgpr src
            asm volatile(X4(X4("v_fma_f32 %0, %0, %4, %2\n v_fma_f32 %1, %1, %4, %3\n")) : "+v"(a0), "+v"(a1) : "v"(a2), "v"(a3), "s"(sgpr_dst = 5));
        } else if constexpr (TEST == 27) {  // v_fma_f32 neg mod
            asm volatile(X4(X4("v_fma_f32 %0, %0, -%2, %3\n v_fma_f32 %1, -%1, %3, %2\n")) : "+v"(a0), "+v"(a1) : "v"(a2), "v"(a3));
        } else if constexpr (TEST == 28) {  // v_max_f32
            asm volatile(X4(X4("v_max_f32 %0, %2, %0\n v_max_f32 %1, %3, %1\n")) : "+v"(a0), "+v"(a1) : "v"(a2), "v"(a3));
        } else if constexpr (TEST == 29) {  // v_min_u32
            asm volatile(X4(X4("v_min_u32 %0, %2, %0\n v_min_u32 %1, %3, %1\n")) : "+v"(u0), "+v"(u1) : "v"(u2), "v"(u3));
        } else if constexpr (TEST == 30) {  // v_and_b32
            asm volatile(X4(X4("v_and_b32 %0, %2, %0\n v_and_b32 %1, %3, %1\n")) : "+v"(u0), "+v"(u1) : "v"(u2), "v"(u3));
        } else if constexpr (TEST == 31) {  // v_and_or_b32
            asm volatile(X4(X4("v_and_or_b32 %0, %2, %3, %0\n v_and_or_b32 %1, %3, %2, %1\n")) : "+v"(u0), "+v"(u1) : "v"(u2), "v"(u3));
        } else if constexpr (TEST == 32) {  // v_add_u32
            asm volatile(X4(X4("v_add_u32 %0, %2, %0\n v_add_u32 %1, %3, %1\n")) : "+v"(u0), "+v"(u1) : "v"(u2), "v"(u3));
        } else if constexpr (TEST == 33) {  // v_lshlrev_b32
            asm volatile(X4(X4("v_lshlrev_b32 %0, 3, %2\n v_lshlrev_b32 %1, 5, %3\n")) : "+v"(u0), "+v"(u1) : "v"(u2), "v"(u3));
        } else if constexpr (TEST == 34) {  // v_mov_b32
            asm volatile(X4(X4("v_mov_b32 %0, %2\n v_mov_b32 %1, %3\n")) : "+v"(u0), "+v"(u1) : "v"(u2), "v"(u3));
        } else if constexpr (TEST == 35) {  // v_cndmask_b32 (vcc fixed)
            asm volatile(X4(X4("v_cndmask_b32 %0, %2, %3, vcc\n v_cndmask_b32 %1, %3, %2, vcc\n")) : "+v"(u0), "+v"(u1) : "v"(u2), "v"(u3));
        } else if constexpr (TEST == 36) {  // v_cmp_lt_f32
            asm volatile(X4(X4("v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %0\n")) : : "v"(a2), "v"(a3) : "vcc");
        } else if constexpr (TEST == 37) {  // v_fma_mix_f32
            asm volatile(X4(X4("v_fma_mix_f32 %0, %2, %3, %0 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %2, %3, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n")) : "+v"(a0), "+v"(a1) : "v"(u2), "v"(a3));
        } else if constexpr (TEST == 38) {  // v_cvt_pkrtz_f16_f32
            asm volatile(X4(X4("v_cvt_pkrtz_f16_f32 %0, %2, %3\n v_cvt_pkrtz_f16_f32 %1, %3, %2\n")) : "+v"(u0), "+v"(u1) : "v"(a2), "v"(a3));
        } else if constexpr (TEST == 39) {  // v_pk_max_f16
            asm volatile(X4(X4("v_pk_max_f16 %0, %2, %0\n v_pk_max_f16 %1, %3, %1\n")) : "+v"(u0), "+v"(u1) : "v"(u2), "v"(u3));
        } else if constexpr (TEST == 40) {  // v_mul_u32_u24
            asm volatile(X4(X4("v_mul_u32_u24 %0, %2, %0\n v_mul_u32_u24 %1, %3, %1\n")) : "+v"(u0), "+v"(u1) : "v"(u2), "v"(u3));
        } else if constexpr (TEST == 41) {  // v_sub_f32 + v_mul_f32 mix
            asm volatile(X4(X4("v_sub_f32 %0, %2, %0\n v_mul_f32 %1, %3, %1\n")) : "+v"(a0), "+v"(a1) : "v"(a2), "v"(a3));
        } else if constexpr (TEST == 42) {  // v_fma_f32 + v_pk_fma alternating
            asm volatile(X4(X4("v_fma_f32 %0, %0, %3, %3\n v_pk_fma_f32 %1, %1, %2, %2\n")) : "+v"(a0), "+v"(p1) : "v"(p2), "v"(a3));
        } else if constexpr (TEST == 43) {  // v_fma_f32 + v_and_or alternating
            asm volatile(X4(X4("v_fma_f32 %0, %0, %3, %3\n v_and_or_b32 %1, %2, %1, %2\n")) : "+v"(a0), "+v"(u1) : "v"(u2), "v"(a3));
        } else if constexpr (TEST == 44) {  // v_max3_f32
            asm volatile(X4(X4("v_max3_f32 %0, %2, %3, %0\n v_max3_f32 %1, %3, %2, %1\n")) : "+v"(a0), "+v"(a1) : "v"(a2), "v"(a3));
        } else if constexpr (TEST == 45) {  // v_cvt_f32_i32
            asm volatile(X4(X4("v_cvt_f32_i32 %0, %2\n v_cvt_f32_i32 %1, %3\n")) : "+v"(a0), "+v"(a1) : "v"(u2), "v"(u3));
        } else if constexpr (TEST == 46) {  // v_log_f32
            asm volatile(X4(X4("v_log_f32 %0, %2\n v_log_f32 %1, %3\n")) : "+v"(a0), "+v"(a1) : "v"(a2), "v"(a3));
        } else if constexpr (TEST == 47) {  // ds_read_b128 (same addr)
            asm volatile(X4(X4("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:16\n")) "s_waitcnt lgkmcnt(0)\n" : "=v"(acc0), "=v"(acc1) : "v"(u2 & 0x3f0));
        } else if constexpr (TEST == 48) {  // ds_read_b64 (per-lane addr)
            asm volatile(X4(X4("ds_read_b64 %0, %2\n ds_read_b64 %1, %2 offset:8\n")) "s_waitcnt lgkmcnt(0)\n" : "=v"(p0), "=v"(p1) : "v"((threadIdx.x & 63) * 8));
        } else if constexpr (TEST == 49) {  // v_rcp_f64
            asm volatile(X4(X4("v_rcp_f64 %0, %2\n v_rcp_f64 %1, %3\n")) : "+v"(d0), "+v"(d1) : "v"(d2), "v"(d3));
        } else if constexpr (TEST == 50) {  // v_fma_f64
            asm volatile(X4(X4("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %3, %2\n")) : "+v"(d0), "+v"(d1) : "v"(d2), "v"(d3));
        } else if constexpr (TEST == 51) {  // v_cmp_lt_f64
            asm volatile(X4(X4("v_cmp_lt_f64 vcc, %0, %1\n v_cmp_lt_f64 vcc, %1, %0\n")) : : "v"(d2), "v"(d3) : "vcc");
        } else if constexpr (TEST == 52) {  // v_min_f64
            asm volatile(X4(X4("v_min_f64 %0, %0, %2\n v_min_f64 %1, %1, %3\n")) : "+v"(d0), "+v"(d1) : "v"(d2), "v"(d3));
        } else if constexpr (TEST == 53) {  // v_cvt_f64_f32
            asm volatile(X4(X4("v_cvt_f64_f32 %0, %2\n v_cvt_f64_f32 %1, %3\n")) : "+v"(d0), "+v"(d1) : "v"(a2), "v"(a3));
        } else if constexpr (TEST == 54) {  // v_min3_f32 |.|
            asm volatile(X4(X4("v_min3_f32 %0, %0, |%2|, |%3|\n v_min3_f32 %1, %1, |%3|, |%2|\n")) : "+v"(a0), "+v"(a1) : "v"(a2), "v"(a3));
        } else if constexpr (TEST == 55) {  // v_pk_mul_f32 clamp
            asm volatile(X4(X4("v_pk_mul_f32 %0, %0, %2 clamp\n v_pk_mul_f32 %1, %1, %3 clamp\n")) : "+v"(p0), "+v"(p1) : "v"(p2), "v"(p3));
        } else if constexpr (TEST == 56) {  // v_and_or_b32 sgpr mask, inline constant
            asm volatile(X4(X4("v_and_or_b32 %0, %0, %2, 7\n v_and_or_b32 %1, %1, %2, 9\n")) : "+v"(u0), "+v"(u1) : "s"(sgpr_dst = 0xffffffe0));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x % 64 == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    float s = a0 + a1 + a2 + a3 + (float)(d0 + d1) + p0.x + p0.y + p1.x + p1.y + acc0[0] + acc1[1] + acc2[2] + acc3[3] + __uint_as_float(u0 ^ u1) + (float)sgpr_dst;
    if (s == 12345.678f) sink[threadIdx.x] = s;
}

// latency tests: dependent loads from LDS through ds_read vs flat_load, and scalar loads
__global__ void lat(unsigned long long* out, float* sink, const int* chain_g) {
    __shared__ int chain[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) chain[i] = (i * 17 + 5) & 1023;
    __syncthreads();
    int idx = threadIdx.x & 1023;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < 256; ++r) idx = chain[idx];  // ds_read_b32 dependent chain
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const int* generic = chain;  // generic pointer to LDS -> flat_load
    int idx2 = threadIdx.x & 1023;
    asm volatile("" : "+v"(generic));
    for (int r = 0; r < 256; ++r) idx2 = generic[idx2];
    unsigned long long t2 = __builtin_amdgcn_s_memtime();
    int sidx = 0;
    for (int r = 0; r < 256; ++r) sidx = __builtin_amdgcn_readfirstlane(chain_g[sidx]);  // global dependent chain (vector load, L2/L1 hit)
    unsigned long long t3 = __builtin_amdgcn_s_memtime();
    typedef const __attribute__((address_space(4))) int* CI;
    CI cg = (CI)(const void*)chain_g;
    int s2 = 0;
    for (int r = 0; r < 256; ++r) s2 = cg[s2];  // s_load dependent chain (scalar cache hit)
    unsigned long long t4 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) {
        out[0] = (t1 - t0);
        out[1] = (t2 - t1);
        out[2] = (t3 - t2);
        out[3] = (t4 - t3);
    }
    if (idx + idx2 + sidx + s2 == -7) sink[0] = 1.0f;
}

static const char* NAMES[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_add_f64", "v_cvt_f32_f64", "v_and_or+v_min_u32", "v_min3_u32",
                              "v_med3_f32", "v_cvt_pk_f16_f32", "v_cvt_f32_f16(+sdwa)", "v_rcp_f32", "v_exp_f32", "v_readlane_b32",
                              "mfma_16x16x32_f16", "1 mfma + 3 fma", "1 mfma + 7 fma", "v_mul_f32 sgpr", "dep v_fma chain", "dep v_pk_fma chain",
                              "v_cmp+v_cndmask", "v_perm_b32", "v_mul_f64", "v_mov_dpp", "v_mul_f32 vgpr", "v_add_f32 vgpr", "v_fmac_f32", "v_fma_f32 sgpr src", "v_fma_f32 neg mod", "v_max_f32", "v_min_u32", "v_and_b32", "v_and_or_b32", "v_add_u32", "v_lshlrev_b32", "v_mov_b32", "v_cndmask_b32 (vcc fixed)", "v_cmp_lt_f32", "v_fma_mix_f32", "v_cvt_pkrtz_f16_f32", "v_pk_max_f16", "v_mul_u32_u24", "v_sub_f32 + v_mul_f32 mix", "v_fma_f32 + v_pk_fma alternating", "v_fma_f32 + v_and_or alternating", "v_max3_f32", "v_cvt_f32_i32", "v_log_f32", "ds_read_b128 (same addr)", "ds_read_b64 (per-lane addr)", "v_rcp_f64", "v_fma_f64", "v_cmp_lt_f64", "v_min_f64", "v_cvt_f64_f32", "v_min3_f32 abs", "v_pk_mul_f32 clamp", "v_and_or_b32 sgpr+inline"};

template <int TEST> void run(unsigned long long* d_out, float* d_sink, float* d_src) {
    const int per_rep = 32;
    for (int waves : {1, 2}) {  // waves per SIMD
        const int threads = 256 * waves;
        hipLaunchKernelGGL(k<TEST>, dim3(256), dim3(threads), 0, 0, d_out, d_sink, d_src);
        hipLaunchKernelGGL(k<TEST>, dim3(256), dim3(threads), 0, 0, d_out, d_sink, d_src);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256 * 4 * waves);
        hipMemcpy(h.data(), d_out, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double med = (double)h[h.size() / 2];
        // cycles per wave-instruction per SIMD = wave's elapsed cycles / (instructions of all waves on that SIMD)
        printf("{\"test\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_instr_per_wave\": %.2f, \"simd_cycles_per_instr\": %.2f}\n", NAMES[TEST], waves,
               med / (REP * per_rep), med / (REP * per_rep * waves));
    }
}

int main() {
    unsigned long long* d_out;
    float *d_sink, *d_src;
    int* d_chain;
    hipMalloc(&d_out, 1 << 20);
    hipMalloc(&d_sink, 1 << 16);
    hipMalloc(&d_src, 1 << 16);
    hipMalloc(&d_chain, 4096);
    std::vector<float> src(4096);
    for (int i = 0; i < 4096; ++i) src[i] = 0.5f + 0.001f * i;
    hipMemcpy(d_src, src.data(), 4096 * 4, hipMemcpyHostToDevice);
    std::vector<int> chain(1024);
    for (int i = 0; i < 1024; ++i) chain[i] = (i * 17 + 5) & 1023;
    hipMemcpy(d_chain, chain.data(), 4096, hipMemcpyHostToDevice);
    printf("# s_memtime ticks; one wave's elapsed ticks / its instruction count (per wave) and / all waves of the SIMD (per SIMD)\n");
    run<0>(d_out, d_sink, d_src); run<1>(d_out, d_sink, d_src); run<2>(d_out, d_sink, d_src); run<3>(d_out, d_sink, d_src);
    run<4>(d_out, d_sink, d_src); run<5>(d_out, d_sink, d_src); run<6>(d_out, d_sink, d_src); run<7>(d_out, d_sink, d_src);
    run<8>(d_out, d_sink, d_src); run<9>(d_out, d_sink, d_src); run<10>(d_out, d_sink, d_src); run<11>(d_out, d_sink, d_src);
    run<12>(d_out, d_sink, d_src); run<13>(d_out, d_sink, d_src); run<14>(d_out, d_sink, d_src); run<15>(d_out, d_sink, d_src);
    run<16>(d_out, d_sink, d_src); run<17>(d_out, d_sink, d_src); run<18>(d_out, d_sink, d_src); run<19>(d_out, d_sink, d_src);
    run<20>(d_out, d_sink, d_src); run<21>(d_out, d_sink, d_src); run<22>(d_out, d_sink, d_src);
    run<23>(d_out, d_sink, d_src); run<24>(d_out, d_sink, d_src); run<25>(d_out, d_sink, d_src); run<26>(d_out, d_sink, d_src); run<27>(d_out, d_sink, d_src); run<28>(d_out, d_sink, d_src); run<29>(d_out, d_sink, d_src); run<30>(d_out, d_sink, d_src); run<31>(d_out, d_sink, d_src); run<32>(d_out, d_sink, d_src); run<33>(d_out, d_sink, d_src); run<34>(d_out, d_sink, d_src); run<35>(d_out, d_sink, d_src); run<36>(d_out, d_sink, d_src); run<37>(d_out, d_sink, d_src); run<38>(d_out, d_sink, d_src); run<39>(d_out, d_sink, d_src); run<40>(d_out, d_sink, d_src); run<41>(d_out, d_sink, d_src); run<42>(d_out, d_sink, d_src); run<43>(d_out, d_sink, d_src); run<44>(d_out, d_sink, d_src); run<45>(d_out, d_sink, d_src); run<46>(d_out, d_sink, d_src); run<47>(d_out, d_sink, d_src); run<48>(d_out, d_sink, d_src);
    run<49>(d_out, d_sink, d_src); run<50>(d_out, d_sink, d_src); run<51>(d_out, d_sink, d_src); run<52>(d_out, d_sink, d_src); run<53>(d_out, d_sink, d_src); run<54>(d_out, d_sink, d_src); run<55>(d_out, d_sink, d_src); run<56>(d_out, d_sink, d_src);
    hipLaunchKernelGGL(lat, dim3(1), dim3(64), 0, 0, d_out, d_sink, d_chain);
    hipDeviceSynchronize();
    unsigned long long h[4];
    hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
    printf("{\"latency_cycles\": {\"ds_read_b32\": %.1f, \"flat_load_lds\": %.1f, \"global_load_hit\": %.1f, \"s_load_hit\": %.1f}}\n", h[0] / 256.0, h[1] / 256.0,
           h[2] / 256.0, h[3] / 256.0);
    return 0;
}
