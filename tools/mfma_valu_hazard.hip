// mfma_valu_hazard.hip -- developer probe (round 3; HISTORICAL -- its conclusion "one instruction in between is enough" was an
// observation on these particular timings, NOT the rule).  The gfx950 rule is TWO wait states between a VALU write of a VGPR and an
// MFMA that reads it: hipcc emits `s_nop 1` between a compiler-generated VALU write and the dependent MFMA, and LLVM's hazard
// recogniser does not see instructions inside inline asm.  Since round 4 the operand split is compiler-generated
// (policy.hpp: split_pair_h) and tests/test_isa_static.py checks the two-wait-state rule statically over every kernel of the library.
// The probe's question: does v_mfma_f32_16x16x32_f16 on gfx950 see a VGPR that a VALU instruction wrote
// N instructions earlier?  (Inline asm is outside the compiler's hazard handling: the operand split of the policy pass writes MFMA
// operands with v_fma_mixlo/mixhi_f16 from inline asm.)  For N = 0..4 independent instructions between the write and the MFMA:
// B operand = all ones (fp16 1.0) except that its last dword is rewritten from `stale` to `fresh` right before the MFMA.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_hazard.hip -o gpurun_out/mfma_valu_hazard && gpurun_out/mfma_valu_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int N, int PRE, int FILL> __global__ void k(float* out, unsigned stale, unsigned fresh, int use_mix) {
    u4 a = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};   // fp16 1.0 pairs
    f4 acc = {0, 0, 0, 0};
    float dummy0 = threadIdx.x, dummy1 = 2.0f;
    const unsigned ones = 0x3c003c00u;
    // the B operand lives in v[100:103] (named registers, so that the rewrite of its last dword and the MFMA's read meet)
    if (use_mix) {
        asm volatile(
            "v_mov_b32 v100, %[o]\n v_mov_b32 v101, %[o]\n v_mov_b32 v102, %[o]\n v_mov_b32 v103, %[st]\n s_nop 7\n s_nop 7\n"
            "v_fma_mixlo_f16 v103, %[f], 1.0, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]\n"
            ".if %c[pre] >= 1\n v_mfma_f32_16x16x32_f16 v[104:107], %[a], v[100:103], 0\n .endif\n"
            ".if %c[pre] >= 2\n v_fma_mixhi_f16 v100, %[f], 1.0, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]\n .endif\n"
            "v_fma_mixhi_f16 v103, %[f], 1.0, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]\n"
            ".rept %c[n]\n .if %c[fill] == 0\n v_add_f32 %[d0], %[d0], %[d1]\n .endif\n .if %c[fill] == 1\n v_med3_f32 %[d0], %[d0], 0, %[d1]\n .endif\n .if %c[fill] == 2\n s_nop 0\n .endif\n .endr\n"
            "v_mfma_f32_16x16x32_f16 %[acc], %[a], v[100:103], 0\n"
            : [acc] "=&v"(acc), [d0] "+v"(dummy0)
            : [a] "v"(a), [o] "v"(ones), [st] "v"(stale), [f] "v"(__uint_as_float(fresh)), [d1] "v"(dummy1), [n] "n"(N), [pre] "n"(PRE), [fill] "n"(FILL)
            : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107");
    } else {
        asm volatile(
            "v_mov_b32 v100, %[o]\n v_mov_b32 v101, %[o]\n v_mov_b32 v102, %[o]\n v_mov_b32 v103, %[st]\n s_nop 7\n s_nop 7\n"
            "v_mov_b32 v103, %[f]\n"
            ".rept %c[n]\n v_add_f32 %[d0], %[d0], %[d1]\n .endr\n"
            "v_mfma_f32_16x16x32_f16 %[acc], %[a], v[100:103], 0\n"
            : [acc] "=&v"(acc), [d0] "+v"(dummy0)
            : [a] "v"(a), [o] "v"(ones), [st] "v"(stale), [f] "v"(fresh), [d1] "v"(dummy1), [n] "n"(N)
            : "v100", "v101", "v102", "v103");
    }
    out[(blockIdx.x * 64 + threadIdx.x) * 4 + 0] = acc[0] + 0 * dummy0;
}

int main() {
    float* d; hipMalloc(&d, 64 * 4 * sizeof(float));
    float h[256];
    // expected: each output = sum over k of a*b; lanes' k slices: the dword b.w carries 2 of the 8 halves of a lane's slice.
    // stale = 0 (two zero halves), fresh = two 1.0 halves: result 32 if fresh is seen by every lane, 24..32 otherwise
    #define RUNF(N, PRE, FILL) for (int mix = 1; mix < 2; ++mix) { hipLaunchKernelGGL((k<N, PRE, FILL>), dim3(1), dim3(64), 0, 0, d, 0u, mix ? 0x3f800000u : 0x3c003c00u, mix); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); \
        float mn = 1e9, mx = -1e9; for (int i = 0; i < 64; ++i) { mn = h[4*i] < mn ? h[4*i] : mn; mx = h[4*i] > mx ? h[4*i] : mx; } \
        printf("{\"filler\": \"%s\", \"independent_mfma_issued_just_before_the_writer\": %d, \"instructions_between\": %d, \"writer\": \"%s\", \"result_min\": %g, \"result_max\": %g, \"fresh_value_seen\": %s}\n", FILL == 0 ? "v_add_f32" : (FILL == 1 ? "v_med3_f32" : "s_nop 0"), PRE, N, mix ? "v_fma_mixlo/mixhi_f16" : "v_mov_b32", mn, mx, (mn == 32 && mx == 32) ? "true" : "false"); }
    RUNF(0, 0, 0) RUNF(1, 0, 0) RUNF(2, 0, 0) RUNF(1, 1, 0) RUNF(1, 2, 0)
    RUNF(1, 0, 1) RUNF(2, 0, 1) RUNF(1, 1, 1) RUNF(2, 1, 1) RUNF(1, 2, 1) RUNF(2, 2, 1) RUNF(3, 2, 1)
    RUNF(1, 0, 2) RUNF(2, 0, 2) RUNF(1, 2, 2) RUNF(2, 2, 2)
    return 0;
}
