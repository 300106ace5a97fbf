// developer probe (round 3; HISTORICAL: see the header of tools/mfma_valu_hazard.hip -- the rule is two wait states, what this
// probe observed with one is timing luck): tools/mfma_valu_hazard.hip's question under CONTENTION -- two waves per SIMD (512 threads), each looping over
//   [independent MFMAs] ; v_fma_mixhi (rewrites the high halves of two dwords of the next MFMA's B operand) ; N fillers ; MFMA ; check
// counts the iterations in which the MFMA did not see the fresh halves.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
template <int N> __global__ __launch_bounds__(512) void k(unsigned* bad, int iters) {
    u4 a = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    float dummy0 = threadIdx.x, dummy1 = 2.0f, onef = 1.0f, r0 = 0.0f;
    const unsigned ones = 0x3c003c00u, lo_only = 0x00003c00u;
    unsigned nbad = 0;
    for (int it = 0; it < iters; ++it) {
        asm volatile(
            "v_mov_b32 v100, %[o]\n v_mov_b32 v101, %[o]\n v_mov_b32 v102, %[o]\n v_mov_b32 v103, %[o]\n"
            "v_mov_b32 v104, %[l]\n v_mov_b32 v105, %[o]\n v_mov_b32 v106, %[o]\n v_mov_b32 v107, %[l]\n"
            "v_mfma_f32_16x16x32_f16 v[112:115], %[a], v[100:103], 0\n"
            "v_mfma_f32_16x16x32_f16 v[116:119], %[a], v[100:103], 0\n"
            "v_mfma_f32_16x16x32_f16 v[100:103], %[a], v[100:103], 0\n"
            "v_fma_mixhi_f16 v104, %[f], 1.0, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]\n"
            "v_fma_mixhi_f16 v107, %[f], 1.0, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]\n"
            ".rept %c[n]\n v_med3_f32 %[d0], %[d0], 0, %[d1]\n .endr\n"
            "v_mfma_f32_16x16x32_f16 v[104:107], %[a], v[104:107], 0\n"
            "s_nop 7\n s_nop 7\n v_mov_b32 %[r0], v104\n"
            : [r0] "=v"(r0), [d0] "+v"(dummy0)
            : [a] "v"(a), [o] "v"(ones), [l] "v"(lo_only), [f] "v"(onef), [d1] "v"(dummy1), [n] "n"(N)
            : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119");
        nbad += r0 != 32.0f;
    }
    if (nbad) atomicAdd(bad, nbad);
    if (dummy0 == 12345.0f) bad[1] = 1;
}
template <int N> void run(unsigned* d) {
    (void)hipMemset(d, 0, 8);
    hipLaunchKernelGGL(k<N>, dim3(256), dim3(512), 0, 0, d, 2000);
    unsigned h[2];
    (void)hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("{\"waves_per_simd\": 2, \"fillers\": %d, \"lane_iterations\": %llu, \"stale_reads\": %u}\n", N, 256ull * 512 * 2000, h[0]);
}
int main() {
    unsigned* d; (void)hipMalloc(&d, 8);
    run<0>(d); run<1>(d); run<2>(d); run<3>(d);
    return 0;
}
