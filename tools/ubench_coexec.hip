// Developer probe: do a matrix-core wave and a vector wave on the SAME SIMD overlap?  512-thread workgroups (2 waves per
// SIMD): waves 0-3 run an MFMA stream, waves 4-7 a VALU stream.  Each role is timed alone and together.
//   mfma kinds: 0 = v_mfma_f32_16x16x32_f16, 1 = v_mfma_f32_16x16x4_f32;  valu kinds: 0 = v_fma_f32 (fast class), 1 = v_pk_fma_f32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define REP 256
#define X4(s) s s s s
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
template <int MK, int VK> __global__ __launch_bounds__(512) void k(unsigned long long* out, float* sink, const float* src, const int run_mfma, const int run_valu) {
    const int wave = threadIdx.x >> 6;
    float a0 = src[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a1, a2}, p3 = {a3, a0};
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    h8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)a0; hb[i] = (_Float16)a1; }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        if (run_mfma)
            for (int r = 0; r < REP; ++r) {
                if constexpr (MK == 0)
                    asm volatile(X4("v_mfma_f32_16x16x32_f16 %0, %4, %5, %0\n v_mfma_f32_16x16x32_f16 %1, %4, %5, %1\n v_mfma_f32_16x16x32_f16 %2, %4, %5, %2\n v_mfma_f32_16x16x32_f16 %3, %4, %5, %3\n")
                                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(ha), "v"(hb));
                else
                    asm volatile(X4("v_mfma_f32_16x16x4_f32 %0, %4, %5, %0\n v_mfma_f32_16x16x4_f32 %1, %4, %5, %1\n v_mfma_f32_16x16x4_f32 %2, %4, %5, %2\n v_mfma_f32_16x16x4_f32 %3, %4, %5, %3\n")
                                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a2), "v"(a3));
            }
    } else {
        if (run_valu)
            for (int r = 0; r < REP; ++r) {
                if constexpr (VK == 0)
                    asm volatile(X4(X4("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %3, %2\n v_fma_f32 %1, %1, %3, %2\n")) : "+v"(a0), "+v"(a1) : "v"(a2), "v"(a3));
                else
                    asm volatile(X4(X4("v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %0, %0, %3, %2\n v_pk_fma_f32 %1, %1, %3, %2\n")) : "+v"(p0), "+v"(p1) : "v"(p2), "v"(p3));
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
    float s = a0 + a1 + p0.x + p1.y + c0[0] + c1[1] + c2[2] + c3[3];
    if (s == 12345.678f) sink[threadIdx.x] = s;
}
template <int MK, int VK> void run(unsigned long long* d_out, float* d_sink, float* d_src, const char* name) {
    double res[3][2];
    int cfg[3][2] = {{1, 0}, {0, 1}, {1, 1}};
    for (int c = 0; c < 3; ++c) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MK, VK>), dim3(256), dim3(512), 0, 0, d_out, d_sink, d_src, cfg[c][0], cfg[c][1]);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256 * 8);
        hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<unsigned long long> m, v;
        for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? m : v).push_back(h[b * 8 + w]);
        std::sort(m.begin(), m.end()); std::sort(v.begin(), v.end());
        res[c][0] = (double)m[m.size() / 2]; res[c][1] = (double)v[v.size() / 2];
    }
    const double nm = REP * 16.0, nv = REP * 64.0;
    printf("{\"pair\": \"%s\", \"mfma_alone_cyc_per_instr\": %.2f, \"valu_alone_cyc_per_instr\": %.2f, \"mfma_together\": %.2f, \"valu_together\": %.2f}\n", name,
           res[0][0] / nm, res[1][1] / nv, res[2][0] / nm, res[2][1] / nv);
}
int main() {
    unsigned long long* d_out; float *d_sink, *d_src;
    hipMalloc(&d_out, 1 << 20); hipMalloc(&d_sink, 1 << 16); hipMalloc(&d_src, 1 << 16);
    std::vector<float> src(4096); for (int i = 0; i < 4096; ++i) src[i] = 0.5f + 0.001f * i;
    hipMemcpy(d_src, src.data(), 4096 * 4, hipMemcpyHostToDevice);
    run<0, 0>(d_out, d_sink, d_src, "mfma_f16 | v_fma_f32");
    run<0, 1>(d_out, d_sink, d_src, "mfma_f16 | v_pk_fma_f32");
    run<1, 0>(d_out, d_sink, d_src, "mfma_f32 | v_fma_f32");
    run<1, 1>(d_out, d_sink, d_src, "mfma_f32 | v_pk_fma_f32");
    return 0;
}
