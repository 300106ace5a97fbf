// Developer probe: does v_mfma_f32_16x16x32_f16 honour fp16 DENORMAL inputs (and v_cvt_pk_f16_f32 produce them)?
// The fp16x2 operand split of the policy GEMMs may keep its low pieces unscaled only if it does.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k(float* out, const float* in) {
    const float tiny = in[0];   // 2^-20
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.0f; b[i] = (_Float16)0.0f; }
    f2 v = {tiny, 1.0f};
    h2 c = __builtin_convertvector(v, h2);   // v_cvt_pk_f16_f32
    a[0] = c.x;            // A[row][k = 8*(lane>>4)] = 2^-20 (fp16 subnormal)
    b[0] = c.y;            // B[k][col] = 1
    f4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    // and the other way round: subnormal in B
    f4 acc2 = {0, 0, 0, 0};
    acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, acc2, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = acc2[0]; out[2] = (float)c.x; out[3] = tiny * 4.0f; }
}
int main() {
    float *d_out, *d_in, h[4], t = 9.5367431640625e-07f;  // 2^-20
    hipMalloc(&d_out, 16); hipMalloc(&d_in, 4);
    hipMemcpy(d_in, &t, 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_out, d_in);
    hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost);
    printf("{\"mfma_f16_denorm_A\": %.10e, \"mfma_f16_denorm_B\": %.10e, \"cvt_pk_f16_of_2^-20\": %.10e, \"expected_4x2^-20\": %.10e, \"honoured\": %s}\n",
           h[0], h[1], h[2], h[3], (h[0] == h[3] && h[1] == h[3]) ? "true" : "false");
    return 0;
}
