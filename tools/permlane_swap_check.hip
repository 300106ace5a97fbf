#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float lane_groups_sum(float t) {
    float a = t, b = t;
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    float s = a + b;
    a = s; b = s;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
__global__ void k(float* o, float* ref) {
    float tv = o[threadIdx.x];
    float s = lane_groups_sum(tv);
    float t = tv; t += __shfl_xor(t, 16, 64); t += __shfl_xor(t, 32, 64);
    o[threadIdx.x] = s; ref[threadIdx.x] = t;
}
int main() {
    float *d, *r; hipMalloc(&d, 256); hipMalloc(&r, 256);
    float h[64], g[64]; for (int i = 0; i < 64; ++i) h[i] = 0.1f * i * i + 1.0f / (i + 3);
    hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, r);
    hipMemcpy(h, d, 256, hipMemcpyDeviceToHost); hipMemcpy(g, r, 256, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 64; ++i) bad += h[i] != g[i];
    printf("permlane swap sums vs shfl_xor sums: %d of 64 differ (%g %g)\n", bad, h[5], g[5]);
    return bad != 0;
}
