// Developer tool (GPU box): the fp16 x 2 operand split through v_fma_mixlo/mixhi_f16 (residual formed and rounded in one
// instruction from the packed high halves) against the plain form (convert back, subtract, convert) -- bit equality over
// random values of every magnitude, fp16 denormal residuals, zeros, the clamp bounds.  hipcc --offload-arch=gfx950 -O3
// -ffp-contract=off tools/split_mix_check.hip -o build/split_mix_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_f16(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2v));
}
__global__ void k(const float* x, unsigned* o, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = x[2 * i], b = x[2 * i + 1];
    const unsigned p0 = pk_f16(a, b);
    const f16x2v h = __builtin_bit_cast(f16x2v, p0);
    const f32x2 r = (f32x2){a, b} - (f32x2){(float)h.x, (float)h.y};
    const unsigned ref = pk_f16(r.x, r.y);
    unsigned p1;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(p1) : "v"(p0), "v"(a));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(p1) : "v"(p0), "v"(b));
    o[2 * i] = ref;
    o[2 * i + 1] = p1;
}
int main() {
    const int n = 1 << 22;
    std::vector<float> x(2 * n);
    uint64_t s = 88172645463325252ull;
    for (int i = 0; i < 2 * n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const int e = (int)((s >> 40) % 44) - 30;                       // 2^-30 .. 2^13
        const float m = 1.0f + (float)((s >> 8) & 0x7fffff) / 8388608.0f;
        float v = ldexpf(m, e) * ((s & 1) ? -1.0f : 1.0f);
        if ((i & 1023) == 0) v = 0.0f;
        if ((i & 1023) == 1) v = 65504.0f;
        if ((i & 1023) == 2) v = -65504.0f;
        if ((i & 1023) == 3) v = ldexpf(m, -14);                       // residuals deep in the fp16 denormals
        x[i] = v;
    }
    float* dx; unsigned* dout;
    hipMalloc(&dx, sizeof(float) * 2 * n); hipMalloc(&dout, sizeof(unsigned) * 2 * n);
    hipMemcpy(dx, x.data(), sizeof(float) * 2 * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    std::vector<unsigned> o(2 * n);
    hipMemcpy(o.data(), dout, sizeof(unsigned) * 2 * n, hipMemcpyDeviceToHost);
    long bad = 0, neg0 = 0;
    for (int i = 0; i < n; ++i) if (o[2 * i] != o[2 * i + 1]) {
        // -0 vs +0 halves are the same operand value
        const unsigned d = o[2 * i] ^ o[2 * i + 1];
        if ((d & ~0x80008000u) == 0 && ((o[2*i] & 0x7fff) == 0 || !(d & 0x8000)) && (((o[2*i] >> 16) & 0x7fff) == 0 || !(d & 0x80000000u))) { ++neg0; continue; }
        if (bad++ < 5) printf("mismatch at %d: a=%a b=%a ref=%08x mix=%08x\n", i, x[2*i], x[2*i+1], o[2*i], o[2*i+1]);
    }
    printf("{\"pairs\": %d, \"mismatches\": %ld, \"signed_zero_only\": %ld}\n", n, bad, neg0);
    return bad != 0;
}
