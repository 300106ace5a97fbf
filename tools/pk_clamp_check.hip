// Developer probe (not part of the product): does v_pk_fma_f32 honour the clamp modifier on gfx950, and how accurate is v_rcp_f64?
//   hipcc --offload-arch=gfx950 -O2 tools/pk_clamp_check.hip -o /tmp/pk_clamp_check && /tmp/pk_clamp_check
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* a, const float* b, const float* c, float* o, const double* d, double* r) {
    const int i = threadIdx.x;
    f32x2 x = {a[i], a[i] * 2.0f}, y = {b[i], b[i]}, z = {c[i], c[i]}, q;
    asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(q) : "v"(x), "v"(y), "v"(z));
    o[2 * i] = q.x;
    o[2 * i + 1] = q.y;
    r[i] = __builtin_amdgcn_rcp(d[i]);
}
int main() {
    const int n = 8;
    float ha[n] = {1, -1, 0.5f, 3, -100, 1e-3f, 0, 2}, hb[n] = {1, 1, 1, 1, 1, 1, 1, -3}, hc[n] = {0, 0, 0.25f, -2.5f, 0, 0, -1, 0.5f};
    double hd[n] = {3.0, 0.1, 123456.789, 1e-7, -7.0, 0.3333, 2.0, 1e10};
    float *a, *b, *c, *o; double *d, *r;
    hipMalloc(&a, 4 * n); hipMalloc(&b, 4 * n); hipMalloc(&c, 4 * n); hipMalloc(&o, 8 * n); hipMalloc(&d, 8 * n); hipMalloc(&r, 8 * n);
    hipMemcpy(a, ha, 4 * n, hipMemcpyHostToDevice); hipMemcpy(b, hb, 4 * n, hipMemcpyHostToDevice); hipMemcpy(c, hc, 4 * n, hipMemcpyHostToDevice);
    hipMemcpy(d, hd, 8 * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(n), 0, 0, a, b, c, o, d, r);
    float ho[2 * n]; double hr[n];
    hipMemcpy(ho, o, 8 * n, hipMemcpyDeviceToHost); hipMemcpy(hr, r, 8 * n, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) {
        const float e0 = fminf(fmaxf(ha[i] * hb[i] + hc[i], 0.f), 1.f), e1 = fminf(fmaxf(2 * ha[i] * hb[i] + hc[i], 0.f), 1.f);
        printf("pk_fma clamp: %g*%g+%g -> (%g, %g) expect (%g, %g)   rcp_f64(%g) rel err %.3g\n", ha[i], hb[i], hc[i], ho[2 * i], ho[2 * i + 1], e0, e1,
               hd[i], fabs(hr[i] * hd[i] - 1.0));
        bad += ho[2 * i] != e0 || ho[2 * i + 1] != e1;
    }
    printf("%s\n", bad ? "CLAMP NOT HONOURED" : "clamp ok");
    return bad;
}
