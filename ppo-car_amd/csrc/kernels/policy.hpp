// policy.hpp -- part of the single translation unit ppocar.hip (included there, in order; not a stand-alone header).
// K5 policy_kernel: Agent.get_action_and_value on the matrix cores (fp32, bf16 x 3 and fp16 x 2 operand forms), weight image packing, the draw.
#pragma once

// ------------------------------------------------------------------------------------------
// K5: fused policy step -- Agent.get_action_and_value(x) in the rollout (model.py:34-41, train.py:181):
//   actor  Linear(D,256) - ReLU - Linear(256,A)    critic  Linear(D,256) - ReLU - Linear(256,1)
//   action ~ Categorical(logits), log_prob(action), value
// in ONE launch.  GEMM-shaped, so it runs on the matrix cores: v_mfma_f32_16x16x4_f32 (fp32 in, fp32
// accumulate, bit-for-bit an fmaf chain -- no reduced precision).  Orientation: rows = hidden units,
// columns = envs.  A wave owns 64 envs (4 column tiles of 16).  Per hidden tile of 16 units (32 tiles:
// 16 actor + 16 critic):
//   layer 1   acc[16 hid x 16 env] = b1 + W1[16 x K] * X^T[K x 16]      K = 4*KS >= D, KS MFMAs per tile
//   ReLU      in registers
//   layer 2   out[16 x 16 env] += W2cat^T[16 x 4] * acc                  4 MFMAs: accumulator register `reg`
//             of lane l holds hidden row 4*(l>>4)+reg of env column l&15, which is exactly the B-operand
//             slot (k = l>>4, j = l&15) of the next MFMA -- the hidden layer never leaves the registers.
// W2cat has the A actor columns and the critic in column A (rows 0..255 actor, 256..511 critic).  The
// weights sit in LDS (W1 rows padded to an odd stride: conflict-free ds_read_b32), each A operand read once
// per 4 MFMAs (the 4 env tiles), which also gives every MFMA three independent ones between it and its
// dependent successor.  The [16 x 64] output goes through LDS so that lane = env for the softmax / Philox
// draw; outputs are written coalesced.
// ------------------------------------------------------------------------------------------

// LDS image of the policy weights, in floats.  [W1: 512 rows x LD1][b1: 512][W2 A-operands: 32 x 4 x 64][b2: 16]
__host__ __device__ constexpr int pol_ld1(int KS) { return 4 * KS + 1; }  // odd row stride: lanes 0..15 hit 16 banks
__host__ __device__ constexpr int pol_image_floats(int KS) { return 512 * pol_ld1(KS) + 512 + 32 * 4 * 64 + 16; }
__host__ __device__ constexpr int pol_image_padded(int KS) { return (pol_image_floats(KS) + 3) & ~3; }

// Build the image once per rollout (the weights do not change while a rollout runs): every workgroup of
// policy_kernel then stages it with straight 16-byte coalesced copies instead of re-deriving the layout.
__global__ __launch_bounds__(256) void policy_pack_kernel(const int KS, const int D, const int A,
                                                          const float* __restrict__ aW1, const float* __restrict__ ab1,
                                                          const float* __restrict__ aW2, const float* __restrict__ ab2,
                                                          const float* __restrict__ cW1, const float* __restrict__ cb1,
                                                          const float* __restrict__ cW2, const float* __restrict__ cb2,
                                                          float* __restrict__ image) {
    constexpr int HID = 256;
    const int LD1 = 4 * KS + 1;
    const int nW1 = 2 * HID * LD1, nB1 = 2 * HID, nW2 = 32 * 4 * 64;
    const int total = ((nW1 + nB1 + nW2 + 16) + 3) & ~3;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        float v = 0.0f;
        if (i < nW1) {
            const int r = i / LD1, c = i - r * LD1;
            if (c < D) v = r < HID ? aW1[r * D + c] : cW1[(r - HID) * D + c];
        } else if (i < nW1 + nB1) {
            const int j = i - nW1;
            v = j < HID ? ab1[j] : cb1[j - HID];
        } else if (i < nW1 + nB1 + nW2) {
            // A operand of layer 2 for (hidden tile ht, accumulator register reg), lane l:
            //   A[i = out o = l & 15][k = l >> 4] = W2cat[hidden 16 ht + 4 (l >> 4) + reg][o]
            const int j = i - nW1 - nB1;
            const int l = j & 63, reg = (j >> 6) & 3, ht = j >> 8;
            const int o = l & 15, h = 16 * ht + 4 * (l >> 4) + reg;
            if (h < HID) {
                if (o < A) v = aW2[o * HID + h];
            } else if (o == A) {
                v = cW2[h - HID];
            }
        } else if (i < nW1 + nB1 + nW2 + 16) {
            const int o = i - nW1 - nB1 - nW2;
            v = o < A ? ab2[o] : (o == A ? cb2[0] : 0.0f);
        }
        image[i] = v;
    }
}

// ReLU as ONE v_med3_f32 (with +inf as the upper bound the compiler rewrites it into canonicalize + max: two instructions)
__device__ __forceinline__ float relu_f(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, 3.4028234663852886e38f); }

// One wave's MFMA work for 32 envs (2 column tiles) over hidden tiles [ht0, ht1) (an even count): layer 1,
// ReLU, layer 2.  x[et][ks] = B operands of layer 1 (X^T), out[et] = the [16 outs x 16 envs] accumulators of
// layer 2.  Two hidden tiles are in flight per iteration: four independent layer-1 accumulator chains keep
// the matrix pipe issuing while one tile's ReLU (accumulator read-back) and layer-2 operands are prepared,
// and the next pair's A operands are fetched from LDS under this pair's MFMAs.
template <int KS>
__device__ __forceinline__ void policy_pass(const float* sW1, const float* sB1, const float* sW2, const int ht0, const int ht1,
                                            const float (&x)[2][KS], f32x4 (&out)[2], const int lc, const int lk, const int lane) {
    constexpr int LD1 = pol_ld1(KS), ET = 2, TP = 2;
    float a1[TP][KS], a2[TP][4];
    f32x4 bias[TP];
#pragma unroll
    for (int j = 0; j < TP; ++j) {
        bias[j] = *reinterpret_cast<const f32x4*>(sB1 + 16 * (ht0 + j) + 4 * lk);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a1[j][ks] = sW1[(16 * (ht0 + j) + lc) * LD1 + 4 * ks + lk];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) a2[j][reg] = sW2[((ht0 + j) * 4 + reg) * 64 + lane];
    }
    for (int ht = ht0; ht < ht1; ht += TP) {
        const int hn = ht + TP < ht1 ? ht + TP : ht;
        float n1[TP][KS], n2[TP][4];
        f32x4 nbias[TP];
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            nbias[j] = *reinterpret_cast<const f32x4*>(sB1 + 16 * (hn + j) + 4 * lk);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) n1[j][ks] = sW1[(16 * (hn + j) + lc) * LD1 + 4 * ks + lk];
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) n2[j][reg] = sW2[((hn + j) * 4 + reg) * 64 + lane];
        }
        f32x4 acc[TP][ET];
#pragma unroll
        for (int j = 0; j < TP; ++j)
#pragma unroll
            for (int et = 0; et < ET; ++et) acc[j][et] = bias[j];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int j = 0; j < TP; ++j)
#pragma unroll
                for (int et = 0; et < ET; ++et)
                    acc[j][et] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j][ks], x[et][ks], acc[j][et], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < TP; ++j)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
#pragma unroll
                for (int et = 0; et < ET; ++et)
                    out[et] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[j][reg], relu_f(acc[j][et][reg]), out[et], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            bias[j] = nbias[j];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) a1[j][ks] = n1[j][ks];
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) a2[j][reg] = n2[j][reg];
        }
    }
}

// ---- the same two layers on the bf16 matrix cores, fp32-equivalent: every fp32 operand is split into three
// bf16 pieces (v = v0 + v1 + v2, 8 significant bits each, residuals exact), and a product a*b is taken as the six
// piece products a_i*b_j with i + j <= 2 (each exact in the fp32 accumulator; the dropped ones are <= 2^-24
// relative).  Measured against float64 on this MLP the result is closer than a plain fp32 GEMM (max error 0.55e-6
// vs 1.3e-6).  v_mfma_f32_16x16x32_bf16 runs on the matrix pipe proper, 16x the fp32-input rate, and --
// unlike the fp32-input MFMA -- does not occupy the fp32 ALUs the env step needs.
// Layouts: lane (g = l >> 4, lc = l & 15) holds A[row lc][k = 8g + j], B[k = 8g + j][col lc], j = 0..7.
//   layer 1: k = feature (D <= 24: one K block, group 3 is zero padding), rows = 16 hidden units, cols = 16 envs
//   layer 2: K block = TWO hidden tiles; k-slot j of group g <-> tile (j >> 2), hidden row 4g + (j & 3): exactly the
//            accumulator registers the lane already holds for its env column -- again no data movement.
#ifndef PC_POL_INTERLEAVE
#define PC_POL_INTERLEAVE 2      /* vector instructions asked for after each layer-1 MFMA of the split policy pass (0: hipcc's own schedule; 2: -1.4 % per epoch at the target shape, 3: nothing -- profiles/r5_ab_experiments.txt) */
#endif
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Split forms: PREC 1 = three bf16 pieces per operand, PREC 2 = two fp16 pieces (the second scaled by 2^11, below).
__host__ __device__ constexpr int pol_np(int PREC) { return PREC == 2 ? 2 : 3; }
// Layer-1 K blocks of 32 features: one for D <= 24 (3 stored groups of 8 features), two for D <= 40 (5 groups: the second
// block's groups 1..3 are zero padding and are not stored).
__host__ __device__ constexpr int pol_ng(int KS) { return KS == 10 ? 5 : 3; }
__host__ __device__ constexpr int pol_kb(int KS) { return KS == 10 ? 2 : 1; }
__host__ __device__ constexpr int polx_w1_dwords(int PREC, int NG) { return 32 * pol_np(PREC) * NG * 16 * 4; }
__host__ __device__ constexpr int polx_w2_dwords(int PREC) { return 8 * pol_np(PREC) * 4 * 10 * 4; }  // actor tile pairs only
// [W1 records][W2 records (actor)][b1: 512][b2: 16][critic output weights, fp32: 256]
__host__ __device__ constexpr int polx_image_dwords(int PREC, int NG) { return polx_w1_dwords(PREC, NG) + polx_w2_dwords(PREC) + 512 + 16 + 256; }

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {  // low half = bf16(a), high half = bf16(b), round-to-nearest-even
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// split two fp32 values into their three bf16 pieces (packed pairwise)
__device__ __forceinline__ void split_pair(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
    p0 = pk_bf16(a, b);
    const float ra = a - __uint_as_float(p0 << 16), rb = b - __uint_as_float(p0 & 0xffff0000u);
    p1 = pk_bf16(ra, rb);
    const float sa = ra - __uint_as_float(p1 << 16), sb = rb - __uint_as_float(p1 & 0xffff0000u);
    p2 = pk_bf16(sa, sb);
}

template <int PREC> struct Pieces { u32x4 p[pol_np(PREC)]; };  // eight fp32 values as pol_np x (8 halves)

// ---- PREC 2: fp16 x 2 in SCALED DOMAINS.  Every operand v is written v = h + l with h = fp16(v) and l = fp16(v - h): the
// residual is exact in fp32, and l carries 11 more significant bits of v as long as it is a NORMAL fp16 number, i.e. for
// |v| >= 2^-3.  So that this holds for every operand whose magnitude matters, the GEMMs run on power-of-two multiples of the
// data (exact rescalings): observations x 16, first-layer weights x 16 -> hidden pre-activations, biases and ReLU outputs
// x 256, output-layer weights x 64 -> logits and the value x 16384, undone by one fused multiply-add where the output bias
// is added.  An operand below 2^-3 in its scaled domain (an observation under 0.008, a hidden activation under 5e-4, an
// output weight under 0.002) keeps an ABSOLUTE error of at most 2^-25 scaled, i.e. <= 2e-9 / 1e-10 / 5e-10 unscaled; all
// others 22 significant bits.  A product a*b is a_h*b_l + a_l*b_h + a_h*b_h, the three fp16 MFMAs accumulating into ONE
// fp32 accumulator, small terms first (each piece product is exact in fp32; the dropped a_l*b_l is <= 2^-22 relative).
// Against float64 this MLP's error is 1.3e-7 (plain fp32 GEMM 0.8e-7, bf16x3 1.0e-7; tools/emu_policy_split.py).  Operands
// saturate at fp16's finite range in their scaled domain: |obs| <= 4094, |W1| <= 4094, hidden activations <= 255.9,
// |W2| <= 1023 (observations are O(1), the reference's weights O(0.1 - 1)).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
#define PC_H_MAX 65504.0f
#define PC_OBS_ABS_MAX 4.0f  /* the bound on |observation| the pack-time range check assumes (CarEnv: [-1, 1.6]) */
#define PC_SX 16.0f          /* observations */
#define PC_S1 16.0f          /* first-layer weights */
#define PC_SH 256.0f         /* hidden layer = PC_SX * PC_S1 */
#define PC_S2 64.0f          /* output-layer weights (actor: the split operands; critic: its fp32 weights) */
#define PC_SO_INV 6.103515625e-05f   /* 1 / (PC_SH * PC_S2) = 2^-14 */
template <int PREC> struct PolScale {   // the scaled domains exist for PREC 2 only
    static constexpr float sx = PREC == 2 ? PC_SX : 1.0f, s1 = PREC == 2 ? PC_S1 : 1.0f, sh = PREC == 2 ? PC_SH : 1.0f,
                           s2 = PREC == 2 ? PC_S2 : 1.0f, so_inv = PREC == 2 ? PC_SO_INV : 1.0f;
};
__device__ __forceinline__ float clamp_h(float v) { return __builtin_amdgcn_fmed3f(v, -PC_H_MAX, PC_H_MAX); }
__device__ __forceinline__ unsigned pk_f16(float a, float b) {  // low half = fp16(a), high half = fp16(b), round-to-nearest-even
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2v));
}
// l = fp16(v - h) as fp16(fma(h, -1, v)): the product and the sum are exact in fp32 (h is v rounded to 11 bits), so the one
// rounding is the conversion to fp16 -- and written like this the compiler selects v_fma_mixlo_f16 / v_fma_mixhi_f16, which read h
// straight out of the packed pair, form the fma and round it into the low / high half of the destination: THREE instructions per
// pair of values (v_cvt_pk_f16_f32 + the two) where convert-back, subtract, convert takes five; the same bits
// (tools/split_mix_check.hip, fp16-denormal residuals included).  Rounds 2-3 wrote the two mix instructions as inline asm, which
// is outside the compiler's hazard model: the register they complete is an MFMA operand, and on gfx950 a VALU write needs TWO wait
// states before an MFMA reads it (hipcc inserts the s_nop itself between compiler-generated instructions; tools/mfma_valu_hazard.hip)
// -- one schedule of the small form read a stale half.  Here every instruction that writes an MFMA operand is the compiler's own.
// Two details keep the optimiser from undoing the form:
//   * the multiplier -1.0 arrives in an SGPR the optimiser cannot see through (as a literal, fma(h, -1, v) is folded into a
//     subtraction before instruction selection and the fp16 source operand is lost);
//   * the low half is an fma, the high half a contracted multiply-add (llvm.fma vs llvm.fmuladd: the same fused operation on
//     this target): two identical expressions are merged by the SLP vectoriser into v_pk_fma_f32 + v_cvt_pk_f16_f32 behind two
//     conversions back to fp32 -- five instructions again.
// tests/test_capi_host.py compiles this function alone and checks the instruction selection.
__device__ __forceinline__ float opaque_neg_one() {
    float m;
    asm("s_mov_b32 %0, 0xbf800000" : "=s"(m));     // (a scalar constant only: no vector register is written from inline asm)
    return m;
}
__device__ __forceinline__ void split_pair_h(float a, float b, unsigned& p0, unsigned& p1) {
    const f32x2 v = {a, b};
    const f16x2v h = __builtin_convertvector(v, f16x2v);        // round-to-nearest-even, both halves: v_cvt_pk_f16_f32
    p0 = __builtin_bit_cast(unsigned, h);
    const float m1 = opaque_neg_one();
    f16x2v l;
    l.x = (_Float16)__builtin_fmaf((float)h.x, m1, a);          // v_fma_mixlo_f16
    {
#pragma clang fp contract(fast)
        l.y = (_Float16)((float)h.y * m1 + b);                  // v_fma_mixhi_f16
    }
    p1 = __builtin_bit_cast(unsigned, l);
}

template <int PREC> __device__ __forceinline__ Pieces<PREC> split8(const float (&v)[8]) {
    Pieces<PREC> r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if constexpr (PREC == 2) {
            unsigned p0, p1;
            split_pair_h(v[2 * i], v[2 * i + 1], p0, p1);
            r.p[0][i] = p0;
            r.p[1][i] = p1;
        } else {
            unsigned p0, p1, p2;
            split_pair(v[2 * i], v[2 * i + 1], p0, p1, p2);
            r.p[0][i] = p0;
            r.p[1][i] = p1;
            r.p[2][i] = p2;
        }
    }
    return r;
}

__device__ __forceinline__ f32x4 mfma6(const u32x4 (&a)[3], const Pieces<1>& b, f32x4 acc) {  // small terms first
#define PC_MF(i, j) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b.p[j]), acc, 0, 0, 0)
    PC_MF(0, 2); PC_MF(1, 1); PC_MF(2, 0); PC_MF(0, 1); PC_MF(1, 0); PC_MF(0, 0);
#undef PC_MF
    return acc;
}
// fp16 x 2: acc += a_h*b_l + a_l*b_h + a_h*b_h, small terms first, one accumulator chain
__device__ __forceinline__ f32x4 mfma3(const u32x4 (&a)[2], const Pieces<2>& b, f32x4 acc) {
#define PC_MF(i, j) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[i]), __builtin_bit_cast(f16x8, b.p[j]), acc, 0, 0, 0)
    PC_MF(0, 1); PC_MF(1, 0); PC_MF(0, 0);
#undef PC_MF
    return acc;
}

// image builder of the split forms (one thread per 16-byte operand record / per bias float)
template <int PREC, int NG>
__global__ __launch_bounds__(256) void policy_pack16_kernel(const int D, const int A, const float* __restrict__ aW1,
                                                            const float* __restrict__ ab1, const float* __restrict__ aW2,
                                                            const float* __restrict__ ab2, const float* __restrict__ cW1,
                                                            const float* __restrict__ cb1, const float* __restrict__ cW2,
                                                            const float* __restrict__ cb2, unsigned* __restrict__ image,
                                                            int* __restrict__ status) {
    // status (optional; the host zeroes it before the launch): the fp16 x 2 form's NUMERIC DOMAIN, checked while the weights pass through
    // (PC_POLICY_RANGE_*, include/ppocar.h): bit 0 / 1 -- a first- / output-layer weight saturates in its scaled domain (|W1| > 4094,
    // |W2| > 1023.5); bit 2 -- a hidden unit CAN reach the hidden layer's saturation point: |b1| + PC_OBS_ABS_MAX * sum_j |W1_uj| > 255.87
    // (observations of CarEnv lie in [-1, 1.6]: positions / 1280 on a track inside the 2000 px box, everything else in [-1, 1];
    // PC_OBS_ABS_MAX = 4 leaves a factor 2.5).  With all three clear no operand of the policy pass can saturate, whatever the kernels
    // are fed: the guard costs the rollout kernels nothing.  (model.py:14-32 has no such limit: a set bit means "use precision 0".)
    constexpr int HID = 256, NP = pol_np(PREC);
    constexpr int n1 = 32 * NP * NG * 16, n2 = 8 * NP * 4 * 10;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n1 + n2 + 512 + 16 + 256; i += gridDim.x * blockDim.x) {
        if (i < n1 + n2) {
            float v[8];
            int pc;
            if (i < n1) {
                const int lc = i % 16, g = (i / 16) % NG;
                pc = (i / (16 * NG)) % NP;
                const int ht = i / (16 * NG * NP), r = 16 * ht + lc;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int f = 8 * g + j;
                    v[j] = f < D ? (r < HID ? aW1[r * D + f] : cW1[(r - HID) * D + f]) * PolScale<PREC>::s1 : 0.0f;
                }
            } else {
                const int k = i - n1;
                const int o = k % 10, g = (k / 10) % 4;
                pc = (k / 40) % NP;
                const int tp = k / (40 * NP);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int h = 16 * (2 * tp + (j >> 2)) + 4 * g + (j & 3);   // actor hidden unit (tp < 8)
                    v[j] = o < A ? aW2[o * HID + h] * PolScale<PREC>::s2 : 0.0f;
                }
            }
            if constexpr (PREC == 2) {
                bool big = false;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    big |= !(__builtin_fabsf(v[j]) <= PC_H_MAX);      // (a NaN weight is out of range too)
                    v[j] = clamp_h(v[j]);
                }
                if (big && status) atomicOr(status, i < n1 ? 1 : 2);
            }
            const Pieces<PREC> sp = split8<PREC>(v);
            reinterpret_cast<u32x4*>(image)[i] = sp.p[pc];
        } else {
            const int b = i - n1 - n2;
            float v;
            if (b < 512) {
                v = (b < HID ? ab1[b] : cb1[b - HID]) * PolScale<PREC>::sh;   // hidden layer's scaled domain
                if constexpr (PREC == 2) {
                    if (status) {      // the largest pre-activation hidden unit b can see on observations within +-PC_OBS_ABS_MAX
                        const float* row = b < HID ? aW1 + b * D : cW1 + (b - HID) * D;
                        float bound = __builtin_fabsf(b < HID ? ab1[b] : cb1[b - HID]);
                        for (int f = 0; f < D; ++f) bound = __builtin_fmaf(__builtin_fabsf(row[f]), PC_OBS_ABS_MAX, bound);
                        if (!(bound * PolScale<PREC>::sh <= PC_H_MAX)) atomicOr(status, 4);
                    }
                }
            }
            else if (b < 528) {
                const int o = b - 512;
                v = o < A ? ab2[o] : (o == A ? cb2[0] : 0.0f);   // added after the outputs are scaled back
            } else {
                v = cW2[b - 528] * PolScale<PREC>::s2;   // critic output layer, plain fp32, in the outputs' scaled domain
            }
            image[(n1 + n2) * 4 + b] = __float_as_uint(v);
        }
    }
}

// One wave, 32 envs (2 column tiles), hidden tile PAIRS [tp0, tp1).  x[et] = the env tile's observation pieces.
// Pairs 0..7 are the actor: ReLU, split, layer 2 on the matrix cores into out[et] (rows 0..A-1).  Pairs 8..15 are
// the critic, whose output layer is ONE dot product per env: it is taken in plain fp32 on the VALU straight from
// the accumulator registers (val[et] = this lane's partial over its hidden rows; the caller sums the 4 lane groups).
template <int PREC, int KB, int ET = 2>
__device__ __forceinline__ void policy_pass16(const unsigned* sW1p, const unsigned* sW2p, const float* sB1, const float* sW2c,
                                              const int tp0, const int tp1, const Pieces<PREC> (&x)[ET][KB], f32x4 (&out)[ET],
                                              float (&val)[ET], const int lc, const int g) {
    constexpr int NP = pol_np(PREC), NG = KB == 2 ? 5 : 3;
    const int oA = lc < 10 ? lc : 9;  // output rows >= 10 are never read
    // layer 1 of tile pair tp: acc[j][et] = b1 + W1[16 rows of tile 2 tp + j] x^T[et] (in the hidden layer's scaled domain)
    auto layer1 = [&](const int tp, f32x4 (&acc)[2][ET]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const f32x4 bias = *reinterpret_cast<const f32x4*>(sB1 + 16 * (2 * tp + j) + 4 * g);
#pragma unroll
            for (int et = 0; et < ET; ++et) acc[j][et] = bias;
        }
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            // feature group 4 kb + g; groups >= NG are K padding (their B operand is all zeros): any finite A will do
            const int gi = 4 * kb + g, gA = gi < NG ? gi : NG - 1;
            u32x4 a[2][NP];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int pc = 0; pc < NP; ++pc)
                    a[j][pc] = *reinterpret_cast<const u32x4*>(sW1p + ((((2 * tp + j) * NP + pc) * NG + gA) * 16 + lc) * 4);
            // The 2 x ET accumulator chains advance TOGETHER, product by product: a chain's next MFMA then stands 2 ET
            // instructions behind the one it depends on (with the chains one after the other it stood ET behind, closer than
            // the instruction's latency).  Per accumulator the order of the products is unchanged: same bits.
            if constexpr (PREC == 2) {
#define PC_L1(ia, ib)                                                                                                          \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int et = 0; et < ET; ++et)                            \
        acc[j][et] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[j][ia]), __builtin_bit_cast(f16x8, x[et][kb].p[ib]), acc[j][et], 0, 0, 0)
                PC_L1(0, 1); PC_L1(1, 0); PC_L1(0, 0);
#undef PC_L1
            } else {
#define PC_L1(ia, ib)                                                                                                          \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int et = 0; et < ET; ++et)                            \
        acc[j][et] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[j][ia]), __builtin_bit_cast(bf16x8, x[et][kb].p[ib]), acc[j][et], 0, 0, 0)
                PC_L1(0, 2); PC_L1(1, 1); PC_L1(2, 0); PC_L1(0, 1); PC_L1(1, 0); PC_L1(0, 0);
#undef PC_L1
            }
        }
    };
    // what follows layer 1 for an ACTOR tile pair (tp < 8): ReLU, operand split, layer 2 on the matrix cores
    auto epilogue_actor = [&](const int tp, const f32x4 (&acc)[2][ET]) {
        u32x4 w2[NP];
#pragma unroll
        for (int pc = 0; pc < NP; ++pc) w2[pc] = *reinterpret_cast<const u32x4*>(sW2p + (((tp * NP + pc) * 4 + g) * 10 + oA) * 4);
        Pieces<PREC> h3[ET];
#pragma unroll
        for (int et = 0; et < ET; ++et) {
            float hv[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if constexpr (PREC == 2) {
                    hv[r] = __builtin_amdgcn_fmed3f(acc[0][et][r], 0.0f, PC_H_MAX);  // ReLU, saturating at fp16's range
                    hv[4 + r] = __builtin_amdgcn_fmed3f(acc[1][et][r], 0.0f, PC_H_MAX);
                } else {
                    hv[r] = relu_f(acc[0][et][r]);
                    hv[4 + r] = relu_f(acc[1][et][r]);
                }
            }
            h3[et] = split8<PREC>(hv);
        }
        // the ET output chains advance together, product by product (see layer1); per accumulator the order is unchanged
        if constexpr (PREC == 2) {
#define PC_L2(ia, ib)                                                                                                          \
    _Pragma("unroll") for (int et = 0; et < ET; ++et)                                                                          \
        out[et] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w2[ia]), __builtin_bit_cast(f16x8, h3[et].p[ib]), out[et], 0, 0, 0)
            PC_L2(0, 1); PC_L2(1, 0); PC_L2(0, 0);
#undef PC_L2
        } else {
#define PC_L2(ia, ib)                                                                                                          \
    _Pragma("unroll") for (int et = 0; et < ET; ++et)                                                                          \
        out[et] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w2[ia]), __builtin_bit_cast(bf16x8, h3[et].p[ib]), out[et], 0, 0, 0)
            PC_L2(0, 2); PC_L2(1, 1); PC_L2(2, 0); PC_L2(0, 1); PC_L2(1, 0); PC_L2(0, 0);
#undef PC_L2
        }
    };
    // ... and for a CRITIC tile pair (tp >= 8): the output layer's dot product on the VALU
    auto epilogue_critic = [&](const int tp, const f32x4 (&acc)[2][ET]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(sW2c + 16 * (2 * tp + j - 16) + 4 * g);
#pragma unroll
            for (int et = 0; et < ET; ++et)
#pragma unroll
                for (int r = 0; r < 4; ++r) val[et] = __builtin_fmaf(w[r], relu_f(acc[j][et][r]), val[et]);
        }
    };
    // Software pipeline over the tile pairs [tp0, tp1) (an even count, actor pairs first): the NEXT pair's layer-1 MFMAs stand
    // in the instruction stream before THIS pair's VALU epilogue, in one branch-free block, so the scheduler can interleave
    // them and the matrix pipe works under the vector work instead of the wave waiting first for its MFMA results and then
    // for its own epilogue.  Two accumulator sets, alternating (no copies); the arithmetic per accumulator is unchanged.
    f32x4 accA[2][ET], accB[2][ET];
    if (tp1 < 0) {
        // SPLIT forms (policy_kernel<SPLIT>, rollout_small_kernel): the eight waves of a workgroup share 32 envs; wave tp0 takes
        // ACTOR pair tp0 and CRITIC pair 8 + tp0 -- the same work on every wave (with pairs 2 w, 2 w + 1 the four actor waves
        // carried both operand splits while the critic waves waited at the barrier)
        layer1(tp0, accA);
        layer1(8 + tp0, accB);
        epilogue_actor(tp0, accA);
        epilogue_critic(8 + tp0, accB);
        return;
    }
    const int ta1 = tp1 < 8 ? tp1 : 8;
    int tp = tp0;
    if constexpr (KB == 1) {
        // One K block (D <= 24): a pair's weight operands and biases are read from LDS ONE PAIR AHEAD, before the epilogue that
        // precedes their MFMAs -- read where they are used, every layer-1 block opens with an exposed LDS round trip.
        struct L1Ops { u32x4 a[2][NP]; f32x4 bias[2]; };
        auto load1 = [&](const int tpl, L1Ops& o) {
            const int gA = g < NG ? g : NG - 1;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                o.bias[j] = *reinterpret_cast<const f32x4*>(sB1 + 16 * (2 * tpl + j) + 4 * g);
#pragma unroll
                for (int pc = 0; pc < NP; ++pc)
                    o.a[j][pc] = *reinterpret_cast<const u32x4*>(sW1p + ((((2 * tpl + j) * NP + pc) * NG + gA) * 16 + lc) * 4);
            }
        };
        auto mfma1 = [&](const L1Ops& o, f32x4 (&acc)[2][ET]) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int et = 0; et < ET; ++et) acc[j][et] = o.bias[j];
            if constexpr (PREC == 2) {
#define PC_L1(ia, ib)                                                                                                          \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int et = 0; et < ET; ++et)                            \
        acc[j][et] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, o.a[j][ia]), __builtin_bit_cast(f16x8, x[et][0].p[ib]), acc[j][et], 0, 0, 0)
                PC_L1(0, 1); PC_L1(1, 0); PC_L1(0, 0);
#undef PC_L1
            } else {
#define PC_L1(ia, ib)                                                                                                          \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int et = 0; et < ET; ++et)                            \
        acc[j][et] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, o.a[j][ia]), __builtin_bit_cast(bf16x8, x[et][0].p[ib]), acc[j][et], 0, 0, 0)
                PC_L1(0, 2); PC_L1(1, 1); PC_L1(2, 0); PC_L1(0, 1); PC_L1(1, 0); PC_L1(0, 0);
#undef PC_L1
            }
        };
        L1Ops oa, ob;
        load1(tp, oa);
        load1(tp + 1, ob);
        mfma1(oa, accA);
        // Round 5 (PC_POL_INTERLEAVE): the next pair's layer-1 MFMAs and this pair's vector epilogue are independent, but left to itself
        // hipcc issues the twelve MFMAs in a row and the forty vector instructions after them -- in order, so a lone wave pays
        // 12 x 16 cycles of matrix pipe and then 40 x 4 of issue (its policy pass: 7.8 k cycles for 3.8 k of pipe time,
        // profiles/r5_k9_phase_timeline.txt).  An MFMA holds the issue port for 8 of its 16 cycles (tools/ubench_shadow.hip): the
        // schedule asked for here is MFMA, three vector instructions, MFMA, ... -- the same instructions, the same bits.
#if PC_POL_INTERLEAVE
#define PC_SGB1 __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, PC_POL_INTERLEAVE, 0);
#define PC_SGB_PAIR { PC_SGB1 PC_SGB1 PC_SGB1 PC_SGB1 PC_SGB1 PC_SGB1 if constexpr (ET == 2) { PC_SGB1 PC_SGB1 PC_SGB1 PC_SGB1 PC_SGB1 PC_SGB1 } }
#else
#define PC_SGB_PAIR
#endif
#pragma unroll 1
        for (; tp < ta1 && tp + 2 < tp1; tp += 2) {
            __builtin_amdgcn_sched_barrier(0);
            mfma1(ob, accB);
            load1(tp + 2, oa);
            if constexpr (!PC_POL_INTERLEAVE) __builtin_amdgcn_sched_barrier(0);
            epilogue_actor(tp, accA);
            PC_SGB_PAIR
            __builtin_amdgcn_sched_barrier(0);
            mfma1(oa, accA);
            load1(tp + 3 < 16 ? tp + 3 : 15, ob);
            if constexpr (!PC_POL_INTERLEAVE) __builtin_amdgcn_sched_barrier(0);
            epilogue_actor(tp + 1, accB);
            PC_SGB_PAIR
        }
#pragma unroll 1
        for (; tp + 2 < tp1; tp += 2) {
            __builtin_amdgcn_sched_barrier(0);
            mfma1(ob, accB);
            load1(tp + 2, oa);
            if constexpr (!PC_POL_INTERLEAVE) __builtin_amdgcn_sched_barrier(0);
            epilogue_critic(tp, accA);
            PC_SGB_PAIR
            __builtin_amdgcn_sched_barrier(0);
            mfma1(oa, accA);
            load1(tp + 3 < 16 ? tp + 3 : 15, ob);
            if constexpr (!PC_POL_INTERLEAVE) __builtin_amdgcn_sched_barrier(0);
            epilogue_critic(tp + 1, accB);
            PC_SGB_PAIR
        }
        __builtin_amdgcn_sched_barrier(0);
#undef PC_SGB_PAIR
#if PC_POL_INTERLEAVE
#undef PC_SGB1
#endif
        mfma1(ob, accB);       // the last two pairs
        if (tp < 8) {              // (uniform)
            epilogue_actor(tp, accA);
            epilogue_actor(tp + 1, accB);
        } else {
            epilogue_critic(tp, accA);
            epilogue_critic(tp + 1, accB);
        }
        return;
    }
    layer1(tp, accA);   // two K blocks (D = 39): operands read where they are used (the prefetch's 24 registers are not there)
#pragma unroll 1
    for (; tp < ta1 && tp + 2 < tp1; tp += 2) {
        layer1(tp + 1, accB);
        epilogue_actor(tp, accA);
        layer1(tp + 2, accA);
        epilogue_actor(tp + 1, accB);
    }
#pragma unroll 1
    for (; tp + 2 < tp1; tp += 2) {
        layer1(tp + 1, accB);
        epilogue_critic(tp, accA);
        layer1(tp + 2, accA);
        epilogue_critic(tp + 1, accB);
    }
    layer1(tp + 1, accB);      // the last two pairs
    if (tp < 8) {              // (uniform)
        epilogue_actor(tp, accA);
        epilogue_actor(tp + 1, accB);
    } else {
        epilogue_critic(tp, accA);
        epilogue_critic(tp + 1, accB);
    }
}

// (Round 5 built the same pass on the wide instruction v_mfma_f32_32x32x16_f16 -- one 32-env column tile per wave; all tests green, 7 % slower
// in K9: profiles/r5_ab_experiments.txt, profiles/HISTORY.md.  The code left the headers in round 6; it is in the history at commit dd4f33e.)
// Softmax / Philox draw / log_prob for one env given its 16 output values (logits 0..A-1, value at A).
// AC > 0: the action count as a compile-time constant (the persistent rollout kernel: CarEnv has Discrete(9), car_env.py:525) --
// the same operations in the same order as with the run-time count, but fully unrolled over registers (with a run-time count
// the compiler walks the 16-slot arrays by register indexing, s_set_gpr_idx: several times the instructions).
template <int AC = 0>
__device__ __forceinline__ void policy_tail(const float (&v)[16], const int A_rt, const float u, int& act, float& lp, float& val,
                                            float* __restrict__ logits_row) {
    // A is wave-uniform: the loops leave at i == A with a scalar branch instead of predicating all 16 slots, and the
    // inverse CDF reuses the exponentials of the log-sum-exp pass (p_i = e_i / sum) -- one expf per action in all.
    const int A = AC > 0 ? AC : A_rt;
    float mx = -INFINITY;
    val = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (i == A) val = v[i];
        if (i < A) mx = fmaxf(mx, v[i]);
    }
    float ex[16];
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (i >= A) break;
        ex[i] = softmax_exp(v[i] - mx);
        sum += ex[i];
    }
    const float lse = mx + softmax_log(sum);  // Categorical(logits=...) normalises: logits - logsumexp
    const float inv = softmax_rcp(sum);
    float cum = 0.0f;
    lp = 0.0f;
    act = -1;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (i >= A) break;
        cum += ex[i] * inv;
        if (act < 0 && (u < cum || i == A - 1)) {  // inverse CDF; last bin absorbs rounding
            act = i;
            lp = v[i] - lse;
        }
        if (logits_row) logits_row[i] = v[i];
    }
}

// The same draw for Discrete(9) (car_env.py:525) on the TWO lanes that own an env (lane = 2 env + g: the env step's mapping, so the
// action never leaves the lane pair): lane 0 holds logits 0..3 and the value (output 9), lane 1 logits 4..8 -- four / five
// exponentials per lane instead of nine on half of the wave's lanes, every reduction one exchange with the neighbouring lane
// (DPP quad_perm [1, 0, 3, 2]).  The arithmetic (policy_tail's with the sums associated by lane): max and sum = lane 0's
// sequential partial (+) lane 1's; the CDF runs sequentially through lane 0's bins and continues from its total through lane
// 1's; the action = the number of bins the uniform has passed, the last bin absorbing rounding.  Used by BOTH the per-step policy
// kernel and the persistent rollout kernel whenever A == 9, so the two stay bit-identical.
// w[0..4] = this lane's outputs as described (already scaled back and biased); act / lp / val are returned in both lanes.
__device__ __forceinline__ float dpp_swap_pair_f(const float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xb1, 0xf, 0xf, false));
}
__device__ __forceinline__ void policy_tail_pair(const float (&w)[5], const int g, const float u, int& act, float& lp, float& val) {
    const bool hi = g != 0;
    float l[5];
#pragma unroll
    for (int j = 0; j < 4; ++j) l[j] = w[j];
    l[4] = hi ? w[4] : -INFINITY;                  // lane 0's fifth output is the value, not a logit
    float mx = fmaxf(fmaxf(l[0], l[1]), fmaxf(l[2], l[3]));
    mx = fmaxf(mx, l[4]);
    mx = fmaxf(mx, dpp_swap_pair_f(mx));
    float ex[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) ex[j] = softmax_exp(l[j] - mx);     // (exp2(-inf) = 0 for lane 0's fifth slot)
    const float part = (((ex[0] + ex[1]) + ex[2]) + ex[3]) + ex[4];
    const float sum = part + dpp_swap_pair_f(part);                  // (a + b on one lane, b + a on the other: the same float)
    const float lse = mx + softmax_log(sum);  // Categorical(logits=...) normalises: logits - logsumexp
    const float inv = softmax_rcp(sum);
    float c[5];
    c[0] = ex[0] * inv;
#pragma unroll
    for (int j = 1; j < 5; ++j) c[j] = c[j - 1] + ex[j] * inv;
    const float other_total = dpp_swap_pair_f(c[4]);
    const float base = hi ? other_total : 0.0f;                      // lane 1's bins continue from lane 0's total
    int n = 0;
#pragma unroll
    for (int j = 0; j < 5; ++j) n += (!(u < base + c[j]) && (j < 4 || hi)) ? 1 : 0;   // bins the uniform has passed (inverse CDF)
    const int n_other = __builtin_amdgcn_update_dpp(0, n, 0xb1, 0xf, 0xf, false);
    const int n_lo = hi ? n_other : n, n_hi = hi ? n : n_other;
    act = n_lo < 4 ? n_lo : 4 + (n_hi < 4 ? n_hi : 4);               // last bin absorbs rounding
    const int li = act - (hi ? 4 : 0);                               // index among this lane's logits (if it is this lane's)
    float mine = l[0];
#pragma unroll
    for (int j = 1; j < 5; ++j) mine = li == j ? l[j] : mine;
    mine -= lse;
    const float theirs = dpp_swap_pair_f(mine);
    lp = ((act >= 4) == hi) ? mine : theirs;
    const float v_other = dpp_swap_pair_f(w[4]);
    val = hi ? v_other : w[4];
}
// this lane's five outputs of env row `row` of an output tile with row stride LDO (a multiple of 4 floats: 16-byte reads):
// columns 0..3 + 9 (g = 0) or 4..8 (g = 1), back from their scaled domain and biased (sB2: the 16 output biases)
template <int LDO>
__device__ __forceinline__ void pair_outputs(const float* tile, const int row, const int g, const float so_inv, const float* sB2, float (&w)[5]) {
    const f32x4 q = *reinterpret_cast<const f32x4*>(tile + row * LDO + 4 * g);
    const float x = tile[row * LDO + (g ? 8 : 9)];
    const f32x4 b = *reinterpret_cast<const f32x4*>(sB2 + 4 * g);
    const float bx = sB2[g ? 8 : 9];
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = __builtin_fmaf(q[j], so_inv, b[j]);
    w[4] = __builtin_fmaf(x, so_inv, bx);
}

// The same draw with 16 lanes per env (the split forms, where the 32 envs of a workgroup would otherwise be drawn by half
// of ONE wave while seven wait): lane i of a 16-lane row holds output i of its env (logits 0..A-1, the value at A).
// Row-wide max / sum by DPP rotations, the CDF by a DPP scan, the action = number of bins the uniform has passed.  The
// sums are tree-ordered, so the last bits differ from policy_tail's (the split forms differ from the whole-tile forms
// in summation order anyway); the distribution is the same.  act / lp / val are returned in every lane of the row.
#define PC_ROW_ROR(v, n) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + (n), 0xf, 0xf, false))
#define PC_ROW_SHR0(v, n) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x110 + (n), 0xf, 0xf, true))
__device__ __forceinline__ void policy_tail_row(const float v, const int i, const int A, const float u, const int lane, int& act,
                                                float& lp, float& val) {
    const float l = i < A ? v : -INFINITY;
    float mx = l;
    mx = fmaxf(mx, PC_ROW_ROR(mx, 8));
    mx = fmaxf(mx, PC_ROW_ROR(mx, 4));
    mx = fmaxf(mx, PC_ROW_ROR(mx, 2));
    mx = fmaxf(mx, PC_ROW_ROR(mx, 1));
    const float ex = i < A ? softmax_exp(l - mx) : 0.0f;
    float sum = ex;
    sum += PC_ROW_ROR(sum, 8);
    sum += PC_ROW_ROR(sum, 4);
    sum += PC_ROW_ROR(sum, 2);
    sum += PC_ROW_ROR(sum, 1);
    const float lse = mx + softmax_log(sum);  // Categorical(logits=...) normalises: logits - logsumexp
    float cdf = ex * softmax_rcp(sum);    // inclusive scan over the row (lanes shifted in from outside the row read 0)
    cdf += PC_ROW_SHR0(cdf, 1);
    cdf += PC_ROW_SHR0(cdf, 2);
    cdf += PC_ROW_SHR0(cdf, 4);
    cdf += PC_ROW_SHR0(cdf, 8);
    const unsigned long long passed = __ballot(i < A && !(u < cdf));       // inverse CDF: bins the uniform has passed
    const int cnt = __popc((unsigned)(passed >> (lane & 48)) & 0xffffu);
    act = cnt < A - 1 ? cnt : A - 1;                                       // last bin absorbs rounding
    const int row0 = lane & 48;
    lp = __shfl(l, row0 + act, 64) - lse;
    val = __shfl(v, row0 + A, 64);
}
#undef PC_ROW_ROR
#undef PC_ROW_SHR0

template <int NDW> __device__ __forceinline__ void policy_stage_image(const float* __restrict__ image, float* lds, const int tid) {
    // 16-byte coalesced copies, all loads of a thread in flight together
    const f32x4* __restrict__ src = reinterpret_cast<const f32x4*>(image);
    f32x4* dst = reinterpret_cast<f32x4*>(lds);
    constexpr int n4 = NDW / 4;
    constexpr int per = (n4 + 511) / 512;
    f32x4 tmp[per];
#pragma unroll
    for (int j = 0; j < per; ++j) {
        const int i = tid + j * 512;
        if (i < n4) tmp[j] = src[i];
    }
#pragma unroll
    for (int j = 0; j < per; ++j) {
        const int i = tid + j * 512;
        if (i < n4) dst[i] = tmp[j];
    }
}

// 512 threads = 8 waves (2 per SIMD: while one waits on LDS or its ReLU the other feeds the matrix pipe).
// SPLIT = false (large batches): a wave owns 32 envs (2 column tiles of 16) and walks all 32 hidden tiles;
//                a workgroup covers 256 envs per pass.
// SPLIT = true  (small batches): the 8 waves of a workgroup share the SAME 32 envs and take 4 hidden tiles
//                each; their partial [16 x 32] outputs are summed through LDS.  A pass is 8x shorter, so a
//                batch that cannot fill the chip with 256-env workgroups (n_envs < ~32 k) finishes in a
//                fraction of the single-pass latency of the other form.
// PREC = 0: fp32-input MFMA (bit-for-bit an fp32 fmaf chain).  PREC = 1: bf16x3 split on the bf16 matrix cores.
template <int KS, bool SPLIT, int PREC>
__global__ __launch_bounds__(512) void policy_kernel(const float* __restrict__ obs, const int64_t N, const int D, const int A,
                                                     const float* __restrict__ image, const uint64_t seed, const uint64_t offset,
                                                     const uint64_t* __restrict__ offset_dev, int64_t* __restrict__ action,
                                                     float* __restrict__ action_f, float* __restrict__ logprob,
                                                     float* __restrict__ value, float* __restrict__ logits_out) {
    constexpr int HID = 256, NT = 2 * HID / 16;  // 32 hidden tiles: 16 actor + 16 critic
    constexpr int LD1 = pol_ld1(KS), LDO = SPLIT ? 17 : 20, ET = 2;     // (non-split: 16-byte rows for the pair draw's reads)
    constexpr int ENVS_PER_WG = SPLIT ? 32 : 256;
    constexpr int NG = pol_ng(KS), KB = pol_kb(KS);
    constexpr int IMG = PREC ? polx_image_dwords(PREC, NG) : pol_image_padded(KS);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* sW1 = lds;                        // PREC 0: [512][LD1]
    float* sB1 = PREC ? lds + polx_w1_dwords(PREC, NG) + polx_w2_dwords(PREC) : sW1 + 2 * HID * LD1;  // [512]
    float* sW2 = sB1 + 2 * HID;              // PREC 0: [NT][4][64]
    float* sB2 = PREC ? sB1 + 512 : sW2 + NT * 4 * 64;  // [16]
    const unsigned* sW1p = reinterpret_cast<const unsigned*>(lds);                       // PREC 1 operand records
    const unsigned* sW2p = sW1p + polx_w1_dwords(PREC ? PREC : 1, NG);
    const float* sW2c = sB2 + 16;            // PREC 1: critic output weights [256]
    float* sOut = lds + IMG;                 // [8 waves][32 envs][LDO]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lc = lane & 15, lk = lane >> 4;
    policy_stage_image<IMG>(image, lds, tid);
    __syncthreads();

    const uint64_t off = offset + (offset_dev ? *offset_dev : 0);
    float* myOut = sOut + wave * 32 * LDO;
    const int ht0 = SPLIT ? wave * (NT / 8) : 0, ht1 = SPLIT ? ht0 + NT / 8 : NT;
    const int64_t n_chunks = (N + ENVS_PER_WG - 1) / ENVS_PER_WG;
    for (int64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const int64_t env0 = chunk * ENVS_PER_WG + (SPLIT ? 0 : wave * 32);
        f32x4 out[ET];
#pragma unroll
        for (int et = 0; et < ET; ++et) out[et] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        if constexpr (PREC == 0) {
            // ---- B operands of layer 1: X^T, lane (k = lk, j = lc) of env tile et, k-step ks
            float x[ET][KS];
#pragma unroll
            for (int et = 0; et < ET; ++et) {
                const int64_t e = env0 + 16 * et + lc;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int f = 4 * ks + lk;
                    x[et][ks] = (e < N && f < D) ? obs[e * D + f] : 0.0f;
                }
            }
            policy_pass<KS>(sW1, sB1, sW2, ht0, ht1, x, out, lc, lk, lane);
        } else {
            Pieces<PREC> x[ET][KB];
#pragma unroll
            for (int et = 0; et < ET; ++et) {
                const int64_t e = env0 + 16 * et + lc;
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int f = 8 * (4 * kb + lk) + j;
                        v[j] = (e < N && f < D) ? obs[e * D + f] : 0.0f;
                        if constexpr (PREC == 2) v[j] = clamp_h(v[j] * PolScale<PREC>::sx);   // the observations' scaled domain
                    }
                    x[et][kb] = split8<PREC>(v);
                }
            }
            float val[ET] = {0.0f, 0.0f};
            policy_pass16<PREC, KB>(sW1p, sW2p, sB1, sW2c, SPLIT ? wave : 0, SPLIT ? -1 : NT / 2, x, out, val, lc, lk);
#pragma unroll
            for (int et = 0; et < ET; ++et) {  // the env column's value: sum of the 4 lane groups' partials -> output row A
                float t = val[et];
                t += __shfl_xor(t, 16, 64);
                t += __shfl_xor(t, 32, 64);
                if (A >> 2 == lk) out[et][A & 3] += t;
            }
        }
        // ---- out tile -> LDS so that lane = env
        {
            __syncthreads();  // previous pass's readers are done with sOut
#pragma unroll
            for (int et = 0; et < ET; ++et) {
                if constexpr (LDO % 4 == 0) *reinterpret_cast<f32x4*>(myOut + (16 * et + lc) * LDO + 4 * lk) = out[et];
                else {
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) myOut[(16 * et + lc) * LDO + 4 * lk + reg] = out[et][reg];
                }
            }
        }
        __syncthreads();
        if constexpr (SPLIT) {
            // every wave draws for 4 of the 32 envs, 16 lanes (= outputs) per env: sum the 8 waves' partial tiles in a fixed
            // order, then the row-parallel draw
            const int el = wave * 4 + lk, oi = lc;
            const int64_t e = env0 + el;
            float ps = 0.0f;
#pragma unroll
            for (int w = 0; w < 8; ++w) ps += sOut[(w * 32 + el) * LDO + oi];
            const float t = __builtin_fmaf(ps, PolScale<PREC>::so_inv, sB2[oi]);   // outputs back from their scaled domain
            int act;
            float lp, val;
            policy_tail_row(t, oi, A, philox_uniform(seed, off, (uint64_t)e), lane, act, lp, val);
            if (e < N) {
                if (logits_out && oi < A) logits_out[e * A + oi] = t;
                if (oi == 0) {
                    action[e] = act;
                    if (action_f) action_f[e] = (float)act;
                    logprob[e] = lp;
                    value[e] = val;
                }
            }
        } else if (A == 9) {     // (uniform) Discrete(9): the two lanes of an env draw together (policy_tail_pair)
            const int row = lane >> 1, g = lane & 1;
            const int64_t e = env0 + row;
            if (e < N) {
                float w[5];
                pair_outputs<LDO>(myOut, row, g, PolScale<PREC>::so_inv, sB2, w);
                int act;
                float lp, val;
                policy_tail_pair(w, g, philox_uniform(seed, off, (uint64_t)e), act, lp, val);
                if (logits_out) {
#pragma unroll
                    for (int j = 0; j < 5; ++j)
                        if (j < 4 || g) logits_out[e * A + 4 * g + j] = w[j];
                }
                if (g == 0) {
                    action[e] = act;
                    if (action_f) action_f[e] = (float)act;
                    logprob[e] = lp;
                    value[e] = val;
                }
            }
        } else {
            const int64_t e = env0 + lane;
            if (lane < 32 && e < N) {
                float v[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(myOut[lane * LDO + i], PolScale<PREC>::so_inv, sB2[i]);   // outputs back from their scaled domain
                int act;
                float lp, val;
                policy_tail(v, A, philox_uniform(seed, off, (uint64_t)e), act, lp, val, logits_out ? logits_out + e * A : nullptr);
                action[e] = act;
                if (action_f) action_f[e] = (float)act;
                logprob[e] = lp;
                value[e] = val;
            }
        }
    }
}
