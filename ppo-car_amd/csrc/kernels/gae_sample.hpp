// gae_sample.hpp -- part of the single translation unit ppocar.hip (included there, in order; not a stand-alone header).
// K3 gae_kernel (Buffer.calculate_advantages) and K4 sample_kernel (Categorical sample / log_prob / entropy, Philox).
#pragma once

// ------------------------------------------------------------------------------------------
// K3: GAE(lambda), buffer.py:36-64.  One lane per env, serial in t (the recurrence), rows
// coalesced across envs.  Operation order = torch's, one float32 rounding per op (no FMA):
//   delta    = (rew[t] + (gamma * next_val) * term_mask) - val[t]                       :60
//   last_gae = delta + (((gamma*lambda) * term_mask) * trunc_mask) * last_gae           :61
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gae_kernel(const float* __restrict__ rew, const float* __restrict__ val,
                                                  const float* __restrict__ term, const float* __restrict__ trunc,
                                                  const float* __restrict__ last_val, const float* __restrict__ last_term,
                                                  const float* __restrict__ last_trunc, const float g, const float gl,
                                                  const int64_t T, const int64_t N, float* __restrict__ adv,
                                                  float* __restrict__ ret) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    float next_val = last_val[e];             // :53
    float tmask = 1.0f - last_term[e];        // :54
    float trmask = 1.0f - last_trunc[e];      // :55
    float last_gae = 0.0f;
    constexpr int U = 8;  // rows in flight per lane: the loads do not depend on the recurrence
    int64_t t = T - 1;
    for (; t >= U - 1; t -= U) {
        float r[U], v[U], tm[U], tr[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int64_t off = (t - j) * N + e;
            r[j] = rew[off];
            v[j] = val[off];
            tm[j] = term[off];
            tr[j] = trunc[off];
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int64_t off = (t - j) * N + e;
            float tmp = g * next_val;
            tmp = tmp * tmask;
            float delta = r[j] + tmp;
            delta = delta - v[j];
            float c = gl * tmask;
            c = c * trmask;
            c = c * last_gae;
            last_gae = delta + c;
            adv[off] = last_gae;           // :62
            ret[off] = last_gae + v[j];    // :63
            next_val = v[j];
            tmask = 1.0f - tm[j];
            trmask = 1.0f - tr[j];
        }
    }
    for (; t >= 0; --t) {
        const int64_t off = t * N + e;
        const float r = rew[off], v = val[off];
        float tmp = g * next_val;
        tmp = tmp * tmask;
        float delta = r + tmp;
        delta = delta - v;
        float c = gl * tmask;
        c = c * trmask;
        c = c * last_gae;
        last_gae = delta + c;
        adv[off] = last_gae;
        ret[off] = last_gae + v;
        next_val = v;
        tmask = 1.0f - term[off];
        trmask = 1.0f - trunc[off];
    }
}

// ------------------------------------------------------------------------------------------
// K4: categorical sample / log_prob / entropy (model.py:35-40), Philox-4x32-10 counter RNG
// ------------------------------------------------------------------------------------------
// exp(x) for the rollout-time softmax, x = logit - max <= 0: v_exp_f32(x * log2(e)), two instructions.  expf() spends eight
// more per call on carrying x * log2(e) in extended precision; here the product's rounding is a relative error of
// |x| * 2^-24 in the result (1e-7 at x = -2, 1e-6 at x = -20 where the probability is 2e-9), against the 1e-5 the log-probs
// are held to and the 2e-6 they are tested at.  Nine calls per env and step: the draw was 3 % of the rollout kernel.
__device__ __forceinline__ float softmax_exp(const float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
// log and reciprocal of the softmax denominator (1 <= sum <= A): the hardware's one-instruction forms.  v_log_f32 (log2, one ulp) times
// ln 2 is within 2 ulps of log -- 2e-7 absolute on a log-sum-exp below 2.2 -- and v_rcp_f32 within one ulp of 1 / sum; the IEEE division
// and the full-range logf they replace cost 16 more vector instructions per draw for bits the sampler's contract (log-prob within 2e-6 of
// Categorical, model.py:34-41) does not ask for.  EVERY draw of the library goes through these two (pc_sample, the fused policy step in
// all its forms, the persistent kernels): their buffers stay bit-identical to each other.
__device__ __forceinline__ float softmax_log(const float sum) { return __builtin_amdgcn_logf(sum) * 0.693147180559945309417f; }
__device__ __forceinline__ float softmax_rcp(const float sum) { return __builtin_amdgcn_rcpf(sum); }

__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (uint32_t)p1;
    c[3] = (uint32_t)p0;
    c[0] = n0;
    c[2] = n2;
}

// The stream: draw number `offset` of element `idx` is word (offset & 3) of the Philox block with counter
// (idx, offset >> 2) and key `seed` -- all four words of a block are used, so a kernel that walks consecutive
// offsets (the persistent rollout) runs the ten rounds once per four draws.
struct PhiloxBlock { uint32_t w[4]; };
__device__ __forceinline__ PhiloxBlock philox_block(uint64_t seed, uint64_t block, uint64_t idx) {
    uint32_t c[4] = {(uint32_t)idx, (uint32_t)(idx >> 32), (uint32_t)block, (uint32_t)(block >> 32)};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return PhiloxBlock{{c[0], c[1], c[2], c[3]}};
}
__device__ __forceinline__ float philox_word_uniform(const PhiloxBlock& b, const unsigned word) {  // word: wave-uniform
    const uint32_t x = word == 0 ? b.w[0] : word == 1 ? b.w[1] : word == 2 ? b.w[2] : b.w[3];
    return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f);  // (0, 1) open, 24 bits
}
__device__ __forceinline__ float philox_uniform(uint64_t seed, uint64_t offset, uint64_t idx) {
    return philox_word_uniform(philox_block(seed, offset >> 2, idx), (unsigned)(offset & 3));
}

template <int AMAX>
__global__ __launch_bounds__(256) void sample_kernel(const float* __restrict__ logits, const int64_t N, const int A,
                                                     const uint64_t seed, const uint64_t offset,
                                                     int64_t* __restrict__ actions, float* __restrict__ logprob,
                                                     float* __restrict__ entropy) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    float l[AMAX];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < AMAX; ++i) {
        l[i] = i < A ? logits[e * A + i] : -INFINITY;
        mx = fmaxf(mx, l[i]);
    }
    float ex[AMAX];
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < AMAX; ++i) {
        if (i >= A) break;
        ex[i] = softmax_exp(l[i] - mx);
        sum += ex[i];
    }
    const float lse = mx + softmax_log(sum);  // Categorical(logits=...) normalises: logits - logsumexp
    const float inv = softmax_rcp(sum);
    const float u = philox_uniform(seed, offset, (uint64_t)e);
    float cum = 0.0f, ent = 0.0f, lp = 0.0f;
    int act = -1;
#pragma unroll
    for (int i = 0; i < AMAX; ++i) {
        if (i >= A) break;
        const float nl = l[i] - lse;
        const float pr = ex[i] * inv;                        // same draw as policy_tail (the fused policy step)
        cum += pr;
        ent -= pr * fmaxf(nl, -3.4028234663852886e38f);  // torch clamps log-probs at finfo.min
        if (act < 0 && (u < cum || i == A - 1)) {        // inverse CDF; last bin absorbs rounding
            act = i;
            lp = nl;
        }
    }
    actions[e] = act;
    logprob[e] = lp;
    if (entropy) entropy[e] = ent;
}
