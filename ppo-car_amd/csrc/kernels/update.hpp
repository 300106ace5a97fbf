// update.hpp -- part of the single translation unit ppocar.hip (included there, in order; not a stand-alone header).
// K6-K8 and K10-K12: the PPO minibatch step (gather, loss, forward / backward of both MLPs, gradient reduction, clip + Adam).
#pragma once

// ------------------------------------------------------------------------------------------
// K6-K8: the PPO minibatch step's non-GEMM work (train.py:230-261), three launches instead of ~140
// ------------------------------------------------------------------------------------------
// K6: gather one minibatch -- traj_obs[batch_indices] etc. (train.py:233-238,249)
__global__ __launch_bounds__(256) void ppo_gather_kernel(const int64_t* __restrict__ idx, const int B, const int D,
                                                         const float* __restrict__ obs, const float* __restrict__ act,
                                                         const float* __restrict__ logprob, const float* __restrict__ adv,
                                                         const float* __restrict__ ret, float* __restrict__ o_obs,
                                                         float* __restrict__ o_act, float* __restrict__ o_logprob,
                                                         float* __restrict__ o_adv, float* __restrict__ o_ret) {
    const int W = D + 4;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * W) return;
    const int b = i / W, c = i - b * W;
    const int64_t src = idx[b];
    if (c < D) o_obs[b * D + c] = obs[src * D + c];
    else if (c == D) o_act[b] = act[src];
    else if (c == D + 1) o_logprob[b] = logprob[src];
    else if (c == D + 2) o_adv[b] = adv[src];
    else o_ret[b] = ret[src];
}

__device__ __forceinline__ float block_sum(float v, float* sh) {  // all threads get the sum; blockDim <= 1024
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    float t = 0.0f;
    for (int i = 0; i < nw; ++i) t += sh[i];
    return t;
}

// K7: clipped-PPO loss of one minibatch, forward AND backward w.r.t. the network outputs (train.py:235-255):
//   ratio = exp(new_lp - old_lp); A = (adv - mean) / max(std_unbiased, 1e-5)
//   L_pi = mean(max(-A r, -A clamp(r, 1-c, 1+c))); L_v = 0.5 mean((v - ret)^2); H = mean(entropy)
//   loss = L_pi + vf L_v - ec H
// One workgroup, one sample per thread (B <= 1024).  Gradients as autograd produces them:
//   dloss/dv_i      = vf (v_i - ret_i) / B
//   dloss/dlp_i     = (1/B) r_i * (-A_i if -A_i r_i >= -A_i clamp(r_i) else 0)     [torch.max / clamp backward]
//   dloss/dlogit_ik = dloss/dlp_i (1[k = a_i] - p_ik) + (ec/B) p_ik (log p_ik + H_i)
// metrics[0..3] += (L_pi, L_v, H, loss)  (train.py:263-266).
template <int AMAX>
__global__ __launch_bounds__(1024) void ppo_loss_kernel(const float* __restrict__ logits, const float* __restrict__ values,
                                                        const float* __restrict__ act, const float* __restrict__ old_lp,
                                                        const float* __restrict__ adv, const float* __restrict__ ret, const int B,
                                                        const int A, const float clip, const float vf, const float ec,
                                                        float* __restrict__ dlogits, float* __restrict__ dvalues,
                                                        float* __restrict__ metrics) {
    __shared__ float sh[16];
    const int i = threadIdx.x;
    const bool on = i < B;
    const float invB = 1.0f / (float)B;
    const float a_raw = on ? adv[i] : 0.0f;
    const float mean = block_sum(a_raw, sh) * invB;
    const float dev = on ? a_raw - mean : 0.0f;
    const float var = block_sum(dev * dev, sh) / (float)(B - 1);   // unbiased, as Tensor.std() (train.py:239)
    const float sd = fmaxf(sqrtf(var), 1e-5f);                     // torch.max(std, 1e-5) (train.py:239-240)
    float pl = 0.0f, vl = 0.0f, ent = 0.0f;
    if (on) {
        float l[AMAX];
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < AMAX; ++k) {
            l[k] = k < A ? logits[i * A + k] : -INFINITY;
            mx = fmaxf(mx, l[k]);
        }
        float sum = 0.0f;
#pragma unroll
        for (int k = 0; k < AMAX; ++k) sum += k < A ? expf(l[k] - mx) : 0.0f;
        const float lse = mx + logf(sum);
        const int a = (int)act[i];
        float new_lp = 0.0f;
        float pk[AMAX], lpk[AMAX];
#pragma unroll
        for (int k = 0; k < AMAX; ++k) {
            lpk[k] = k < A ? l[k] - lse : 0.0f;
            pk[k] = k < A ? expf(lpk[k]) : 0.0f;
            ent -= pk[k] * lpk[k];
            if (k == a) new_lp = lpk[k];
        }
        const float r = expf(new_lp - old_lp[i]);                                  // :235
        const float An = dev / sd;                                                 // :238-240
        const float rc = fminf(fmaxf(r, 1.0f - clip), 1.0f + clip);
        const float pl1 = -An * r, pl2 = -An * rc;                                 // :243-244
        pl = fmaxf(pl1, pl2);                                                      // :245
        const float dv = values[i] - ret[i];
        vl = 0.5f * dv * dv;                                                       // :249
        const float g_lp = (pl1 >= pl2 ? -An : 0.0f) * r * invB;
        dvalues[i] = vf * dv * invB;
#pragma unroll
        for (int k = 0; k < AMAX; ++k)
            if (k < A) dlogits[i * A + k] = g_lp * ((k == a ? 1.0f : 0.0f) - pk[k]) + ec * invB * pk[k] * (lpk[k] + ent);
    }
    const float s_pl = block_sum(pl, sh) * invB, s_vl = block_sum(vl, sh) * invB, s_en = block_sum(ent, sh) * invB;
    if (i == 0) {
        metrics[0] += s_pl;
        metrics[1] += s_vl;
        metrics[2] += s_en;
        metrics[3] += s_pl + vf * s_vl - ec * s_en;                                // :255
    }
}

// K8: nn.utils.clip_grad_norm_(params, max_norm) (train.py:260) + Adam.step() (train.py:261, lr from the device,
// eps 1e-5, betas (0.9, 0.999), no weight decay / amsgrad) over the flat parameter bucket, one workgroup.
// grad_scale folds the 1/world_size of the gradient average in.  state[0] = step count (float), updated here.
__global__ __launch_bounds__(1024) void clip_adam_kernel(float* __restrict__ param, float* __restrict__ grad,
                                                         float* __restrict__ exp_avg, float* __restrict__ exp_avg_sq,
                                                         float* __restrict__ step_count, const float* __restrict__ lr_dev,
                                                         const int n, const float max_norm, const float grad_scale,
                                                         const float beta1, const float beta2, const float eps) {
    __shared__ float sh[16];
    float ss = 0.0f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float g = grad[i] * grad_scale;
        ss += g * g;
    }
    const float total_norm = sqrtf(block_sum(ss, sh));
    const float coef = fminf(max_norm / (total_norm + 1e-6f), 1.0f);   // clip_coef_clamped
    const float step = step_count[0] + 1.0f;
    const float bc1 = 1.0f - powf(beta1, step), bc2 = 1.0f - powf(beta2, step);
    const float step_size = lr_dev[0] / bc1;
    const float bc2_sqrt = sqrtf(bc2);
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float g = grad[i] * grad_scale * coef;
        grad[i] = g;                                                   // clip_grad_norm_ scales the grads in place
        const float m = exp_avg[i] + (1.0f - beta1) * (g - exp_avg[i]);            // exp_avg.lerp_(grad, 1 - beta1)
        const float v = beta2 * exp_avg_sq[i] + (1.0f - beta2) * g * g;            // exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2)
        exp_avg[i] = m;
        exp_avg_sq[i] = v;
        const float denom = sqrtf(v) / bc2_sqrt + eps;
        param[i] -= step_size * (m / denom);                                        // param.addcdiv_(exp_avg, denom, -step_size)
    }
    __syncthreads();
    if (threadIdx.x == 0) step_count[0] = step;
}

// ------------------------------------------------------------------------------------------
// K10-K12: one PPO minibatch step (train.py:230-261) without any library GEMM: the two MLPs are 14.9 k
// parameters and a minibatch is 44 MFLOP -- twelve library GEMM launches of 5-21 us each were the cost.
//   K10 ppo_fwdbwd_kernel : gather + forward + loss + backward for 8 samples per workgroup; thread u owns
//                           hidden unit u of BOTH nets (its W1 rows, W2 column and their gradient accumulators
//                           live in registers); per-workgroup gradient partials, no atomics (deterministic)
//   K11 grad_reduce_kernel: sums the partials into the flat gradient, per-block squared-norm partials, metrics
//   K12 adam_kernel       : clip_grad_norm_ + Adam over the flat bucket, one element per thread
// Parameter order = torch's module.parameters(): aW1 [H][D], ab1 [H], aW2 [A][H], ab2 [A], cW1, cb1, cW2 [1][H], cb2.
// ------------------------------------------------------------------------------------------
constexpr int FB_S = 8;  // samples per workgroup

// clip_grad_norm_ + Adam for ONE element, shared by K12 (adam_kernel), K12m (clip_adam_mb_kernel) and K10's deferred form: the
// same instruction sequence wherever a parameter is updated, so every path gives the same bits.  torch's update is
//   g = grad * clip_coef;  m = m + (1 - b1)(g - m);  v = b2 v + (1 - b2) g g;  p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
// (train.py:260-261, Adam(eps = 1e-5) train.py:146), with torch's own operations: a correctly rounded square root and two
// correctly rounded divisions (sqrtf, `/`: ~35 instructions per element, one element per thread in K12 -- nothing against the
// launch; denormal second moments are honoured).  Round 4 had v_sqrt_f32 / v_rcp_f32 here for the sake of K10's deferred form
// (off by default, measured slower): an approximation in the shipped update for a disabled feature -- gone.
struct AdamCoef { float coef, step_size, bc2_sqrt, beta1, beta2, eps; };
__device__ __forceinline__ AdamCoef adam_coef(const float norm_sq, const float step, const float lr, const float max_norm, const float beta1,
                                              const float beta2, const float eps) {
    AdamCoef c;
    c.coef = fminf(max_norm / (sqrtf(norm_sq) + 1e-6f), 1.0f);                 // clip_coef_clamped
    const float bc1 = 1.0f - powf(beta1, step), bc2 = 1.0f - powf(beta2, step);
    c.step_size = lr / bc1;
    c.bc2_sqrt = sqrtf(bc2);
    c.beta1 = beta1;
    c.beta2 = beta2;
    c.eps = eps;
    return c;
}
__device__ __forceinline__ void adam_elem(const AdamCoef& c, const float g_raw, float& p, float& m, float& v) {
    const float g = g_raw * c.coef;
    m = m + (1.0f - c.beta1) * (g - m);                                        // exp_avg.lerp_(grad, 1 - beta1)
    v = c.beta2 * v + (1.0f - c.beta2) * g * g;                                // exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2)
    const float denom = sqrtf(v) / c.bc2_sqrt + c.eps;                         // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
    p = p - c.step_size * (m / denom);                                         // param.addcdiv_(exp_avg, denom, -step_size)
}
// the squared gradient norm from K11's per-block partials, summed in index order (every thread, every workgroup: the same float)
__device__ __forceinline__ float norm_sq_from_partials(const float* __restrict__ norm_partial, const int n_norm) {
    float ss = 0.0f;
    for (int j = 0; j < n_norm; ++j) ss += norm_partial[j];
    return ss;
}
// K10's deferred form: the clip + Adam step of the PREVIOUS minibatch (K12) is taken by the next forward / backward launch as it
// loads the parameters -- every workgroup needs all of them anyway, and every thread exactly the 2 D + 12 + ... elements of its own
// hidden unit: it reads (param, grad, exp_avg, exp_avg_sq) of generation i, updates them in registers (adam_elem) and runs on the
// result; workgroup 0 also writes generation i + 1 into the OTHER of two state buffers (readers and the writer of one launch never
// meet).  Per minibatch the chain is two launches (K10, K11) instead of three; the last gradient of an epoch is applied by one
// K12 launch that also brings the state home to the caller's tensors.
struct AdamDefer {
    const float* __restrict__ grad;          // the previous minibatch's raw gradient (K11's output); nullptr = nothing to apply
    const float* __restrict__ m_in;
    const float* __restrict__ v_in;
    float* __restrict__ p_out;
    float* __restrict__ m_out;
    float* __restrict__ v_out;
    const float* __restrict__ norm_partial;
    int n_norm;
    const float* __restrict__ step_count;
    const float* __restrict__ lr_dev;
    float max_norm, beta1, beta2, eps;
};

// 64-lane sum with DPP row operations (VALU only; the __shfl_xor butterfly goes through the LDS crossbar
// with ~100 cycles of dependent latency per step).  The total lands in lane 63; readlane broadcasts it.
__device__ __forceinline__ float wave_sum(float v) {
#define PC_DPP(ctrl, rmask) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xf, false))
    PC_DPP(0x111, 0xf);  // row_shr:1
    PC_DPP(0x112, 0xf);  // row_shr:2
    PC_DPP(0x114, 0xf);  // row_shr:4
    PC_DPP(0x118, 0xf);  // row_shr:8   -> lane 15 of each row holds the row sum
    PC_DPP(0x142, 0xa);  // row_bcast:15 -> rows 1 and 3 add the previous row's total
    PC_DPP(0x143, 0xc);  // row_bcast:31 -> rows 2 and 3 add lane 31's total
#undef PC_DPP
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// (mean, max(unbiased std, 1e-5)) of a minibatch's advantages (train.py:238-240) over a 256-thread workgroup, thread u
// holding elements u, u + 256, ...  One code path for the minibatch kernel and the prepare kernel: same bits.
__device__ __forceinline__ void adv_stats(const float (&a_loc)[4], const float a_sum, const int B, float* sh, float& mean, float& sd) {
    const int u = threadIdx.x;
    mean = block_sum(a_sum, sh) * (1.0f / (float)B);
    float d2 = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = u + j * 256;
        const float dv = i < B ? a_loc[j] - mean : 0.0f;
        d2 += dv * dv;
    }
    sd = fmaxf(sqrtf(block_sum(d2, sh) / (float)(B - 1)), 1e-5f);
}

// K10p: gather n_mb minibatches in one launch (one workgroup each): sample rows, per-sample scalars, advantage statistics.
// What every workgroup of K10 otherwise does for itself at the head of its critical path -- an index load, then the
// dependent row loads (two cold misses in a row), then two workgroup reductions -- is done here once per epoch.
__global__ __launch_bounds__(256) void ppo_prepare_kernel(const int64_t* __restrict__ idx, const int64_t idx_ld, const int B, const int D,
                                                          const float* __restrict__ obs, const float* __restrict__ act,
                                                          const float* __restrict__ old_lp, const float* __restrict__ adv,
                                                          const float* __restrict__ ret, float* __restrict__ prepared,
                                                          const int64_t prep_ld) {
    __shared__ float sh[16];
    const int u = threadIdx.x;
    const int64_t* ix = idx + (int64_t)blockIdx.x * idx_ld;
    float* out = prepared + (int64_t)blockIdx.x * prep_ld;
    float* ps = out + (size_t)B * D;
    float a_loc[4], a_sum = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = u + j * 256;
        const int64_t src = i < B ? ix[i] : 0;
        a_loc[j] = i < B ? adv[src] : 0.0f;
        a_sum += a_loc[j];
        if (i < B) {
            ps[i] = act[src];
            ps[B + i] = old_lp[src];
            ps[2 * B + i] = a_loc[j];
            ps[3 * B + i] = ret[src];
        }
    }
    for (int i = u; i < B * D; i += 256) {
        const int b = i / D, f = i - b * D;
        out[i] = obs[ix[b] * D + f];
    }
    float mean, sd;
    adv_stats(a_loc, a_sum, B, sh, mean, sd);
    if (u == 0) {
        ps[4 * B] = mean;
        ps[4 * B + 1] = sd;
        ps[4 * B + 2] = 0.0f;
        ps[4 * B + 3] = 0.0f;
    }
}

// AC > 0: the action count as a compile-time constant (CarEnv: Discrete(9)) -- with the run-time count every `o < A` inside the
// unrolled 16-slot loops is a v_cndmask per slot and sample and the FMAs of the unused slots are executed (backward: 219
// selects + 112 FMAs per thread); the same operations in the same order either way.
// DC > 0: likewise the observation width (6 + 12 / 17 / 33 rays).
template <int DMAX, int AC = 0, int DC = 0, bool DEFER = false>
__device__ __forceinline__ void ppo_fwdbwd_body(const int wg, const int64_t* __restrict__ idx, const int B, const int D_rt, const int A_rt,
                                                const float* __restrict__ obs, const float* __restrict__ act,
                                                const float* __restrict__ old_lp, const float* __restrict__ adv,
                                                const float* __restrict__ ret, const float* __restrict__ param,
                                                const float clip, const float vf, const float ec,
                                                float* __restrict__ partial, float* __restrict__ metric_partial,
                                                const float* __restrict__ prep, const AdamDefer df) {
    constexpr int H = 256, S = FB_S, LDT = DMAX + 1, LDH = H + 1;
    const int A = AC > 0 ? AC : A_rt, D = DC > 0 ? DC : D_rt;
    static_assert(DC <= DMAX, "observation width");
    // Everything in this kernel is latency: a minibatch is 44 MFLOP.  So: every global access coalesced (the [H][D]
    // weight matrices and their gradients go through an LDS tile, transposed there), all loads of a phase in flight
    // together, and no cross-lane reduction chains (layer 2 is a small GEMV out of LDS).
    __shared__ float sh[16];
    __shared__ float sX[S][DMAX];
    __shared__ float sOut[S][16];
    __shared__ float sDout[S][16];
    __shared__ float sMet[S][3];
    __shared__ float sSmp[S][4];                                          // act, old_lp, adv, ret of my samples
    __shared__ __attribute__((aligned(16))) float sT[H * LDT > 2 * S * LDH + 16 * LDH + 4 * S * 16 ? H * LDT : 2 * S * LDH + 16 * LDH + 4 * S * 16];
    static_assert(H * LDT >= H * DMAX + 8, "the tile holds one [H][D] block in natural order plus an alignment shift");
    float* sHid = sT;                    // [2][S][LDH]  hidden activations (actor, critic)         } alias the transposition
    float* sW2 = sT + 2 * S * LDH;       // [16][LDH]    output-layer weights, row A = the critic's  } tile: used between
    float* sP2 = sW2 + 16 * LDH;         // [<= 4][S][16] the k-parts of layer 2                     } the load and store phases
    const int u = threadIdx.x;
    PC_STAMP_U(0)
    __syncthreads();  // a previous pass's readers of the shared arrays are done (persistent epoch kernel)
    // flat parameter offsets
    const int o_aW1 = 0, o_ab1 = H * D, o_aW2 = o_ab1 + H, o_ab2 = o_aW2 + A * H, o_cW1 = o_ab2 + A, o_cb1 = o_cW1 + H * D,
              o_cW2 = o_cb1 + H, o_cb2 = o_cW2 + H, n_param = o_cb2 + 1;

    // ---- loads that do not depend on anything, all issued before the first wait
    // prep != nullptr: this minibatch was gathered by ppo_prepare_kernel -- rows [B][D], act / old_lp / adv / ret [B] and
    // (mean, std) of the advantages, contiguous -- so nothing here depends on an index load and no statistics are reduced
    const int s0 = wg * S;
    int64_t my_src = 0;                                                   // threads 0..S-1: my sample's row
    int64_t a_src[4] = {0, 0, 0, 0};
    if (!prep) {
        if (u < S && s0 + u < B) my_src = idx[s0 + u];
#pragma unroll
        for (int j = 0; j < 4; ++j) a_src[j] = u + j * 256 < B ? idx[u + j * 256] : 0;
    }
    // defer (uniform): `param` holds generation i, the previous minibatch's clip + Adam step has not been taken yet (AdamDefer):
    // every parameter element this thread loads is loaded with its gradient and moments, updated in registers, and -- by
    // workgroup 0 -- written to generation i + 1.  All loads of the phase are issued before the first wait either way.
    // DEFER is compiled in for CarEnv's shapes only (the generic kernels take the three-launch step).
    const bool defer = DEFER && df.grad != nullptr;
    const bool wr = defer && wg == 0;
    float w2a[16], w2a_g[16], w2a_m[16], w2a_v[16];
#pragma unroll
    for (int o = 0; o < 16; ++o) {
        w2a[o] = o < A ? param[o_aW2 + o * H + u] : 0.0f;
        w2a_g[o] = w2a_m[o] = w2a_v[o] = 0.0f;
        if (defer && o < A) {
            w2a_g[o] = df.grad[o_aW2 + o * H + u];
            w2a_m[o] = df.m_in[o_aW2 + o * H + u];
            w2a_v[o] = df.v_in[o_aW2 + o * H + u];
        }
    }
    float w2c = param[o_cW2 + u], b1a = param[o_ab1 + u], b1c = param[o_cb1 + u];
    const int ob = u & 15;                                                // my output index in the layer-2 epilogue
    const int o_b2 = ob < A ? o_ab2 + ob : o_cb2;                         // (ob > A: a dummy slot, value unused)
    float b2 = ob <= A ? param[o_b2] : 0.0f;
    float sc_g[4] = {0.0f, 0.0f, 0.0f, 0.0f}, sc_m[4] = {0.0f, 0.0f, 0.0f, 0.0f}, sc_v[4] = {0.0f, 0.0f, 0.0f, 0.0f};   // w2c, b1a, b1c, b2
    const int sc_i[4] = {o_cW2 + u, o_ab1 + u, o_cb1 + u, o_b2};
    if (defer) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q < 3 || ob <= A) {
                sc_g[q] = df.grad[sc_i[q]];
                sc_m[q] = df.m_in[sc_i[q]];
                sc_v[q] = df.v_in[sc_i[q]];
            }
        }
    }
    // W1 of both nets with 16-byte loads: the [H][D] block at parameter offset `off` is fetched as the aligned float4 window
    // [off & ~3, off + H D) -- NV4 loads per thread and net instead of D dword loads (the kernel's memory instructions were a
    // third of its time: profiles/, K10 phase stamps) -- and goes through the LDS tile in that same natural order.
    constexpr int NV4 = (H * DMAX + 3 + 1023) / 1024 + 1;
    constexpr int NVD = DEFER ? NV4 : 1;
    f32x4 w1raw4[2][NV4], w1g[2][NVD], w1m[2][NVD], w1v[2][NVD];
    int w1_shift[2], w1_n4[2], w1_b4[2];
#pragma unroll
    for (int net = 0; net < 2; ++net) {
        const int off = net == 0 ? o_aW1 : o_cW1, b4 = off & ~3;
        w1_shift[net] = off - b4;
        w1_b4[net] = b4;
        w1_n4[net] = (off + H * D - b4 + 3) >> 2;
        const f32x4* __restrict__ src = reinterpret_cast<const f32x4*>(param + b4);
#pragma unroll
        for (int j = 0; j < NV4; ++j) {
            const int i4 = u + 256 * j;
            const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
            w1raw4[net][j] = i4 < w1_n4[net] ? src[i4] : z;
            if constexpr (DEFER) {
                w1g[net][j] = w1m[net][j] = w1v[net][j] = z;
                if (defer && i4 < w1_n4[net]) {
                    w1g[net][j] = reinterpret_cast<const f32x4*>(df.grad + b4)[i4];
                    w1m[net][j] = reinterpret_cast<const f32x4*>(df.m_in + b4)[i4];
                    w1v[net][j] = reinterpret_cast<const f32x4*>(df.v_in + b4)[i4];
                }
            }
        }
    }
    if constexpr (DEFER) if (defer) {      // (uniform) the deferred clip + Adam step, element by element in registers
        const AdamCoef ac = adam_coef(norm_sq_from_partials(df.norm_partial, df.n_norm), df.step_count[0], df.lr_dev[0], df.max_norm, df.beta1,
                                      df.beta2, df.eps);
#pragma unroll
        for (int o = 0; o < 16; ++o) {
            if (o < A) {
                adam_elem(ac, w2a_g[o], w2a[o], w2a_m[o], w2a_v[o]);
                if (wr) {
                    df.p_out[o_aW2 + o * H + u] = w2a[o];
                    df.m_out[o_aW2 + o * H + u] = w2a_m[o];
                    df.v_out[o_aW2 + o * H + u] = w2a_v[o];
                }
            }
        }
        float sc_p[4] = {w2c, b1a, b1c, b2};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q < 3 || ob <= A) {
                adam_elem(ac, sc_g[q], sc_p[q], sc_m[q], sc_v[q]);
                if (wr && (q < 3 || u <= A)) {     // (the 16 copies of an output bias: one writer)
                    df.p_out[sc_i[q]] = sc_p[q];
                    df.m_out[sc_i[q]] = sc_m[q];
                    df.v_out[sc_i[q]] = sc_v[q];
                }
            }
        }
        w2c = sc_p[0];
        b1a = sc_p[1];
        b1c = sc_p[2];
        b2 = sc_p[3];
#pragma unroll
        for (int net = 0; net < 2; ++net) {
#pragma unroll
            for (int j = 0; j < NV4; ++j) {
                const int i4 = u + 256 * j;
                if (i4 < w1_n4[net]) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {      // (a vector element does not bind to a reference)
                        float pe = w1raw4[net][j][c], me = w1m[net][j][c], ve = w1v[net][j][c];
                        adam_elem(ac, w1g[net][j][c], pe, me, ve);
                        w1raw4[net][j][c] = pe;
                        w1m[net][j][c] = me;
                        w1v[net][j][c] = ve;
                    }
                    if (wr) {      // (the window's few elements outside the block belong to its neighbours: the same values again)
                        reinterpret_cast<f32x4*>(df.p_out + w1_b4[net])[i4] = w1raw4[net][j];
                        reinterpret_cast<f32x4*>(df.m_out + w1_b4[net])[i4] = w1m[net][j];
                        reinterpret_cast<f32x4*>(df.v_out + w1_b4[net])[i4] = w1v[net][j];
                    }
                }
            }
        }
    }
    if (ob > A) b2 = 0.0f;
    // ---- second-level loads (addresses came from idx)
    float a_loc[4] = {0.0f, 0.0f, 0.0f, 0.0f}, a_sum = 0.0f;
    float pre_mean = 0.0f, pre_sd = 1.0f;
    if (prep) {
        const float* ps = prep + (size_t)B * D;                           // act | old_lp | adv | ret | (mean, std, -, -)
        pre_mean = ps[4 * B];
        pre_sd = ps[4 * B + 1];
        if (u < S) {
            const bool lv = s0 + u < B;
#pragma unroll
            for (int c = 0; c < 4; ++c) sSmp[u][c] = lv ? ps[c * B + s0 + u] : 0.0f;
        }
        for (int i = u; i < S * DMAX; i += 256) {
            const int sidx = i / DMAX, f = i - sidx * DMAX;
            const int b = s0 + sidx;
            sX[sidx][f] = (b < B && f < D) ? prep[(size_t)b * D + f] : 0.0f;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a_loc[j] = u + j * 256 < B ? adv[a_src[j]] : 0.0f;
            a_sum += a_loc[j];
        }
        if (u < S) {
            const bool lv = s0 + u < B;
            sSmp[u][0] = lv ? act[my_src] : 0.0f;
            sSmp[u][1] = lv ? old_lp[my_src] : 0.0f;
            sSmp[u][2] = lv ? adv[my_src] : 0.0f;
            sSmp[u][3] = lv ? ret[my_src] : 0.0f;
        }
        for (int i = u; i < S * DMAX; i += 256) {                         // gather my workgroup's samples (train.py:233-238)
            const int sidx = i / DMAX, f = i - sidx * DMAX;
            const int b = s0 + sidx;
            sX[sidx][f] = (b < B && f < D) ? obs[idx[b] * D + f] : 0.0f;
        }
    }
    PC_STAMP_U(1)
    // ---- W1 rows into registers through the LDS tile (one net at a time: the tile holds [H][D] once, natural order)
    float w1a[DMAX], w1c[DMAX];
#pragma unroll
    for (int net = 0; net < 2; ++net) {
        lds_barrier();  // LDS-only: __syncthreads() would also wait for every outstanding global access
#pragma unroll
        for (int j = 0; j < NV4; ++j) {
            const int i4 = u + 256 * j;
            if (i4 < w1_n4[net]) reinterpret_cast<f32x4*>(sT)[i4] = w1raw4[net][j];
        }
        lds_barrier();  // LDS-only: __syncthreads() would also wait for every outstanding global access
        const float* row = sT + w1_shift[net] + u * D;       // (row stride D floats: conflict-free for odd D)
#pragma unroll
        for (int f = 0; f < DMAX; ++f) {
            const float w = f < D ? row[f] : 0.0f;
            if (net == 0) w1a[f] = w;
            else w1c[f] = w;
        }
    }
    // ---- per-minibatch advantage statistics (train.py:238-240), recomputed identically by every workgroup
    const float invB = 1.0f / (float)B;
    float mean = pre_mean, sd = pre_sd;
    if (!prep) adv_stats(a_loc, a_sum, B, sh, mean, sd);   // (uniform branch)
    __syncthreads();   // (also: every thread has read its W1 row out of the tile, which sHid / sW2 alias)

    PC_STAMP_U(2)
    // ---- forward, layer 1 (Linear + ReLU), both nets
    // (DMAX = 40: the sample loops stay rolled and the activations are re-read from LDS in the backward pass -- fully
    // unrolled, the compiler keeps all S x D sample values live at once and spills a hundred registers)
    constexpr bool ROLLED = DMAX > 24;
    float ha[ROLLED ? 1 : S], hc[ROLLED ? 1 : S];
#pragma unroll
    for (int sidx = 0; sidx < (ROLLED ? 0 : S); ++sidx) {
        float za = b1a, zc = b1c;
#pragma unroll
        for (int f = 0; f < DMAX; ++f) {
            za = __builtin_fmaf(w1a[f], sX[sidx][f], za);
            zc = __builtin_fmaf(w1c[f], sX[sidx][f], zc);
        }
        ha[sidx] = fmaxf(za, 0.0f);
        hc[sidx] = fmaxf(zc, 0.0f);
        sHid[sidx * LDH + u] = ha[sidx];
        sHid[(S + sidx) * LDH + u] = hc[sidx];
    }
    if constexpr (ROLLED) {
#pragma unroll 1
        for (int sidx = 0; sidx < S; ++sidx) {
            float za = b1a, zc = b1c;
#pragma unroll
            for (int f = 0; f < DMAX; ++f) {
                za = __builtin_fmaf(w1a[f], sX[sidx][f], za);
                zc = __builtin_fmaf(w1c[f], sX[sidx][f], zc);
            }
            sHid[sidx * LDH + u] = fmaxf(za, 0.0f);
            sHid[(S + sidx) * LDH + u] = fmaxf(zc, 0.0f);
        }
    }
#pragma unroll
    for (int o = 0; o < 16; ++o) sW2[o * LDH + u] = o < A ? w2a[o] : (o == A ? w2c : 0.0f);
    __syncthreads();
    PC_STAMP_U(3)
    // ---- forward, layer 2: out[s][o] = sum_u W2[o][u] h[s][u]: the S (A + 1) dot products, each cut into as many k-parts as
    // 256 threads allow (3 at A = 9: 86 hidden units per thread instead of the 128 of a fixed split in halves), sequential in u
    // within a part and parts summed in order (deterministic); LDH = 257 keeps the rows a wave touches in distinct banks
    const int n_out = A + 1, n_pair = S * n_out, n_kp = 256 / n_pair < 4 ? 256 / n_pair : 4;
    {
        const int kp = u / n_pair, pr = u - kp * n_pair, sidx = pr / n_out, o = pr - sidx * n_out;
        if (kp < n_kp) {
            const int k0 = H * kp / n_kp, k1 = H * (kp + 1) / n_kp;
            const float* hrow = sHid + ((o < A ? 0 : S) + sidx) * LDH;
            const float* wrow = sW2 + o * LDH;
            float acc = 0.0f;
#pragma unroll 8
            for (int k = k0; k < k1; ++k) acc = __builtin_fmaf(wrow[k], hrow[k], acc);
            sP2[(kp * S + sidx) * 16 + o] = acc;
        }
    }
    __syncthreads();
    if (u < S * 16) {
        const int sidx = u >> 4, o = u & 15;
        if (o <= A) {
            float t = b2;
            for (int kp = 0; kp < n_kp; ++kp) t += sP2[(kp * S + sidx) * 16 + o];
            sOut[sidx][o] = t;
        }
    }
    __syncthreads();
    PC_STAMP_U(4)
    // ---- loss and its gradient w.r.t. the outputs (train.py:235-255; as ppo_loss_kernel), 16 lanes per sample: lane k of a
    // 16-lane row holds output k (logits 0..A-1, the value at A); row-wide max / sums by DPP rotations (tree order), everything
    // after the reductions is computed redundantly by the row's lanes.  (One THREAD per sample walked the ten exponentials, the
    // logarithm and the division as one dependent chain: 6 k cycles, an eighth of the kernel.)
    if (u < S * 16) {
#define PC_ROW_ROR(v, n) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + (n), 0xf, 0xf, false))
        const int sidx = u >> 4, k = u & 15, b = s0 + sidx;
        const bool live = b < B;
        const float o = sOut[sidx][k < 16 ? k : 0];
        const float l = k < A ? o : -INFINITY;
        float mx = l;
        mx = fmaxf(mx, PC_ROW_ROR(mx, 8));
        mx = fmaxf(mx, PC_ROW_ROR(mx, 4));
        mx = fmaxf(mx, PC_ROW_ROR(mx, 2));
        mx = fmaxf(mx, PC_ROW_ROR(mx, 1));
        const float ex = k < A ? expf(l - mx) : 0.0f;
        float sum = ex;
        sum += PC_ROW_ROR(sum, 8);
        sum += PC_ROW_ROR(sum, 4);
        sum += PC_ROW_ROR(sum, 2);
        sum += PC_ROW_ROR(sum, 1);
        const float lse = mx + logf(sum), inv = 1.0f / sum;
        const float lpk = k < A ? l - lse : 0.0f;
        const float pk = ex * inv;                                            // softmax, one expf per action
        float ent = -(pk * lpk);
        ent += PC_ROW_ROR(ent, 8);
        ent += PC_ROW_ROR(ent, 4);
        ent += PC_ROW_ROR(ent, 2);
        ent += PC_ROW_ROR(ent, 1);
        const int a = (int)sSmp[sidx][0];
        const float new_lp = __shfl(lpk, (u & 48) + (a & 15), 64);            // the row's lane a
        const float r = expf(new_lp - sSmp[sidx][1]);                         // :235
        const float An = (sSmp[sidx][2] - mean) / sd;                         // :238-240
        const float rc = fminf(fmaxf(r, 1.0f - clip), 1.0f + clip);
        const float pl1 = -An * r, pl2 = -An * rc;                            // :243-244
        const float pl = fmaxf(pl1, pl2);                                     // :245
        const float dv = __shfl(o, (u & 48) + A, 64) - sSmp[sidx][3];
        const float vl = 0.5f * dv * dv;                                      // :249
        const float g_lp = (pl1 >= pl2 ? -An : 0.0f) * r * invB;
        float dk = 0.0f;
        if (k < A) dk = g_lp * ((k == a ? 1.0f : 0.0f) - pk) + ec * invB * pk * (lpk + ent);
        else if (k == A) dk = vf * dv * invB;
        sDout[sidx][k] = live ? dk : 0.0f;
        if (k == 0) {
            sMet[sidx][0] = live ? pl : 0.0f;
            sMet[sidx][1] = live ? vl : 0.0f;
            sMet[sidx][2] = live ? ent : 0.0f;
        }
#undef PC_ROW_ROR
    }
    __syncthreads();
    PC_STAMP_U(5)
    // ---- backward: every thread for its hidden unit; gradient accumulators in registers
    float g1a[DMAX], g1c[DMAX], g2a[16], g2c = 0.0f, gb1a = 0.0f, gb1c = 0.0f;
#pragma unroll
    for (int f = 0; f < DMAX; ++f) {
        g1a[f] = 0.0f;
        g1c[f] = 0.0f;
    }
#pragma unroll
    for (int o = 0; o < 16; ++o) g2a[o] = 0.0f;
    auto backward_sample = [&](const int sidx, const float h_a, const float h_c) {
        float dha = 0.0f;
#pragma unroll
        for (int o = 0; o < 16; ++o) {
            if (o < A) {
                const float d = sDout[sidx][o];
                dha = __builtin_fmaf(w2a[o], d, dha);
                g2a[o] = __builtin_fmaf(d, h_a, g2a[o]);
            }
        }
        const float dval = sDout[sidx][A];
        g2c = __builtin_fmaf(dval, h_c, g2c);
        dha = h_a > 0.0f ? dha : 0.0f;                           // ReLU backward (threshold at 0)
        const float dhc = h_c > 0.0f ? w2c * dval : 0.0f;
        gb1a += dha;
        gb1c += dhc;
#pragma unroll
        for (int f = 0; f < DMAX; ++f) {
            g1a[f] = __builtin_fmaf(dha, sX[sidx][f], g1a[f]);
            g1c[f] = __builtin_fmaf(dhc, sX[sidx][f], g1c[f]);
        }
    };
    if constexpr (ROLLED) {
#pragma unroll 1
        for (int sidx = 0; sidx < S; ++sidx) backward_sample(sidx, sHid[sidx * LDH + u], sHid[(S + sidx) * LDH + u]);
    } else {
#pragma unroll
        for (int sidx = 0; sidx < S; ++sidx) backward_sample(sidx, ha[sidx], hc[sidx]);
    }
    PC_STAMP_U(6)
    // ---- this workgroup's gradient partial.  Layout of a partial (pc_internal: ppo_partial_index): [aW1 (H D)][cW1 (H D)]
    // [ab1, aW2, ab2][cb1, cW2, cb2], rows of n_pad = n_param rounded up to 4 floats -- both [H][D] blocks 16-byte aligned, so
    // they leave through the LDS tile (natural order) as float4 stores: D / 4 instead of D stores per thread and net.
    const int HD = H * D, n_pad = (n_param + 3) & ~3;
    float* __restrict__ P = partial + (size_t)wg * n_pad;
    float* __restrict__ Pm = P + HD;                  // natural index i of the middle / tail blocks -> Pm[i] (see ppo_partial_index)
    Pm[o_ab1 + u] = gb1a;
    P[o_cb1 + u] = gb1c;
#pragma unroll
    for (int o = 0; o < 16; ++o)
        if (o < A) Pm[o_aW2 + o * H + u] = g2a[o];
    P[o_cW2 + u] = g2c;
    if (u <= A) {  // output-layer biases: sum of dout over my samples
        float t = 0.0f;
#pragma unroll
        for (int sidx = 0; sidx < S; ++sidx) t += sDout[sidx][u];
        if (u < A) Pm[o_ab2 + u] = t;
        else P[o_cb2] = t;
    }
    if (u < 3) {
        float t = 0.0f;
#pragma unroll
        for (int sidx = 0; sidx < S; ++sidx) t += sMet[sidx][u];
        metric_partial[wg * 4 + u] = t;
    }
    PC_STAMP_U(8)
#pragma unroll
    for (int net = 0; net < 2; ++net) {
        lds_barrier();  // (not __syncthreads(): that waits for the stores already in flight, ~2 us each time)
        if (net == 0) { PC_STAMP_U(9) } else { PC_STAMP_U(12) }
#pragma unroll
        for (int f = 0; f < DMAX; ++f)
            if (f < D) sT[u * D + f] = net == 0 ? g1a[f] : g1c[f];
        lds_barrier();  // (not __syncthreads(): that waits for the stores already in flight, ~2 us each time)
        if (net == 0) { PC_STAMP_U(10) } else { PC_STAMP_U(13) }
        f32x4* __restrict__ dst = reinterpret_cast<f32x4*>(P + net * HD);
#pragma unroll
        for (int j = 0; j < NV4; ++j) {
            const int i4 = u + 256 * j;
            if (i4 < (HD >> 2)) dst[i4] = reinterpret_cast<const f32x4*>(sT)[i4];
        }
        if (net == 0) { PC_STAMP_U(11) }
    }
    PC_STAMP_U(7)
}

template <int DMAX, int AC = 0, int DC = 0, bool DEFER = false>
__global__ __launch_bounds__(256) void ppo_fwdbwd_kernel(const int64_t* __restrict__ idx, const int B, const int D, const int A,
                                                         const float* __restrict__ obs, const float* __restrict__ act,
                                                         const float* __restrict__ old_lp, const float* __restrict__ adv,
                                                         const float* __restrict__ ret, const float* __restrict__ param,
                                                         const float clip, const float vf, const float ec,
                                                         float* __restrict__ partial, float* __restrict__ metric_partial,
                                                         const float* __restrict__ prep, const AdamDefer df) {
    ppo_fwdbwd_body<DMAX, AC, DC, DEFER>(blockIdx.x, idx, B, D, A, obs, act, old_lp, adv, ret, param, clip, vf, ec, partial, metric_partial, prep, df);
}

// K11: flat_grad[i] = sum_p partial[p][i] (fixed order: deterministic); block-wise squared-norm partials for the clip;
// block 0 folds the metric partials into the running sums (train.py:263-266) and advances the Adam step counter.
__device__ __forceinline__ void grad_reduce_body(const int blk, const float* __restrict__ partial, const int n_part, const int n,
                                                          const int HD, const int mid_end, const int n_pad,
                                                          float* __restrict__ grad, float* __restrict__ norm_partial,
                                                          const float* __restrict__ metric_partial, const int B, const float vf,
                                                          const float ec, float* __restrict__ metrics, float* __restrict__ step_count) {
    __shared__ float sh[16];
    __syncthreads();  // (shared scratch reuse when called in a loop)
    const int i = blk * blockDim.x + threadIdx.x;
    // workgroup 0 also folds the metric partials: their first 256 words are requested now, under the gradient loads
    const float mp_first = (blk == 0 && (int)threadIdx.x < n_part * 4) ? metric_partial[threadIdx.x] : 0.0f;
    float g = 0.0f;
    if (i < n) {
        // natural flat index i -> index inside a partial (ppo_fwdbwd_body's layout: both [H][D] blocks first, 16-byte aligned)
        const int pm = i < HD ? i : (i < mid_end ? i + HD : (i < mid_end + HD ? i - (mid_end - HD) : i));
        const float* __restrict__ pp = partial + pm;
        int pidx = 0;
        for (; pidx + 64 <= n_part; pidx += 64) {  // 64 independent loads in flight: a minibatch of 512 samples has exactly 64
            float t[64];                              // partials -- one round trip (same summation order as the narrower stages)
#pragma unroll
            for (int j = 0; j < 64; ++j) t[j] = pp[(size_t)(pidx + j) * n_pad];
#pragma unroll
            for (int j = 0; j < 64; ++j) g += t[j];
        }
        for (; pidx + 32 <= n_part; pidx += 32) {  // 32 independent loads in flight (the partials were written by other
            float t[32];                              // workgroups: every load is a cold miss); summed in index order
#pragma unroll
            for (int j = 0; j < 32; ++j) t[j] = pp[(size_t)(pidx + j) * n_pad];
#pragma unroll
            for (int j = 0; j < 32; ++j) g += t[j];
        }
        for (; pidx + 8 <= n_part; pidx += 8) {
            float t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = pp[(size_t)(pidx + j) * n_pad];
#pragma unroll
            for (int j = 0; j < 8; ++j) g += t[j];
        }
        for (; pidx < n_part; ++pidx) g += pp[(size_t)pidx * n_pad];
    }
    if (i < n) grad[i] = g;
    const float ss = block_sum(g * g, sh);
    if (threadIdx.x == 0) norm_partial[blk] = ss;
    if (blk == 0) {
        __shared__ float sMp[256];
        float mt[3] = {0.0f, 0.0f, 0.0f};
        for (int p0 = 0; p0 < n_part; p0 += 64) {  // 64 workgroups' (pl, vl, ent, -) at a time, one coalesced load
            __syncthreads();
            sMp[threadIdx.x] = p0 == 0 ? mp_first : (p0 * 4 + (int)threadIdx.x < n_part * 4 ? metric_partial[p0 * 4 + threadIdx.x] : 0.0f);
            __syncthreads();
            if (threadIdx.x < 3) {   // all 64 reads first, then the sum in index order (rolled, every read is an LDS round trip in
                float mv[64];        // series: 64 x ~70 cycles on the one workgroup the whole launch then waits for)
#pragma unroll
                for (int pidx = 0; pidx < 64; ++pidx) mv[pidx] = sMp[pidx * 4 + threadIdx.x];   // (slots past n_part hold 0)
#pragma unroll
                for (int pidx = 0; pidx < 64; ++pidx) mt[threadIdx.x] += mv[pidx];
            }
        }
        if (threadIdx.x < 3) sh[threadIdx.x] = mt[threadIdx.x] / (float)B;
        __syncthreads();
        if (threadIdx.x == 0) {
            metrics[0] += sh[0];
            metrics[1] += sh[1];
            metrics[2] += sh[2];
            metrics[3] += sh[0] + vf * sh[1] - ec * sh[2];  // :255
            if (step_count) step_count[0] += 1.0f;
        }
    }
}

// HD = H * D (the size of one first-layer weight block), mid_end = the natural offset of the critic's (actor.0.weight, actor.0.bias,
// actor.2.weight, actor.2.bias | critic.0.weight ...), n_pad = the partials' row stride
__global__ __launch_bounds__(256) void grad_reduce_kernel(const float* __restrict__ partial, const int n_part, const int n,
                                                          const int HD, const int mid_end, const int n_pad,
                                                          float* __restrict__ grad, float* __restrict__ norm_partial,
                                                          const float* __restrict__ metric_partial, const int B, const float vf,
                                                          const float ec, float* __restrict__ metrics, float* __restrict__ step_count) {
    grad_reduce_body(blockIdx.x, partial, n_part, n, HD, mid_end, n_pad, grad, norm_partial, metric_partial, B, vf, ec, metrics, step_count);
}

// K12: clip_grad_norm_ + Adam, one element per thread; the squared norm arrives as per-block partials of K11 and
// the step counter has already been advanced there.
__device__ __forceinline__ void adam_body(const int blk, const float* __restrict__ p_in, const float* __restrict__ m_in, const float* __restrict__ v_in,
                                          float* __restrict__ grad, float* __restrict__ p_out, float* __restrict__ m_out, float* __restrict__ v_out,
                                          const float* __restrict__ step_count, const float* __restrict__ lr_dev,
                                          const float* __restrict__ norm_partial, const int n_norm, const int n, const float max_norm,
                                          const float beta1, const float beta2, const float eps) {
    // all loads first (cold misses: the operands were written by other workgroups), the norm partials once per
    // workgroup through LDS; every thread then sums them in index order
    __shared__ float sNorm[256];
    const int i = blk * blockDim.x + threadIdx.x;
    const bool live = i < n;
    const float g_raw = live ? grad[i] : 0.0f;
    float m = live ? m_in[i] : 0.0f, v = live ? v_in[i] : 0.0f, p = live ? p_in[i] : 0.0f;
    const float step = step_count[0], lr = lr_dev[0];
    float ss = 0.0f;
    for (int j0 = 0; j0 < n_norm; j0 += 256) {
        __syncthreads();
        if (j0 + (int)threadIdx.x < n_norm) sNorm[threadIdx.x] = norm_partial[j0 + threadIdx.x];
        __syncthreads();
        const int cnt = n_norm - j0 < 256 ? n_norm - j0 : 256;
        for (int j = 0; j < cnt; ++j) ss += sNorm[j];
    }
    const AdamCoef ac = adam_coef(ss, step, lr, max_norm, beta1, beta2, eps);
    if (!live) return;
    grad[i] = g_raw * ac.coef;             // clip_grad_norm_ scales the grads in place
    adam_elem(ac, g_raw, p, m, v);
    m_out[i] = m;
    v_out[i] = v;
    p_out[i] = p;
}

// (state in -> state out: the same tensors for the classic three-launch step, the deferred chain's current generation -> the
// caller's tensors at the end of an epoch)
__global__ __launch_bounds__(256) void adam_kernel(const float* __restrict__ p_in, const float* __restrict__ m_in, const float* __restrict__ v_in,
                                                   float* __restrict__ grad, float* __restrict__ p_out, float* __restrict__ m_out,
                                                   float* __restrict__ v_out, const float* __restrict__ step_count,
                                                   const float* __restrict__ lr_dev, const float* __restrict__ norm_partial,
                                                   const int n_norm, const int n, const float max_norm, const float beta1,
                                                   const float beta2, const float eps) {
    adam_body(blockIdx.x, p_in, m_in, v_in, grad, p_out, m_out, v_out, step_count, lr_dev, norm_partial, n_norm, n, max_norm, beta1, beta2, eps);
}


// K12m: clip_grad_norm_ + Adam for the MULTI-RANK step, after the gradient all-reduce: the bucket holds the SUM over ranks
// (grad_scale = 1 / world_size averages it), so the squared norm cannot come from K11's per-block partials.  One element per
// thread as in K12; every workgroup first sums the squares of the whole bucket itself (59 KB out of L2, the same fixed order in
// every workgroup and on every rank: replicas stay bit-identical) instead of one 1024-thread workgroup walking the bucket twice
// (pc_clip_adam).  The step counter has already been advanced by K11 (pc_ppo_minibatch with apply = 2).
__global__ __launch_bounds__(256) void clip_adam_mb_kernel(float* __restrict__ param, float* __restrict__ grad, float* __restrict__ exp_avg,
                                                           float* __restrict__ exp_avg_sq, const float* __restrict__ step_count,
                                                           const float* __restrict__ lr_dev, const int n, const float max_norm,
                                                           const float grad_scale, const float beta1, const float beta2, const float eps) {
    __shared__ float sh[16];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n;
    const float g_own = live ? grad[i] * grad_scale : 0.0f, m0 = live ? exp_avg[i] : 0.0f, v0 = live ? exp_avg_sq[i] : 0.0f;
    const float p0 = live ? param[i] : 0.0f;
    const float step = step_count[0], lr = lr_dev[0];
    // the bucket was written by another kernel (other XCDs' L2s): every load pays the fabric's latency, so all of a thread's
    // loads are issued before the first is used -- 16 x 16 bytes in flight cover 16 k floats per pass
    float ss = 0.0f;
    const bool vec = (reinterpret_cast<uintptr_t>(grad) & 15) == 0;
    const int n4 = vec ? n >> 2 : 0;
    const float4* __restrict__ g4 = reinterpret_cast<const float4*>(grad);
    for (int j0 = threadIdx.x; j0 < n4; j0 += 16 * 256) {
        float4 t[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) t[q] = j0 + 256 * q < n4 ? g4[j0 + 256 * q] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float a = t[q].x * grad_scale, b = t[q].y * grad_scale, c = t[q].z * grad_scale, d = t[q].w * grad_scale;
            ss += a * a; ss += b * b; ss += c * c; ss += d * d;
        }
    }
    for (int j = 4 * n4 + threadIdx.x; j < n; j += 256) { const float a = grad[j] * grad_scale; ss += a * a; }
    const AdamCoef ac = adam_coef(block_sum(ss, sh), step, lr, max_norm, beta1, beta2, eps);
    if (!live) return;
    // (the bucket itself is left as the all-reduce delivered it: other workgroups may still be reading it for their norm)
    float p = p0, m = m0, v = v0;
    adam_elem(ac, g_own, p, m, v);
    exp_avg[i] = m;
    exp_avg_sq[i] = v;
    param[i] = p;
}
