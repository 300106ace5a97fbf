// exchange.hpp -- part of the single translation unit ppocar.hip (included there, in order; not a stand-alone header).
// K13 xchg_allreduce_kernel: the per-minibatch gradient exchange (one all-reduce(SUM) of the flat bucket between
// loss.backward() and clip_grad_norm_, train.py:259-260) as a ONE-SHOT all-reduce over peer-mapped buffers.
//
// The bucket is 59 KB (14 858 floats at D = 23): over xGMI that is < 1 us of wire time per link, so a ring / tree schedule's
// several dependent hops are pure latency.  Here every rank WRITES its bucket straight into a slot of every peer's staging
// buffer (device memory mapped into this process through hipIpc: W - 1 independent xGMI writes, no hop depends on another),
// raises an arrival flag per peer, waits for its own W flags and sums the W slots locally in RANK ORDER -- every replica adds
// the same floats in the same order, so the reduced buckets are bit-identical on all ranks.
//
// Work decomposition: one workgroup per CHUNK of 1024 floats (256 lanes x float4), entirely independent of the other chunks:
// its own epoch counter, its own flags -- no grid-wide barrier.  Double buffered by epoch parity: a rank can start epoch k + 1
// (other parity) while a peer still sums epoch k, and cannot start k + 2 before that peer has raised its k + 1 flags, i.e. has
// finished summing k.  The staging memory is allocated uncached / fine-grained (hipExtMallocWithFlags) so that remote writes
// are visible to a running kernel; flags are released / acquired at system scope.  Ranks reach an exchange at different times
// (one may still be capturing its graph while the other already replays): a wait is patient -- the handle's timeout, 20 s of
// the device's real-time counter by default (pc_xchg_set_timeout) -- but not endless: when it expires it raises the handle's error word and the kernel
// finishes with a wrong sum (pc_xchg_status reports PC_ERR_TIMEOUT; the host layer checks it wherever it synchronises and aborts
// the job), and every later exchange on that handle skips its wait: the grid always drains.
#pragma once

constexpr int XCHG_MAX_RANKS = 8;
constexpr int XCHG_CHUNK = 1024;   // floats per workgroup
// The patience is measured with s_memrealtime: the 100 MHz reference clock, one counter for the whole device.  (s_memtime is a
// per-XCD shader-clock counter: a wave that the driver saves and restores while two processes time-slice one GPU -- the
// same-device rehearsal -- can resume where the counter reads something unrelated, and a "20 s" wait then expires at once.)
constexpr double XCHG_TICKS_PER_SECOND = 1.0e8;

struct XchgView {
    float* data[XCHG_MAX_RANKS];       // data[r]: rank r's staging area [2 parities][W writers][n_pad floats] (data[rank] is local)
    unsigned* flags[XCHG_MAX_RANKS];   // flags[r]: rank r's arrival flags [2][W][n_chunks]
    unsigned* epoch;                   // local [n_chunks]: this rank's call count per chunk
    int* error;                        // local [1]
    int rank, world, n, n_pad, n_chunks;
    unsigned long long timeout_ticks;  // patience of a wait, in s_memrealtime ticks (100 MHz)
};

__device__ __forceinline__ void xchg_allreduce_body(const XchgView& v, float* __restrict__ bucket, const int c) {
    const int tid = threadIdx.x;
    __shared__ unsigned s_ep;
    if (tid == 0) {
        s_ep = v.epoch[c] + 1u;
        v.epoch[c] = s_ep;
    }
    __syncthreads();
    const unsigned ep = s_ep;
    const int par = (int)(ep & 1u), W = v.world;
    const int i0 = c * XCHG_CHUNK + 4 * tid;                       // this lane's four floats (n_pad is a multiple of XCHG_CHUNK)
    // ---- phase 1: my chunk into slot [par][rank] of every rank (my own included), then one flag per rank
    f32x4 mine = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (i0 + j < v.n) mine[j] = bucket[i0 + j];
    for (int r = 0; r < W; ++r) {
        const int dst = (v.rank + r) % W;                           // every rank starts with a different peer
        *reinterpret_cast<f32x4*>(v.data[dst] + ((size_t)(par * W + v.rank)) * v.n_pad + i0) = mine;
    }
    __threadfence_system();
    __syncthreads();
    if (tid < W) __hip_atomic_store(v.flags[tid] + (par * W + v.rank) * v.n_chunks + c, ep, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // ---- phase 2: wait for the W writers of my chunk, then sum their slots in rank order
    if (tid < W) {
        const unsigned* f = v.flags[v.rank] + (par * W + tid) * v.n_chunks + c;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        bool dead = __hip_atomic_load(v.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;   // an earlier exchange gave up: do not wait again
        while (!dead && __hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != ep) {
            __builtin_amdgcn_s_sleep(32);
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (now > t0 && now - t0 > v.timeout_ticks) {   // give up, say so
                __hip_atomic_store(v.error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                dead = true;
            }
        }
    }
    __syncthreads();
    __threadfence_system();
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    const float* base = v.data[v.rank] + (size_t)(par * W) * v.n_pad + i0;
    for (int w = 0; w < W; ++w) {                                  // fixed order: the same sum, bit for bit, on every rank
        const unsigned* q = reinterpret_cast<const unsigned*>(base + (size_t)w * v.n_pad);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += __uint_as_float(__hip_atomic_load(q + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (i0 + j < v.n) bucket[i0 + j] = acc[j];
}

__global__ __launch_bounds__(256) void xchg_allreduce_kernel(const XchgView v, float* __restrict__ bucket) {
    xchg_allreduce_body(v, bucket, blockIdx.x);
}

// The exchanges of ALL ranks of an in-process group (pc_xchg_connect_local) in one launch: blockIdx.y = rank.  The W x n_chunks
// workgroups are co-resident by construction (8 x 23 at most), which separate launches on separate streams are not: HIP multiplexes
// streams onto a few hardware queues (4 by default), and ranks whose kernels queue behind each other would wait for flags that
// cannot be raised.  Same body, same slots, same flags as the per-rank kernel.
struct XchgGroup {
    XchgView v[XCHG_MAX_RANKS];
    float* bucket[XCHG_MAX_RANKS];
};
__global__ __launch_bounds__(256) void xchg_allreduce_group_kernel(const XchgGroup g) {
    xchg_allreduce_body(g.v[blockIdx.y], g.bucket[blockIdx.y], blockIdx.x);
}
