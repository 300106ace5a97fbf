// ppocar.hip -- HIP kernels (gfx950 / CDNA4) and the C-ABI of libppocar.so.
//
// Kernels
//   K1  env_step_kernel<T, RPL, MIXED>  the whole CarEnv.step transition (car_env.py:693-760) for one vector-env call,
//                                       with TransformReward and gymnasium's same-step auto-reset folded in
//                                       (train.py:65,68,185)
//   K2  env_reset_kernel<T>             CarEnv.reset for every env (car_env.py:605-691); reset_obs_kernel<T> computes
//                                       each track's constant reset observation once at create
//   K3  gae_kernel                      Buffer.calculate_advantages (buffer.py:36-64)
//   K4  sample_kernel                   Categorical(logits).sample / log_prob / entropy (model.py:35-40)
//   K5  policy_kernel<KS, SPLIT, PREC>  Agent.get_action_and_value(x) of the rollout (model.py:34-41): both MLPs on the
//                                       matrix cores + the draw; policy_pack*_kernel build its LDS weight image
//   K6-8 ppo_gather / ppo_loss / clip_adam   non-GEMM pieces of a PPO minibatch step (train.py:230-261)
//   K9  rollout_kernel<KS, RPL, PREC>   the whole rollout (train.py:173-195) as one persistent launch
//   K10-12 ppo_fwdbwd / grad_reduce / adam   one PPO minibatch step without any library GEMM
//
// Work decomposition of K1: an env is owned by G = 2^lg consecutive lanes of one
// wavefront ("lanes per env", chosen on the host from n_envs so the chip is filled); lane g of the
// group sweeps rays g, g+G, g+2G, ... (RPL = rays per lane, a template constant so the per-ray
// direction / running-minimum live in registers) against all wall segments.  Wall and gate
// segments are wave-uniform data: they are read with SCALAR loads (s_load_dwordx4 through the
// scalar cache) straight into SGPRs and enter the VALU as the one free SGPR operand per
// instruction -- cheaper than an LDS broadcast (no ds_read issue, no staging prologue, no
// barrier); a wave whose envs sit on different tracks runs the body once per distinct track
// (waterfall on the track id), so mixed-track batches stay correct.  Per-env reductions (any
// collision ray < 10 px) are DPP/shuffle butterflies inside the group; there is no LDS, no
// atomics and no inter-workgroup communication.  No MFMA: this is branchy fp32/fp64 geometry.
//
// Numerics
//   T = double  follows the reference's float64 operation order literally (Ray.cast :166-181,
//               np.linalg.norm's fused ddot tail, np.radians = x * (pi/180)); the translation unit
//               is compiled with -ffp-contract=off so nothing is fused behind our back.
//   T = float   the throughput path: float32 RAY GEOMETRY over a float64 kinematic state.  The
//               per-env scalar work (thrust, friction, clip, integrate, reward) is a few dozen
//               float64 operations and stays exactly the reference's; heading is an integer count
//               of 5-degree turns (Car.move_car only adds +-5.0, :440-442) looked up in a 72-entry
//               float64 cos/sin table built on the host; ray directions by float32 angle addition
//               with a per-ray table; ray casts in coordinates RELATIVE to the car, the difference
//               p1 - pos formed in float64 and then rounded (no pos+dir-pos cancellation,
//               :169-175), with u = cross(e,a)/cross(e,d) as the distance.  A float32 position
//               would by itself cost ~1e-4 px at x ~ 1280, i.e. most of the 1e-5 obs tolerance on
//               grazing rays (measured: DESIGN.md).
//               Rewards are bit-exact with the reference's float32(r * reward_scaling).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "ppocar_internal.h"

// ------------------------------------------------------------------------------------------
// device-side data
// ------------------------------------------------------------------------------------------
struct TrackHdr {        // one per track, read with scalar loads
    int wall_off, S;     // segs[wall_off .. wall_off+S): the walls
    int gate_off, G;     // segs[gate_off .. gate_off+G): the reward gates
    int head_off;        // F32: heading table [72] (cos, sin) of radians(start_rot + 5 j)
    int start_collides;  // Car.update at reset already hits a wall (car_env.py:686,468-469)
    int vtx_off, nV;     // F32: the walls again as vertex chains, vtx[vtx_off .. vtx_off+nV); nV is padded to a multiple of 4
    int dir_off;         // F32: ray direction table [361] of this track in dirtab (entry 360 = (0, 0): no ray)
    int rden_off;        // F32: 1/den table [361][nV] of this track in rden (row 360 = +inf: never hits)
    int n_chain, pad_;   // F32: chain vertices before the padding to a multiple of 4 (vtx[n_chain .. nV) are sentinels)
    double start_x, start_y, start_rot;
};

// One wall / gate segment as the reference holds it (Boundary.get_points, car_env.py:74): 32 bytes.
struct Seg { double x1, y1, x2, y2; };

// F32 wall sweep: the walls as chains of vertices.  Vertex k closes the segment (k-1, k) unless it
// starts a new chain (brk).  (ex, ey) = p[k-1] - p[k] rounded from float64.  32 bytes = one s_load_dwordx8.
// Why chains: the reference's hit test 0 < t < 1 (car_env.py:178) is "the two endpoints lie strictly on
// opposite sides of the ray line".  Evaluated per VERTEX -- one cross product c_k = cross(p_k - pos, dir)
// shared by the two segments that meet there -- a float32 ray cannot slip between two adjacent walls
// through the rounding-wide crack that two independently rounded t's leave at their common corner.
struct Vtx { double x, y; float ex, ey; int brk, pad; };

template <typename T> struct EnvParams {
    int64_t N;
    int lg;              // log2(lanes per env)
    int n_nominal;       // Car num_rays (car_env.py:227)
    int q;               // n // 4: stride of the collision rays (car_env.py:389)
    int step_deg;        // 360 // n (car_env.py:269)
    int R, D;            // actual ray count, obs dim 6 + R
    uint64_t colbits;    // bit r set <=> ray r < 64 is one of Car.check_collision's rays (r < n and r % (n // 4) == 0)
    double4* __restrict__ pv;               // [N] (px, py, vx, vy): kinematic state, float64 in BOTH modes
    int4* __restrict__ iv;                  // [N] (rot_k [F32 only], time_step, next_gate, passed)
    double* __restrict__ rot;               // [N] heading in degrees, F64 only
    const uint8_t* __restrict__ track_id;   // [N] or nullptr
    const TrackHdr* __restrict__ hdr;       // [n_tracks]
    const Seg* __restrict__ segs;           // walls and gates of all tracks
    const Vtx* __restrict__ vtx;            // F32 only: wall vertex chains of all tracks
    const double2* __restrict__ headtab;    // F32 only: (cos, sin) of radians(start_rot + 5 j), j < 72, per track
    // F32 only.  A ray's direction angle is start_rot + 5 k + step_deg * ray degrees (k = integer turn count): an integer
    // offset from start_rot, so all directions live on a 360-entry lattice per track.
    const float2* __restrict__ dirtab;      // [n_tracks][361] (cos, sin) of radians(start_rot + j), float64 libm, rounded
    const float* __restrict__ rden;         // [n_tracks][361][nV] 1 / (ey*dx - ex*dy) exactly as the sweep computes it (device-built)
    const float* __restrict__ reset_obs;    // [n_tracks][D]
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define PC_PI 3.141592653589793238462643383279502884 /* NPY_PI */

__device__ __forceinline__ double d_radians(double deg) { return deg * (PC_PI / 180.0); }  // np.radians

// ---- float64: the reference's own arithmetic --------------------------------------------------
// Ray.cast (car_env.py:155-184) + np.linalg.norm(pos - pt) (car_env.py:205).  Returns the hit
// distance, or 1000.0 (Ray.get_distance's `largest_distance`, :198) when there is no hit.
__device__ __forceinline__ double cast_ref(double x1, double y1, double x2, double y2, double x3, double y3,
                                           double dx, double dy) {
    const double x4 = x3 + dx, y4 = y3 + dy;                                  // :169
    const double den = (x1 - x2) * (y3 - y4) - (y1 - y2) * (x3 - x4);         // :171
    if (den == 0) return 1000.0;                                              // :172
    const double t = ((x1 - x3) * (y3 - y4) - (y1 - y3) * (x3 - x4)) / den;   // :175
    const double u = -((x1 - x2) * (y1 - y3) - (y1 - y2) * (x1 - x3)) / den;  // :176
    if (0 < t && t < 1 && u > 0) {                                            // :178
        const double ptx = x1 + t * (x2 - x1), pty = y1 + t * (y2 - y1);      // :180-181
        const double d0 = x3 - ptx, d1 = y3 - pty;
        return sqrt(fma(d1, d1, d0 * d0));  // np.linalg.norm -> cblas_ddot with a fused tail (see oracle)
    }
    return 1000.0;
}

// ---- float32: relative-coordinate cast -----------------------------------------------------------
// a_k = p_k - pos (formed in float64, then rounded: small near a wall, so nearly exact where it matters),
// c_k = cross(a_k, dir) = ay_k*dx - ax_k*dy, e = p1 - p2.  With the reference's t, u (car_env.py:171-176):
//   den = ey*dx - ex*dy = c1 - c2,   t = c1/den,   u = (ey*ax1 - ex*ay1)/den = un/den
//   0 < t < 1  <=>  c1 and c2 have strictly opposite signs  <=>  c1*c2 < 0
// and the distance |pos - pt| equals u because |dir| = 1.  den is formed from e directly (not as
// c1 - c2, which cancels badly for short far segments).  Parallel (den == 0, :172): c1 == c2, no hit.
// Returns min(best, hit distance): Ray.get_distance's running minimum (:203-207).
__device__ __forceinline__ float cross_f(float ax, float ay, float dx, float dy) {
    return __builtin_fmaf(ay, dx, -(ax * dy));
}
__device__ __forceinline__ float cast_fast(float best, float c1, float c2, float un, float ex, float ey, float dx,
                                           float dy) {
    const float den = __builtin_fmaf(ey, dx, -(ex * dy));
    const float u = un * __builtin_amdgcn_rcpf(den);
    // u > 0 and u < best in ONE compare: for non-negative floats the unsigned bit patterns order like the values,
    // and a negative (or NaN) u has the sign (or all exponent) bits set, i.e. compares above any finite best.
    // (u == +0 passes where the reference's u > 0 rejects: the ray origin exactly on a wall line.)
    const bool better = (c1 * c2 < 0.0f) & (__float_as_uint(u) < __float_as_uint(best));
    return better ? u : best;
}

template <typename T> struct Math;

template <> struct Math<double> {
    // heading (cos, sin): computed from the float64 heading as the reference does (:426-427, :584)
    static __device__ __forceinline__ void heading(const EnvParams<double>&, const TrackHdr&, int, double rot, double& c,
                                                   double& s) {
        const double a = d_radians(rot);
        c = cos(a);
        s = sin(a);
    }
    static __device__ __forceinline__ void ray_dir(const EnvParams<double>& p, const TrackHdr&, int ray, int, double rot,
                                                   double& dx, double& dy) {
        const double a = d_radians(rot + (double)(ray * p.step_deg));  // Ray.update(x, y, rot + a) :463-466, :153
        dx = cos(a);
        dy = sin(a);
    }
    // distance of one ray to one segment; (px, py) float64 ray origin
    static __device__ __forceinline__ double cast(const Seg& sg, double px, double py, double dx, double dy) {
        return cast_ref(sg.x1, sg.y1, sg.x2, sg.y2, px, py, dx, dy);
    }
    static __device__ __forceinline__ float norm_dist(double d) { return (float)(d / 1000.0); }             // :593,:595
    static __device__ __forceinline__ float norm(double v, double d) { return (float)(v / d); }             // :578-581
};

template <> struct Math<float> {
    static __device__ __forceinline__ int mod72(int k) {
        int m = k % 72;
        return m < 0 ? m + 72 : m;
    }
    // heading from the integer turn count: table of float64 cos/sin built on the host (glibc)
    static __device__ __forceinline__ void heading(const EnvParams<float>& p, const TrackHdr& h, int k, double, double& c,
                                                   double& s) {
        const double2 cs = p.headtab[h.head_off + mod72(k)];
        c = cs.x;
        s = cs.y;
    }
    // lattice index of ray `ray` at turn count k: (5 k + step_deg * ray) mod 360; 360 = "no ray"
    static __device__ __forceinline__ int dir_index(const EnvParams<float>& p, int k, int ray) {
        const int m = 5 * mod72(k) + ray * p.step_deg;  // ray * step_deg < 360 for every ray < R
        return m >= 360 ? m - 360 : m;
    }
    static __device__ __forceinline__ void ray_dir(const EnvParams<float>& p, const TrackHdr& h, int ray, int k, double,
                                                   float& dx, float& dy) {
        const float2 cs = p.dirtab[h.dir_off + dir_index(p, k, ray)];
        dx = cs.x;
        dy = cs.y;
    }
    static __device__ __forceinline__ float cast(const Seg& sg, double px, double py, float dx, float dy) {
        const float ax1 = (float)(sg.x1 - px), ay1 = (float)(sg.y1 - py);
        const float ax2 = (float)(sg.x2 - px), ay2 = (float)(sg.y2 - py);
        const float ex = (float)(sg.x1 - sg.x2), ey = (float)(sg.y1 - sg.y2);
        const float un = __builtin_fmaf(ey, ax1, -(ex * ay1));
        return cast_fast(1000.0f, cross_f(ax1, ay1, dx, dy), cross_f(ax2, ay2, dx, dy), un, ex, ey, dx, dy);
    }
    static __device__ __forceinline__ float norm_dist(float d) { return d * 0.001f; }
    // float64 multiply by the reciprocal, then the float32 cast: equals (float)(v / d) unless v/d sits
    // within 1e-16 (relative) of a float32 rounding boundary
    static __device__ __forceinline__ float norm(double v, double d) { return (float)(v * (1.0 / d)); }
};

// ------------------------------------------------------------------------------------------
// K1: env step
// ------------------------------------------------------------------------------------------
// Wave-uniform tables are read through the CONSTANT address space: the compiler then knows the memory is
// invariant and emits scalar loads (s_load_dwordx8 into SGPRs) even inside loops that also store to global
// memory (the mixed-track waterfall, the persistent rollout kernel), where its no-clobber analysis gives up.
template <typename S> __device__ __forceinline__ S cload(const S* ptr) {
    static_assert(sizeof(S) % 4 == 0, "word-sized records");
    typedef const __attribute__((address_space(4))) int* CI;
    const CI w = (CI)(const void*)ptr;
    int raw[sizeof(S) / 4];
#pragma unroll
    for (unsigned i = 0; i < sizeof(S) / 4; ++i) raw[i] = w[i];
    S out;
    __builtin_memcpy(&out, raw, sizeof(S));
    return out;
}

struct EnvRegs {  // one env's state, held identically by all lanes of its group
    double px, py, vx, vy, rot;
    int k, time, next, passed;
};

template <typename T> __device__ __forceinline__ EnvRegs env_load(const EnvParams<T>& p, const int64_t e) {
    const double4 sv = p.pv[e];
    const int4 si = p.iv[e];
    EnvRegs st;
    st.px = sv.x; st.py = sv.y; st.vx = sv.z; st.vy = sv.w;
    st.rot = 0.0;
    if constexpr (sizeof(T) == 8) st.rot = p.rot[e];
    st.k = si.x; st.time = si.y; st.next = si.z; st.passed = si.w;
    return st;
}

template <typename T> __device__ __forceinline__ void env_store(const EnvParams<T>& p, const int64_t e, const EnvRegs& st) {
    double4 ov;
    ov.x = st.px; ov.y = st.py; ov.z = st.vx; ov.w = st.vy;
    p.pv[e] = ov;
    p.iv[e] = make_int4(st.k, st.time, st.next, st.passed);
    if constexpr (sizeof(T) == 8) p.rot[e] = st.rot;
}

typedef const __attribute__((address_space(3))) float* lds_cfp;  // read-only float data in LDS (ds_read, not flat)

// 0x80000000 in an SGPR the optimiser cannot see through, and (a & m) | c as ONE instruction (the compiler splits the
// and-or when the mask is a literal: VOP3 takes no literals on gfx9).
__device__ __forceinline__ unsigned sign_mask() {
    unsigned m;
    asm("s_brev_b32 %0, 1" : "=s"(m));
    return m;
}
__device__ __forceinline__ unsigned and_or(unsigned a, unsigned m, unsigned c) {
    unsigned d;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(m), "v"(c));
    return d;
}

// Workgroup barrier for data exchanged through LDS only: waits for this wave's LDS traffic, NOT for its outstanding
// global stores (__syncthreads() waits vmcnt(0) too: ~1 us of store latency per barrier in the rollout loop).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- the float32 wall sweep: Car.get_distances (car_env.py:360-374) for the RPL ray slots of one lane against the vertex
// chain `vt` (nV vertices, a multiple of 4), part `part` of PARTS.  dx / dy = the slots' directions, didx = their lattice
// indices (TAB: rows of the 1/den table `rdl` in LDS).  bb = the slots' minimum distances as float bit patterns.
// ADDR: didx holds the LDS BYTE ADDRESSES of the slots' 1/den rows (env_step_fast's direction table delivers them) instead of
// lattice indices.
template <int RPL, int PARTS, bool TAB, bool ADDR = false>
__device__ __forceinline__ void wall_sweep_f32(const Vtx* vt, const int nV, const int part, const double npx, const double npy,
                                               const float (&dx)[RPL], const float (&dy)[RPL], const int (&didx)[RPL], lds_cfp rdl,
                                               unsigned (&bb)[2 * ((RPL + 1) / 2)]) {
    // Ray slots in PAIRS (packed fp32: one v_pk_* per two rays); an odd last slot is padded with a direction-0 ray that
    // never hits.  The running minimum is kept as the float's bit pattern: for non-negative floats unsigned order is
    // value order, so   best = min_u32(best, u_bits | sign(-(c1*c2)))   accepts u exactly when the endpoints lie
    // on strictly opposite sides of the ray line (c1*c2 < 0) AND 0 <= u < best -- a rejected candidate (same side,
    // u negative, u NaN) has its sign or all exponent bits set and compares above any finite best.  Two VALU
    // instructions per ray after the products instead of two compares and a select.
    // (u == +0 passes where the reference's u > 0 rejects: the ray origin exactly on a wall line.)
    constexpr int NP = (RPL + 1) / 2;
    f32x2 dx2[NP], dy2[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        dx2[j] = (f32x2){dx[2 * j], 2 * j + 1 < RPL ? dx[2 * j + 1] : 0.0f};
        dy2[j] = (f32x2){dy[2 * j], 2 * j + 1 < RPL ? dy[2 * j + 1] : 0.0f};
        bb[2 * j] = bb[2 * j + 1] = 0x447a0000u;  // 1000.0f, Ray.get_distance :198
    }
    const unsigned sgn = sign_mask();
    // side values of vertex k: a_k = p_k - pos (float64, then rounded), c_k = cross(a_k, dir) per ray
    auto side = [&](const Vtx& v, float& ax, float& ay, f32x2 (&c)[NP]) {
        ax = (float)(v.x - npx);
        ay = (float)(v.y - npy);
        const f32x2 ax2 = {ax, ax}, ay2 = {ay, ay};
#pragma unroll
        for (int j = 0; j < NP; ++j) c[j] = __builtin_elementwise_fma(ay2, dx2[j], -(ax2 * dy2[j]));
    };
    // vertex k closes the segment (k-1, k): (axp, ayp, cp) belong to k-1, c to k; rdv = the slots' 1/den (TAB)
    auto close = [&](const Vtx& v, const float axp, const float ayp, const f32x2 (&cp)[NP], const f32x2 (&c)[NP],
                     const float (&rdv)[2 * NP]) {
        const float un = __builtin_fmaf(v.ey, axp, -(v.ex * ayp));
        const f32x2 un2 = {un, un}, ex2 = {v.ex, v.ex}, ey2 = {v.ey, v.ey};
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            f32x2 u;
            if constexpr (TAB) {
                u = (f32x2){un * rdv[2 * j], un * rdv[2 * j + 1]};
            } else {
                const f32x2 den = __builtin_elementwise_fma(ey2, dx2[j], -(ex2 * dy2[j]));  // = rden_build_kernel's
                const f32x2 rc = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
                u = un2 * rc;
            }
            const f32x2 t = cp[j] * (-c[j]);  // sign clear <=> strictly opposite sides
            bb[2 * j] = min(bb[2 * j], and_or(__float_as_uint(t.x), sgn, __float_as_uint(u.x)));
            bb[2 * j + 1] = min(bb[2 * j + 1], and_or(__float_as_uint(t.y), sgn, __float_as_uint(u.y)));
        }
    };
    // Vertex GROUPS of four (the host pads every track's chain to a multiple of 4 with chain-break sentinels);
    // this part's groups [gbeg, gend).  The vertex before the range supplies the chain's previous side values.
    const int ngrp = nV >> 2;
    const int gbeg = PARTS > 1 ? ngrp * part / PARTS : 0;
    const int gend = PARTS > 1 ? ngrp * (part + 1) / PARTS : ngrp;
    // The "previous vertex" registers alternate between sets A and B (no copies); wave-uniform vertex records ->
    // s_load_dwordx8, prefetched one vertex ahead under the VALU work.
    float axA = 0.0f, ayA = 0.0f, axB = 0.0f, ayB = 0.0f;
    f32x2 cA[NP], cB[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) cA[j] = cB[j] = (f32x2){0.0f, 0.0f};
    if (PARTS > 1 && gbeg > 0) side(cload(vt + 4 * gbeg - 1), axA, ayA, cA);
    // TAB: one 16-byte LDS read per ray slot and group = the slot's 1/den for the group's four vertices
    typedef const __attribute__((address_space(3))) f32x4* lds_row;
    lds_row rrow[2 * NP];
    if constexpr (TAB) {
#pragma unroll
        for (int s = 0; s < 2 * NP; ++s) {
            if constexpr (ADDR) rrow[s] = (lds_row)(size_t)(s < RPL ? (unsigned)didx[s] : (unsigned)(size_t)rdl + 1440u * (unsigned)nV) + gbeg;
            else rrow[s] = (lds_row)(rdl + __umul24(s < RPL ? didx[s] : 360, nV)) + gbeg;  // full-rate 24-bit multiply
        }
    }
    Vtx nxt = cload(vt + (gbeg < gend ? 4 * gbeg : 0));
    // (A one-group-ahead prefetch of the table rows into a second register set was measured and dropped: inside the
    // benchmark, with the wave priorities, it is 1.5 % slower than loading each group's rows at its top.)
#define PC_VERTEX(RD, I, PAX, PAY, PC, NAX, NAY, NC)                                                                     \
    {                                                                                                                \
        const Vtx v = nxt;                                                                                           \
        nxt = cload(vt + (k + I + 1 < 4 * gend ? k + I + 1 : k + I));                                                \
        side(v, NAX, NAY, NC);                                                                                       \
        float rdv[2 * NP];                                                                                           \
        _Pragma("unroll") for (int s = 0; s < 2 * NP; ++s) rdv[s] = TAB ? RD[s][I] : 0.0f;                          \
        if (!v.brk) close(v, PAX, PAY, PC, NC, rdv);                                                                 \
    }
    for (int gq = gbeg; gq < gend; ++gq) {
        f32x4 rd[2 * NP];
        if constexpr (TAB) {
#pragma unroll
            for (int s = 0; s < 2 * NP; ++s) rd[s] = rrow[s][gq - gbeg];
        }
        const int k = 4 * gq;
        PC_VERTEX(rd, 0, axA, ayA, cA, axB, ayB, cB)
        PC_VERTEX(rd, 1, axB, ayB, cB, axA, ayA, cA)
        PC_VERTEX(rd, 2, axA, ayA, cA, axB, ayB, cB)
        PC_VERTEX(rd, 3, axB, ayB, cB, axA, ayA, cA)
    }
#undef PC_VERTEX
}

// The same sweep for a track whose chain has exactly NGRP groups of four vertices, fully unrolled and WITHOUT a branch per
// vertex (persistent big-form kernel: big_track has 24 walls in 2 loops = 26 chain vertices, padded to 28):
//   * the 1/den rows are read with immediate offsets (no address arithmetic per group);
//   * two consecutive vertices share one v_min3_u32 per ray slot instead of two v_min_u32;
//   * a chain-break vertex is not skipped but computed: its edge (ex, ey) is (0, 0), so un = 0 and 1/den = +-inf (what
//     rden_build_kernel's v_rcp_f32 of 0 stores, too), u = 0 * inf = NaN, whose bit pattern lies above every finite distance:
//     the candidate can never win the unsigned minimum.  Only the trailing padding pair(s) are skipped (n_chain).
// The minimum is exact, so the result is the very same bits as wall_sweep_f32's.
template <int RPL, bool TAB, int NGRP, bool ADDR = false>
__device__ __forceinline__ void wall_sweep_unrolled(const Vtx* vt, const int n_chain, const double npx, const double npy,
                                                    const float (&dx)[RPL], const float (&dy)[RPL], const int (&didx)[RPL], lds_cfp rdl,
                                                    unsigned (&bb)[2 * ((RPL + 1) / 2)]) {
    constexpr int NP = (RPL + 1) / 2, nV = 4 * NGRP;
    f32x2 dx2[NP], dy2[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        dx2[j] = (f32x2){dx[2 * j], 2 * j + 1 < RPL ? dx[2 * j + 1] : 0.0f};
        dy2[j] = (f32x2){dy[2 * j], 2 * j + 1 < RPL ? dy[2 * j + 1] : 0.0f};
        bb[2 * j] = bb[2 * j + 1] = 0x447a0000u;  // 1000.0f, Ray.get_distance :198
    }
    const unsigned sgn = sign_mask();
    auto side = [&](const Vtx& v, float& ax, float& ay, f32x2 (&c)[NP]) {
        ax = (float)(v.x - npx);
        ay = (float)(v.y - npy);
        const f32x2 ax2 = {ax, ax}, ay2 = {ay, ay};
#pragma unroll
        for (int j = 0; j < NP; ++j) c[j] = __builtin_elementwise_fma(ay2, dx2[j], -(ax2 * dy2[j]));
    };
    // candidates of the segment that vertex v closes: the hit distance's bits, with the sign bit set unless the segment's
    // endpoints lie on strictly opposite sides of the ray line (as wall_sweep_f32's `close`)
    auto cand = [&](const Vtx& v, const float axp, const float ayp, const f32x2 (&cp)[NP], const f32x2 (&c)[NP], const f32x4 (&rd)[2 * NP],
                    const int I, unsigned (&q)[2 * NP]) {
        const float un = __builtin_fmaf(v.ey, axp, -(v.ex * ayp));
        const f32x2 un2 = {un, un}, ex2 = {v.ex, v.ex}, ey2 = {v.ey, v.ey};
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            f32x2 u;
            if constexpr (TAB) {
                u = (f32x2){un * rd[2 * j][I], 2 * j + 1 < RPL ? un * rd[2 * j + 1][I] : 0.0f};   // (odd RPL: the last slot is padding)
            } else {
                const f32x2 den = __builtin_elementwise_fma(ey2, dx2[j], -(ex2 * dy2[j]));
                const f32x2 rc = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
                u = un2 * rc;
            }
            const f32x2 t = cp[j] * (-c[j]);
            q[2 * j] = and_or(__float_as_uint(t.x), sgn, __float_as_uint(u.x));
            if (2 * j + 1 < RPL) q[2 * j + 1] = and_or(__float_as_uint(t.y), sgn, __float_as_uint(u.y));
        }
    };
    float axA = 0.0f, ayA = 0.0f, axB = 0.0f, ayB = 0.0f;
    f32x2 cA[NP], cB[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) cA[j] = cB[j] = (f32x2){0.0f, 0.0f};
    typedef const __attribute__((address_space(3))) f32x4* lds_row;
    lds_row rrow[2 * NP];
    if constexpr (TAB) {
#pragma unroll
        for (int s = 0; s < 2 * NP; ++s) {
            if constexpr (ADDR) rrow[s] = (lds_row)(size_t)(s < RPL ? (unsigned)didx[s] : (unsigned)(size_t)rdl + 1440u * (unsigned)nV);
            else rrow[s] = (lds_row)(rdl + __umul24(s < RPL ? didx[s] : 360, nV));
        }
    }
#pragma unroll
    for (int gq = 0; gq < NGRP; ++gq) {
        f32x4 rd[2 * NP];
        if constexpr (TAB) {
#pragma unroll
            for (int s = 0; s < RPL; ++s) rd[s] = rrow[s][gq];
        }
#pragma unroll
        for (int I = 0; I < 4; I += 2) {
            if (gq == NGRP - 1 && 4 * gq + I >= n_chain) break;   // (wave-uniform; only the last group can hold a padding pair)
            const Vtx v0 = cload(vt + 4 * gq + I), v1 = cload(vt + 4 * gq + I + 1);
            unsigned q0[2 * NP], q1[2 * NP];
            side(v0, axB, ayB, cB);
            cand(v0, axA, ayA, cA, cB, rd, I, q0);
            side(v1, axA, ayA, cA);
            cand(v1, axB, ayB, cB, cA, rd, I + 1, q1);
#pragma unroll
            for (int s = 0; s < RPL; ++s) bb[s] = min(min(bb[s], q0[s]), q1[s]);   // v_min3_u32
        }
        // one scheduling region per group: left alone, the scheduler hoists every group's table rows and vertex records
        // to the top of the 1300-instruction block and spills
        __builtin_amdgcn_sched_barrier(0);
    }
}

// One CarEnv.step (car_env.py:693-760) + TransformReward + same-step auto-reset for the env whose state the
// 2^lg lanes of this group hold in `st` (updated in place, identically in every lane).  Lane g sweeps rays
// g, g + G, ...  Observation entries go to orow (global row), frow (pre-reset obs, optional) and lrow (an LDS
// copy for the persistent rollout kernel, optional).  The per-env scalars come back in registers.
template <typename T, int RPL, int PARTS = 1, bool TAB = false>
__device__ __forceinline__ void env_step_core(const EnvParams<T>& p, const int trk, const int g, const int lg, EnvRegs& st,
                                              const int64_t a, const double reward_scale, float* __restrict__ orow,
                                              float* __restrict__ frow, float* lrow, float& reward_f, bool& term, bool& trunc,
                                              int& passed_out, const int part = 0, float* exch = nullptr, lds_cfp rdl = nullptr) {
    // TAB (persistent kernels, when the track's 1/den table fits LDS): `rdl` = this track's [361][nV] table in LDS; the sweep
    // reads 1/den instead of forming den and its reciprocal (9 quarter-rate v_rcp_f32 per vertex otherwise) -- the table
    // holds exactly the bits the arithmetic path produces, so both paths are interchangeable.
    // PARTS > 1 (rollout_small_kernel): the env's wall sweep is split over PARTS waves of the workgroup -- this wave
    // sweeps vertex range `part`, the per-ray minima meet in LDS (`exch`: this env's [R][PARTS] floats) across ONE
    // workgroup barrier, and every wave then finishes the step on identical values (min is exact: bit-identical to
    // PARTS == 1).  Only part 0 stores.  Every thread of the workgroup must make the call.
    // trk is the same in every active lane; readfirstlane tells the compiler so
    const TrackHdr h = cload(p.hdr + __builtin_amdgcn_readfirstlane(trk));
    const int G = 1 << lg;
    const double rot_old = st.rot;

    // ---- action translation (car_env.py:698-722): thrust first with the PRE-turn heading, then the turn
    const bool fwd = (a == 0) | (a == 4) | (a == 5), bwd = (a == 1) | (a == 6) | (a == 7);
    const bool left = (a == 2) | (a == 4) | (a == 6), right = (a == 3) | (a == 5) | (a == 7);
    double ch0, sh0;  // heading before the turn
    Math<T>::heading(p, h, st.k, rot_old, ch0, sh0);
    double accx = 0.0, accy = 0.0;
    if (fwd) {  // Car.move_car("forward") :423-430
        accx = ch0 * 0.8;
        accy = sh0 * 0.8;
    } else if (bwd) {  // "backward" :431-438: -force_dir * 0.8
        accx = -ch0 * 0.8;
        accy = -sh0 * 0.8;
    }
    double rot_new = rot_old;
    int k_new = st.k;
    if (left) {  // :440
        rot_new -= 5.0;
        k_new -= 1;
    }
    if (right) {  // :442
        rot_new += 5.0;
        k_new += 1;
    }
    const bool turned = left | right;
    double ch1 = ch0, sh1 = sh0;  // heading after the turn
    if (turned) Math<T>::heading(p, h, k_new, rot_new, ch1, sh1);

    // ---- Car.update physics (car_env.py:452-461), float64 in both modes
    double nvx = st.vx + accx, nvy = st.vy + accy;  // :452
    if (!(fwd | bwd)) {                             // :454 ||acc|| == 0  <=>  no thrust
        nvx *= 1 - 0.2;                             // :455
        nvy *= 1 - 0.2;
    }
    nvx = nvx < -10.0 ? -10.0 : (nvx > 10.0 ? 10.0 : nvx);  // :457 np.clip per component
    nvy = nvy < -10.0 ? -10.0 : (nvy > 10.0 ? 10.0 : nvy);
    const double opx = st.px, opy = st.py;
    const double npx = opx + nvx, npy = opy + nvy;  // :459

    // ---- my rays: directions at the new pose; gate test at the OLD pose for the collision rays
    T dx[RPL], dy[RPL], best[RPL];
    int didx[RPL];  // F32: the slots' direction-lattice indices
    bool gate_hit = false;
    uint64_t colmask = 0;  // which of my ray slots are collision rays
    const Seg gate = p.segs[h.gate_off + st.next];  // only gate[next] can fire (the oracle does the full scan)
    // F32 lattice indices with adds only: ray * step_deg = g * step_deg + s * (G * step_deg), the second term wave-uniform;
    // (m mod 360) for m < 720 as min_u32(m, m - 360)
    const int rs0 = g * p.step_deg, gstep = G * p.step_deg;
    int k5_new = 0;
    if constexpr (sizeof(T) == 4) {
        k5_new = 5 * Math<float>::mod72(k_new) + rs0;
    }
#pragma unroll
    for (int s = 0; s < RPL; ++s) {
        const int ray = g + s * G;
        const bool valid = ray < p.R;
        const int rr = valid ? ray : 0;
        if constexpr (sizeof(T) == 4) {  // direction lattice: entry / row 360 = "no ray" (direction 0, 1/den = +inf)
            const unsigned m = (unsigned)(k5_new + s * gstep);
            didx[s] = valid ? (int)min(m, m - 360u) : 360;
            const float2 cs = p.dirtab[h.dir_off + didx[s]];
            dx[s] = cs.x;
            dy[s] = cs.y;
        } else {
            Math<T>::ray_dir(p, h, rr, k_new, rot_new, dx[s], dy[s]);
            if (!valid) {  // den == 0 for every segment -> never hits
                dx[s] = 0;
                dy[s] = 0;
            }
        }
        best[s] = (T)1000;  // Ray.get_distance :198
        // Car.check_collision's rays: r in range(0, n, n // 4) (:389) -- nominal n, not R
        // (host-built bitmask for rays < 64: a runtime modulo per ray slot costs ~20 VALU instructions)
        const bool is_col = valid & (ray < 64 ? (bool)((p.colbits >> ray) & 1) : ((ray < p.n_nominal) & (ray % p.q == 0)));
        colmask |= (uint64_t)is_col << s;
        if constexpr (sizeof(T) == 8) {
            if (is_col) {  // Car.get_passed_gate (:394-408) uses the rays of the PREVIOUS update
                T odx = dx[s], ody = dy[s];
                if (turned) Math<T>::ray_dir(p, h, rr, st.k, rot_old, odx, ody);
                gate_hit |= Math<T>::cast(gate, opx, opy, odx, ody) < (T)10;  // :387,:390
            }
        }
    }
    if constexpr (sizeof(T) == 4) {
        // Car.get_passed_gate (:394-408): the four collision rays j * (n // 4) at the PREVIOUS pose against gate[next].  Any
        // lane can cast any ray (directions come from the lattice table), so the four casts are dealt round-robin to the
        // env's lanes instead of falling on whichever lane owns those rays (with rays strided over the lanes: all on lane 0).
        const int k5o = 5 * Math<float>::mod72(st.k);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int j = g + jj * G;
            if (jj * G < 4) {  // uniform
                const unsigned m = (unsigned)(k5o + (j < 4 ? j : 0) * p.q * p.step_deg);
                const float2 cs = p.dirtab[h.dir_off + (int)min(m, m - 360u)];
                const bool hit = Math<float>::cast(gate, opx, opy, cs.x, cs.y) < 10.0f;  // :387,:390
                gate_hit |= hit & (j < 4);
            }
        }
    }

    // ---- wall sweep: Car.get_distances (:360-374) -- also serves Car.check_collision (E2)
    if constexpr (sizeof(T) == 4) {
        unsigned bb[2 * ((RPL + 1) / 2)];
        wall_sweep_f32<RPL, PARTS, TAB>(p.vtx + h.vtx_off, h.nV, part, npx, npy, dx, dy, didx, rdl, bb);
#pragma unroll
        for (int s = 0; s < RPL; ++s) best[s] = __uint_as_float(bb[s]);
    } else {
        const Seg* walls = p.segs + h.wall_off;
        Seg nxt = cload(walls);
        for (int w = 0; w < h.S; ++w) {
            const Seg sg = nxt;  // wave-uniform -> s_load_dwordx8
            nxt = cload(walls + (w + 1 < h.S ? w + 1 : w));
#pragma unroll
            for (int s = 0; s < RPL; ++s) {
                const double d = cast_ref(sg.x1, sg.y1, sg.x2, sg.y2, npx, npy, dx[s], dy[s]);
                if (d < best[s]) best[s] = d;  // :203-207
            }
        }
    }
    if constexpr (PARTS > 1) {
#pragma unroll
        for (int s = 0; s < RPL; ++s) {
            const int ray = g + s * G;
            if (ray < p.R) exch[ray * PARTS + part] = (float)best[s];
        }
        lds_barrier();
#pragma unroll
        for (int s = 0; s < RPL; ++s) {
            const int ray = g + s * G;
            if (ray < p.R) {
                float m = exch[ray * PARTS];
#pragma unroll
                for (int q = 1; q < PARTS; ++q) m = fminf(m, exch[ray * PARTS + q]);
                best[s] = (T)m;
            }
        }
    }
    const bool store = PARTS == 1 || part == 0;
    bool wall_hit = false;
#pragma unroll
    for (int s = 0; s < RPL; ++s) wall_hit |= ((colmask >> s) & 1) & (best[s] < (T)10);  // :390

    // ---- any() over the env's lanes: xor butterfly inside the 2^lg-lane group
    int flags = (gate_hit ? 1 : 0) | (wall_hit ? 2 : 0);
    for (int m = 1; m < G; m <<= 1) flags |= __shfl_xor(flags, m, 64);
    gate_hit = flags & 1;
    wall_hit = flags & 2;

    // ---- bookkeeping (car_env.py:694-750), float64 reward exactly as the reference accumulates it
    double rw = 0.0;
    if (fwd) rw += 0.01;  // :700,:710,:714
    int next = st.next, passed = st.passed;
    if (gate_hit) {               // :726 (gate.get_index() == next_gate_index by E1)
        rw += 1.0;                // :727
        if (next == h.G - 1) {    // :730 remaining == 0
            rw += 10.0;           // :732
            passed += 1;
            next = 0;             // :734-737
        } else {
            passed += 1;          // :740
            next += 1;            // :741
        }
    }
    const int time = st.time + 1;  // :745
    const bool destroyed = wall_hit | (h.start_collides != 0);
    term = false;
    trunc = false;
    if (destroyed) {  // :746-748
        term = true;
        rw -= 3.0;
    } else if (time >= 1000) {  // :749-750
        trunc = true;
    }
    const bool done = term | trunc;
    reward_f = (float)(rw * reward_scale);  // TransformReward then float32 store (buffer.py:29)
    passed_out = passed;

    // ---- observation.  Auto-reset (gymnasium 0.29.1 AsyncVectorEnv): a done env returns its reset obs.
    const float* __restrict__ robs = p.reset_obs + (size_t)trk * p.D;
#pragma unroll
    for (int s = 0; s < RPL; ++s) {
        const int ray = g + s * G;
        if (ray < p.R && store && orow) {
            const float v = Math<T>::norm_dist(best[s]);  // :593
            const float o = done ? robs[6 + ray] : v;
            orow[6 + ray] = o;
            if (lrow) lrow[6 + ray] = o;
            if (frow) frow[6 + ray] = v;
        }
    }
    if (g == 0 && store && orow) {
        float hd[6];
        hd[0] = Math<T>::norm(npx, 1280.0);  // :578-581
        hd[1] = Math<T>::norm(npy, 720.0);
        hd[2] = Math<T>::norm(nvx, 10.0);
        hd[3] = Math<T>::norm(nvy, 10.0);
        hd[4] = (float)ch1;  // :584-588
        hd[5] = (float)sh1;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const float o = done ? robs[i] : hd[i];
            orow[i] = o;
            if (lrow) lrow[i] = o;
            if (frow) frow[i] = hd[i];
        }
    }
    // ---- new state (every lane of the group keeps the same copy)
    if (done) {  // CarEnv.reset (:677-686): start pose, zero velocity, counters cleared
        st.px = h.start_x; st.py = h.start_y; st.vx = 0.0; st.vy = 0.0; st.rot = h.start_rot;
        st.k = 0; st.time = 0; st.next = 0; st.passed = 0;
    } else {
        st.px = npx; st.py = npy; st.vx = nvx; st.vy = nvy; st.rot = rot_new;
        st.k = k_new; st.time = time; st.next = next; st.passed = passed;
    }
}

template <typename T, int RPL>
__device__ __forceinline__ void env_step_body(const EnvParams<T>& p, const int trk, const int64_t e, const int g,
                                              const int64_t* __restrict__ actions, const double reward_scale,
                                              float* __restrict__ obs, float* __restrict__ reward,
                                              float* __restrict__ term_out, float* __restrict__ trunc_out,
                                              int32_t* __restrict__ gates_passed, float* __restrict__ final_obs) {
    // state in (coalesced 32/16-byte vectors; the G lanes of an env read the same address)
    EnvRegs st = env_load<T>(p, e);
    float rw;
    bool term, trunc;
    int passed;
    env_step_core<T, RPL>(p, trk, g, p.lg, st, actions[e], reward_scale, obs + (size_t)e * p.D,
                          final_obs ? final_obs + (size_t)e * p.D : nullptr, nullptr, rw, term, trunc, passed);
    if (g == 0) {
        reward[e] = rw;
        term_out[e] = term ? 1.0f : 0.0f;
        trunc_out[e] = trunc ? 1.0f : 0.0f;
        if (gates_passed) gates_passed[e] = passed;
        env_store<T>(p, e, st);
    }
}

// MIXED = false: every env is on track 0 -- straight-line body, all track data through scalar loads.
// MIXED = true : per-env track ids.  Waterfall: the body runs once per distinct track id present in the
// wavefront, so header / segment addresses stay wave-uniform.  (The loop is driven by a ballot of the lanes
// still to do: a plain readfirstlane(mine) is loop-invariant to the compiler and gets hoisted.)
template <typename T, int RPL, bool MIXED>
__global__ __launch_bounds__(256) void env_step_kernel(const EnvParams<T> p, const int64_t* __restrict__ actions,
                                                       const double reward_scale, float* __restrict__ obs,
                                                       float* __restrict__ reward, float* __restrict__ term_out,
                                                       float* __restrict__ trunc_out, int32_t* __restrict__ gates_passed,
                                                       float* __restrict__ final_obs) {
    const int64_t lane = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t e = lane >> p.lg;
    const int g = (int)(lane & ((1 << p.lg) - 1));
    if (e >= p.N) return;  // whole env groups leave together (N*G lanes are a multiple of G)
    if constexpr (!MIXED) {
        env_step_body<T, RPL>(p, 0, e, g, actions, reward_scale, obs, reward, term_out, trunc_out, gates_passed, final_obs);
    } else {
        const int mine = p.track_id[e];
        uint64_t todo = __ballot(1);
        while (todo) {
            const int first = __ffsll((unsigned long long)todo) - 1;
            const int cur = __builtin_amdgcn_readlane(mine, first);
            const bool match = mine == cur;
            if (match)
                env_step_body<T, RPL>(p, cur, e, g, actions, reward_scale, obs, reward, term_out, trunc_out, gates_passed,
                                      final_obs);
            todo &= ~__ballot(match);
        }
    }
}

// ------------------------------------------------------------------------------------------
// K2: reset, and the per-track reset observation
// ------------------------------------------------------------------------------------------
// CarEnv.reset (car_env.py:677-688) for ONE track: Car.reset + Car.update with zero velocity, then
// _get_obs.  One thread per track; runs once at pc_env_create.  Also reports start_collides.
// F32: the 1/den table of every track, rden[rden_off + idx * nV + k] for lattice direction idx and chain vertex k, by the
// very instructions the sweep uses (fma of the float32 edge and direction, v_rcp_f32): table and arithmetic path agree
// bit for bit.  Row 360 ("no ray") is +inf.
__global__ void rden_build_kernel(const EnvParams<float> p, const int n_tracks, float* __restrict__ rden) {
    for (int trk = 0; trk < n_tracks; ++trk) {
        const TrackHdr h = p.hdr[trk];
        const int total = 361 * h.nV;
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
            const int idx = i / h.nV, k = i - idx * h.nV;
            const Vtx v = p.vtx[h.vtx_off + k];
            const float2 d = p.dirtab[h.dir_off + idx];
            const float den = __builtin_fmaf(v.ey, d.x, -(v.ex * d.y));
            rden[h.rden_off + i] = idx == 360 ? __builtin_inff() : __builtin_amdgcn_rcpf(den);
        }
    }
}

template <typename T>
__global__ void reset_obs_kernel(const EnvParams<T> p, int n_tracks, float* __restrict__ reset_obs,
                                 int* __restrict__ start_collides) {
    const int trk = blockIdx.x * blockDim.x + threadIdx.x;
    if (trk >= n_tracks) return;
    const TrackHdr h = p.hdr[trk];
    double nvx = 0.0 + 0.0, nvy = 0.0 + 0.0;  // :452
    nvx *= 1 - 0.2;                           // :455 friction on zero velocity
    nvy *= 1 - 0.2;
    const double npx = h.start_x + nvx, npy = h.start_y + nvy;
    double ch, sh;
    Math<T>::heading(p, h, 0, h.start_rot, ch, sh);
    float* o = reset_obs + (size_t)trk * p.D;
    o[0] = Math<T>::norm(npx, 1280.0);
    o[1] = Math<T>::norm(npy, 720.0);
    o[2] = Math<T>::norm(nvx, 10.0);
    o[3] = Math<T>::norm(nvy, 10.0);
    o[4] = (float)ch;
    o[5] = (float)sh;
    bool hit = false;
    for (int ray = 0; ray < p.R; ++ray) {
        T dx, dy;
        Math<T>::ray_dir(p, h, ray, 0, h.start_rot, dx, dy);
        T best = (T)1000;
        for (int w = 0; w < h.S; ++w) {
            const T d = Math<T>::cast(p.segs[h.wall_off + w], npx, npy, dx, dy);
            if (d < best) best = d;
        }
        o[6 + ray] = Math<T>::norm_dist(best);
        if (ray < p.n_nominal && ray % p.q == 0 && best < (T)10) hit = true;
    }
    start_collides[trk] = hit ? 1 : 0;
}

template <typename T>
__global__ __launch_bounds__(256) void env_reset_kernel(const EnvParams<T> p, float* __restrict__ obs) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= p.N) return;
    const int trk = p.track_id ? p.track_id[e] : 0;
    const TrackHdr h = p.hdr[trk];
    double4 ov;
    ov.x = h.start_x; ov.y = h.start_y; ov.z = 0.0; ov.w = 0.0;
    p.pv[e] = ov;
    p.iv[e] = make_int4(0, 0, 0, 0);
    if constexpr (sizeof(T) == 8) p.rot[e] = h.start_rot;
    if (obs) {
        const float* r = p.reset_obs + (size_t)trk * p.D;
        float* o = obs + (size_t)e * p.D;
        for (int i = 0; i < p.D; ++i) o[i] = r[i];
    }
}

// CarEnv._get_info (car_env.py:599-603) of every env's CURRENT state: what the vector env's `infos` holds after a step
// (for an env that was auto-reset in that step: the reset state's counters, 0 / 0 -- gymnasium 0.29.1 moves the
// finished episode's info to "final_info"; its gates_passed is pc_env_step's `gates_passed` output).
__global__ __launch_bounds__(256) void env_info_kernel(const int4* __restrict__ iv, const int64_t N, int32_t* __restrict__ gates_passed,
                                                       int32_t* __restrict__ time_passed) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    const int4 s = iv[e];
    if (gates_passed) gates_passed[e] = s.w;
    if (time_passed) time_passed[e] = s.y;
}

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
        ex[i] = expf(l[i] - mx);
        sum += ex[i];
    }
    const float lse = mx + logf(sum);  // Categorical(logits=...) normalises: logits - logsumexp
    const float inv = 1.0f / sum;
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
// l = fp16(v - h): v_fma_mixlo/mixhi_f16 read h straight out of the packed pair (as fp16), form fma(h, -1, v) -- exact -- and
// round it once to fp16 into the low / high half: three instructions per pair of values, where converting h back, subtracting
// and converting again takes five (tools/split_mix_check.hip: the same bits on 4 M pairs, fp16-denormal residuals included).
__device__ __forceinline__ void split_pair_h(float a, float b, unsigned& p0, unsigned& p1) {
    p0 = pk_f16(a, b);
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(p1) : "v"(p0), "v"(a));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(p1) : "v"(p0), "v"(b));
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
                                                            const float* __restrict__ cb2, unsigned* __restrict__ image) {
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
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = clamp_h(v[j]);
            }
            const Pieces<PREC> sp = split8<PREC>(v);
            reinterpret_cast<u32x4*>(image)[i] = sp.p[pc];
        } else {
            const int b = i - n1 - n2;
            float v;
            if (b < 512) v = (b < HID ? ab1[b] : cb1[b - HID]) * PolScale<PREC>::sh;   // hidden layer's scaled domain
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
            const int ht = 2 * tp + j;
            const f32x4 bias = *reinterpret_cast<const f32x4*>(sB1 + 16 * ht + 4 * g);
            f32x4 hi[ET];
#pragma unroll
            for (int et = 0; et < ET; ++et) hi[et] = bias;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                // feature group 4 kb + g; groups >= NG are K padding (their B operand is all zeros): any finite A will do
                const int gi = 4 * kb + g, gA = gi < NG ? gi : NG - 1;
                u32x4 a[NP];
#pragma unroll
                for (int pc = 0; pc < NP; ++pc)
                    a[pc] = *reinterpret_cast<const u32x4*>(sW1p + (((ht * NP + pc) * NG + gA) * 16 + lc) * 4);
#pragma unroll
                for (int et = 0; et < ET; ++et) {
                    if constexpr (PREC == 2) hi[et] = mfma3(a, x[et][kb], hi[et]);
                    else hi[et] = mfma6(a, x[et][kb], hi[et]);
                }
            }
#pragma unroll
            for (int et = 0; et < ET; ++et) acc[j][et] = hi[et];
        }
    };
    // what follows layer 1 for an ACTOR tile pair (tp < 8): ReLU, operand split, layer 2 on the matrix cores
    auto epilogue_actor = [&](const int tp, const f32x4 (&acc)[2][ET]) {
        u32x4 w2[NP];
#pragma unroll
        for (int pc = 0; pc < NP; ++pc) w2[pc] = *reinterpret_cast<const u32x4*>(sW2p + (((tp * NP + pc) * 4 + g) * 10 + oA) * 4);
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
            const Pieces<PREC> h3 = split8<PREC>(hv);
            if constexpr (PREC == 2) out[et] = mfma3(w2, h3, out[et]);
            else out[et] = mfma6(w2, h3, out[et]);
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
    layer1(tp, accA);
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
        ex[i] = expf(v[i] - mx);
        sum += ex[i];
    }
    const float lse = mx + logf(sum);  // Categorical(logits=...) normalises: logits - logsumexp
    const float inv = 1.0f / sum;
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
    const float ex = i < A ? expf(l - mx) : 0.0f;
    float sum = ex;
    sum += PC_ROW_ROR(sum, 8);
    sum += PC_ROW_ROR(sum, 4);
    sum += PC_ROW_ROR(sum, 2);
    sum += PC_ROW_ROR(sum, 1);
    const float lse = mx + logf(sum);  // Categorical(logits=...) normalises: logits - logsumexp
    float cdf = ex * (1.0f / sum);     // inclusive scan over the row (lanes shifted in from outside the row read 0)
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
    constexpr int LD1 = pol_ld1(KS), LDO = 17, ET = 2;
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
        __syncthreads();  // previous pass's readers are done with sOut
#pragma unroll
        for (int et = 0; et < ET; ++et)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) myOut[(16 * et + lc) * LDO + 4 * lk + reg] = out[et][reg];
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

// ------------------------------------------------------------------------------------------
// Developer-only phase timeline of the big-form rollout kernel (make stamps -> libppocar_stamps.so, -DPC_STAMPS): workgroup 0's
// eight waves write s_memtime at every phase boundary of steps 64..71 into a device array that tools/k9_timeline.py reads back.
// The product build contains none of this.
#ifdef PC_STAMPS
constexpr int STAMP_T0 = 64, STAMP_NT = 8, STAMP_NPH = 8;
__device__ unsigned long long g_stamps[8 * STAMP_NT * STAMP_NPH];
#define PC_STAMP(ph)                                                                                                      \
    if (blockIdx.x == 0 && t >= STAMP_T0 && t < STAMP_T0 + STAMP_NT && lane == 0)                                        \
        g_stamps[((wave * STAMP_NT) + (t - STAMP_T0)) * STAMP_NPH + (ph)] = __builtin_amdgcn_s_memtime();
__device__ unsigned long long g_stamps_u[16];    // the minibatch kernel (K10), workgroup 0, thread 0, of the last launch
#define PC_STAMP_U(ph) if (wg == 0 && threadIdx.x == 0) g_stamps_u[ph] = __builtin_amdgcn_s_memtime();
#else
#define PC_STAMP(ph)
#define PC_STAMP_U(ph)
#endif
// ------------------------------------------------------------------------------------------
// The env step of the persistent big-form rollout (K9), single track, every gather table in LDS.
// Same arithmetic as env_step_core<float> -- its buffers are compared bit for bit with the per-step kernels' -- but laid out
// for a wave that owns its 32 envs outright (2 lanes per env) and whose cost is VALU issue slots, not latency:
//   * no branches: the action is decoded through a 16-entry table (thrust factor, friction factor, turn, forward bonus),
//     rewards / counters / the reset are selects;
//   * every table access is an explicit LDS read (ds_read), never a generic (flat) load -- those count on vmcnt AND lgkmcnt,
//     so each one used to wait for every global store the wave had in flight;
//   * the heading index is kept reduced mod 72 through a 74-entry wrap table instead of an integer division per step;
//   * an unused ray slot (17 rays on 2 lanes: 9 + 8) repeats the lane's last ray instead of being predicated off;
//   * the observation row goes to LDS only; the wave then copies its 32 rows -- contiguous in the rollout buffer -- to
//     global memory with 16-byte stores (three per lane instead of 23 scattered dword stores), and an env that finished its
//     episode gets its reset observation in a rarely taken, wave-uniformly skipped fix-up.
// ------------------------------------------------------------------------------------------
constexpr int TAB_MAX_GATES = 128;  // reward gates of a track staged in LDS (32 bytes each)
struct ActLut {          // one per action 0..15 (9..15: no-op, car_env.py:721), 32 bytes
    double thrust;       // acc = heading * thrust: +0.8 forward, -0.8 backward, 0 none (car_env.py:423-438)
    double fric;         // velocity factor after the thrust: 1 - 0.2 without thrust, 1 with (car_env.py:454-455)
    int dk;              // turn in 5-degree steps: -1 left, +1 right (car_env.py:440-442)
    int fwd;             // 1: the +0.01 forward bonus (car_env.py:700,710,714)
    int pad0, pad1;
};
// (plain ext_vector element types: a struct cannot be copied out of an address-space-qualified pointer in C++)
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) f64x2* lds_cd2;
typedef const __attribute__((address_space(3))) f64x4* lds_cd4;
typedef const __attribute__((address_space(3))) f32x2* lds_cf2;
typedef const __attribute__((address_space(3))) i32x2* lds_ci2;
typedef const __attribute__((address_space(3))) int* lds_ci;
typedef const __attribute__((address_space(3))) f32x4* lds_f4c;
typedef __attribute__((address_space(3))) float* lds_fp;

struct FastTabs {        // LDS addresses of the staged tables (wave-uniform)
    lds_cd2 head;        // [72] (cos, sin) of radians(start_rot + 5 j), float64
    lds_ci wrap;         // [74] j - 1 reduced mod 72, j = 0..73
    lds_cd2 act;         // [16] ActLut records, 32 bytes each: (thrust, fric) then (dk, fwd)
    lds_cd4 gates;       // [G] (x1, y1, x2, y2)
    lds_f4c dir;         // [720] direction lattice, twice around: (cos, sin, LDS byte address of the direction's 1/den row, -)
    lds_cfp reset;       // [D] the track's reset observation
    lds_cd2 vtx;         // [nV] the wall vertex chain, 32 bytes each: (x, y) float64 then (ex, ey, brk, -) -- small form only (nV <= 64)
    lds_cfp rden;        // [361][nV] or unused
};
constexpr int FT_HEAD = 0, FT_WRAP = FT_HEAD + 72 * 4, FT_ACT = FT_WRAP + 76, FT_GATES = FT_ACT + 16 * 8,
              FT_DIR = FT_GATES + TAB_MAX_GATES * 8, FT_RESET = FT_DIR + 720 * 4, FT_VTX = FT_RESET + 40,
              FT_VTX_MAX = 64, FT_FLOATS = FT_VTX + FT_VTX_MAX * 8;
static_assert(FT_ACT % 4 == 0 && FT_GATES % 4 == 0 && FT_DIR % 4 == 0 && FT_VTX % 4 == 0, "16-byte aligned records");

__device__ __forceinline__ FastTabs stage_fast_tables(const EnvParams<float>& p, const TrackHdr& h0, const int trk, float* sTab,
                                                      const int tid, const int nthreads) {
    int* dst = reinterpret_cast<int*>(sTab);
    const int* head = reinterpret_cast<const int*>(p.headtab + h0.head_off);
    for (int i = tid; i < 72 * 4; i += nthreads) dst[FT_HEAD + i] = head[i];
    for (int i = tid; i < 74; i += nthreads) dst[FT_WRAP + i] = i == 0 ? 71 : (i == 73 ? 0 : i - 1);
    if (tid < 16) {
        const int a = tid;
        const bool fwd = (a == 0) | (a == 4) | (a == 5), bwd = (a == 1) | (a == 6) | (a == 7);     // car_env.py:698-722
        const bool left = (a == 2) | (a == 4) | (a == 6), right = (a == 3) | (a == 5) | (a == 7);
        ActLut L;
        L.thrust = fwd ? 0.8 : (bwd ? -0.8 : 0.0);
        L.fric = (fwd | bwd) ? 1.0 : 1 - 0.2;
        L.dk = (left ? -1 : 0) + (right ? 1 : 0);
        L.fwd = fwd ? 1 : 0;
        L.pad0 = L.pad1 = 0;
        *reinterpret_cast<ActLut*>(sTab + FT_ACT + 8 * a) = L;
    }
    const int* gates = reinterpret_cast<const int*>(p.segs + h0.gate_off);
    for (int i = tid; i < h0.G * 8; i += nthreads) dst[FT_GATES + i] = gates[i];
    // The direction lattice twice around (a ray's index 5 k + step_deg * ray < 720 needs no reduction mod 360), each entry
    // with the LDS byte address of its row of the 1/den table: one 16-byte read per ray slot replaces the index arithmetic.
    const float2* dir = p.dirtab + h0.dir_off;
    const unsigned rden_base = (unsigned)(size_t)(lds_cfp)(sTab + FT_FLOATS);
    for (int i = tid; i < 720; i += nthreads) {
        const int j = i < 360 ? i : i - 360;
        const float2 cs = dir[j];
        *reinterpret_cast<f32x4*>(sTab + FT_DIR + 4 * i) = (f32x4){cs.x, cs.y, __uint_as_float(rden_base + (unsigned)(j * h0.nV) * 4u), 0.0f};
    }
    const int* ro = reinterpret_cast<const int*>(p.reset_obs + (size_t)trk * p.D);
    for (int i = tid; i < p.D; i += nthreads) dst[FT_RESET + i] = ro[i];
    if (h0.nV <= FT_VTX_MAX) {
        const int* vs = reinterpret_cast<const int*>(p.vtx + h0.vtx_off);
        for (int i = tid; i < h0.nV * 8; i += nthreads) dst[FT_VTX + i] = vs[i];
    }
    FastTabs ft;
    ft.head = (lds_cd2)(sTab + FT_HEAD);
    ft.wrap = (lds_ci)(sTab + FT_WRAP);
    ft.act = (lds_cd2)(sTab + FT_ACT);
    ft.gates = (lds_cd4)(sTab + FT_GATES);
    ft.dir = (lds_f4c)(sTab + FT_DIR);
    ft.reset = (lds_cfp)(sTab + FT_RESET);
    ft.vtx = (lds_cd2)(sTab + FT_VTX);
    ft.rden = (lds_cfp)(sTab + FT_FLOATS);
    return ft;
}

struct FastLane {        // per-lane invariants of the env step (a handful of registers instead of three per ray slot)
    int rs0, rstep, rs_last;  // ray slot s of lane g (of G per env) is ray min(g + G s, R - 1): angle offsets step_deg * ray, x 16 (bytes
                              // of the direction table), the table's LDS address folded into rs0 / rs_last
    int colmask;              // bit s: slot s is one of Car.check_collision's rays
    lds_fp lray, llast;       // this lane's first ray column of its observation row (slot s: + G s floats), and the last slot's
};
template <int RPL, int G>
__device__ __forceinline__ FastLane fast_lane(const EnvParams<float>& p, const FastTabs& ft, const int g, float* row) {
    FastLane fl;
    const int dir_base = (int)(size_t)ft.dir;
    fl.rs0 = dir_base + 16 * g * p.step_deg;
    fl.rstep = 16 * G * p.step_deg;
    fl.rs_last = dir_base + 16 * (p.R - 1) * p.step_deg;
    fl.colmask = 0;
#pragma unroll
    for (int s = 0; s < RPL; ++s) {
        const int ray = min(g + G * s, p.R - 1);
        // Car.check_collision's rays: r in range(0, n, n // 4) (:389) -- nominal n, not R
        const bool is_col = ray < 64 ? (bool)((p.colbits >> ray) & 1) : ((ray < p.n_nominal) & (ray % p.q == 0));
        fl.colmask |= (is_col ? 1 : 0) << s;
    }
    fl.lray = (lds_fp)(row + 6 + g);
    fl.llast = (lds_fp)(row + 6 + min(g + G * (RPL - 1), p.R - 1));
    return fl;
}
// exchange with the neighbouring lane (the other lane of the env): DPP quad_perm [1, 0, 3, 2], one VALU instruction
__device__ __forceinline__ int swap_pair(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xb1, 0xf, 0xf, false); }

// The float32 wall sweep of the SMALL persistent form: part `part` of PARTS of the vertex chain, read from its LDS copy
// (ft.vtx) instead of through scalar loads.  A part is only one or two groups of four vertices, so what counts is latency, not
// issue slots: a group's vertex records and 1/den rows are all requested at its top, the four vertices' side values are
// independent instruction chains, chain-break vertices are computed rather than branched around (their candidates are NaN: see
// wall_sweep_unrolled), and two vertices share a v_min3_u32.  Same bits as wall_sweep_f32<RPL, PARTS, TAB>.
template <int RPL, int PARTS, bool TAB, bool ADDR = false>
__device__ __forceinline__ void wall_sweep_lds(lds_cd2 vt, const int nV, const int part, const double npx, const double npy,
                                               const float (&dx)[RPL], const float (&dy)[RPL], const int (&didx)[RPL], lds_cfp rdl,
                                               unsigned (&bb)[2 * ((RPL + 1) / 2)]) {
    constexpr int NP = (RPL + 1) / 2;
    typedef const __attribute__((address_space(3))) f32x4* lds_f4;
    f32x2 dx2[NP], dy2[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        dx2[j] = (f32x2){dx[2 * j], 2 * j + 1 < RPL ? dx[2 * j + 1] : 0.0f};
        dy2[j] = (f32x2){dy[2 * j], 2 * j + 1 < RPL ? dy[2 * j + 1] : 0.0f};
        bb[2 * j] = bb[2 * j + 1] = 0x447a0000u;  // 1000.0f, Ray.get_distance :198
    }
    const unsigned sgn = sign_mask();
    auto side = [&](const f64x2 xy, float& ax, float& ay, f32x2 (&c)[NP]) {
        ax = (float)(xy.x - npx);
        ay = (float)(xy.y - npy);
        const f32x2 ax2 = {ax, ax}, ay2 = {ay, ay};
#pragma unroll
        for (int j = 0; j < NP; ++j) c[j] = __builtin_elementwise_fma(ay2, dx2[j], -(ax2 * dy2[j]));
    };
    auto cand = [&](const float ex, const float ey, const float axp, const float ayp, const f32x2 (&cp)[NP], const f32x2 (&c)[NP],
                    const f32x4 (&rd)[2 * NP], const int I, unsigned (&q)[2 * NP]) {
        const float un = __builtin_fmaf(ey, axp, -(ex * ayp));
        const f32x2 un2 = {un, un}, ex2 = {ex, ex}, ey2 = {ey, ey};
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            f32x2 u;
            if constexpr (TAB) {
                u = (f32x2){un * rd[2 * j][I], 2 * j + 1 < RPL ? un * rd[2 * j + 1][I] : 0.0f};   // (odd RPL: the last slot is padding)
            } else {
                const f32x2 den = __builtin_elementwise_fma(ey2, dx2[j], -(ex2 * dy2[j]));
                const f32x2 rc = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
                u = un2 * rc;
            }
            const f32x2 t = cp[j] * (-c[j]);
            q[2 * j] = and_or(__float_as_uint(t.x), sgn, __float_as_uint(u.x));
            if (2 * j + 1 < RPL) q[2 * j + 1] = and_or(__float_as_uint(t.y), sgn, __float_as_uint(u.y));
        }
    };
    const int ngrp = nV >> 2;
    const int gbeg = PARTS > 1 ? ngrp * part / PARTS : 0;
    const int gend = PARTS > 1 ? ngrp * (part + 1) / PARTS : ngrp;
    float axA = 0.0f, ayA = 0.0f, axB = 0.0f, ayB = 0.0f;
    f32x2 cA[NP], cB[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) cA[j] = cB[j] = (f32x2){0.0f, 0.0f};
    if (PARTS > 1 && gbeg > 0) side(vt[2 * (4 * gbeg - 1)], axA, ayA, cA);   // the vertex before the range: the chain's previous side values
    lds_f4 rrow[2 * NP];
    if constexpr (TAB) {
#pragma unroll
        for (int s = 0; s < 2 * NP; ++s) {
            if constexpr (ADDR) rrow[s] = (lds_f4)(size_t)(s < RPL ? (unsigned)didx[s] : (unsigned)(size_t)rdl + 1440u * (unsigned)nV);
            else rrow[s] = (lds_f4)(rdl + __umul24(s < RPL ? didx[s] : 360, nV));
        }
    }
    for (int gq = gbeg; gq < gend; ++gq) {
        f32x4 rd[2 * NP];
        if constexpr (TAB) {
#pragma unroll
            for (int s = 0; s < RPL; ++s) rd[s] = rrow[s][gq];
        }
        f64x2 xy[4];
        f32x4 ee[4];
#pragma unroll
        for (int I = 0; I < 4; ++I) {
            xy[I] = vt[2 * (4 * gq + I)];
            ee[I] = *(lds_f4)(vt + 2 * (4 * gq + I) + 1);
        }
#pragma unroll
        for (int I = 0; I < 4; I += 2) {
            unsigned q0[2 * NP], q1[2 * NP];
            side(xy[I], axB, ayB, cB);
            cand(ee[I].x, ee[I].y, axA, ayA, cA, cB, rd, I, q0);
            side(xy[I + 1], axA, ayA, cA);
            cand(ee[I + 1].x, ee[I + 1].y, axB, ayB, cB, cA, rd, I + 1, q1);
#pragma unroll
            for (int s = 0; s < RPL; ++s) bb[s] = min(min(bb[s], q0[s]), q1[s]);   // v_min3_u32
        }
    }
}

// LG = log2 of the lanes per env (1: K9, a wave owns 32 envs; 2: K9s, 16 envs per wave).  PARTS > 1 (K9s): the wall sweep is split
// over PARTS waves of the workgroup -- this wave sweeps vertex part `part`, the per-ray minima meet in LDS (`exch`: the env's
// [rays][PARTS] floats) across ONE workgroup barrier (every thread of the workgroup must make the call), and all waves finish
// the step on identical values; only `write_row` waves store the observation row.
template <int RPL, bool TAB, int LG = 1, int PARTS = 1>
__device__ __forceinline__ bool env_step_fast(const EnvParams<float>& p, const TrackHdr& h, const FastTabs& ft, const FastLane& fl,
                                              const int (&gq)[2], const int g,
                                              EnvRegs& st, int& k72, const int a, const double reward_scale, lds_fp lrow,
                                              float& reward_f, float& term_f, float& trunc_f, const int t = 0, const int lane = 0,
                                              const int wave = 0, const int part = 0, float* exch = nullptr, const bool write_row = true) {
    constexpr int G = 1 << LG;
    // ---- action, heading before and after the turn (car_env.py:698-722, :440-442)
    const f64x2 Lf = ft.act[2 * a];                                     // (thrust, fric)
    const i32x2 Li = *(lds_ci2)(ft.act + 2 * a + 1);                    // (dk, fwd)
    struct { double thrust, fric; int dk, fwd; } L = {Lf.x, Lf.y, Li.x, Li.y};
    const f64x2 cs0 = ft.head[k72];
    const int k72n = ft.wrap[k72 + L.dk + 1];
    const f64x2 cs1 = ft.head[k72n];
    // ---- Car.update physics (car_env.py:452-461), float64: thrust with the PRE-turn heading, friction without thrust, clip
    double nvx = (st.vx + cs0.x * L.thrust) * L.fric, nvy = (st.vy + cs0.y * L.thrust) * L.fric;
    nvx = fmin(fmax(nvx, -10.0), 10.0);      // np.clip per component (:457); the velocity is never NaN
    nvy = fmin(fmax(nvy, -10.0), 10.0);
    const double opx = st.px, opy = st.py;
    const double npx = opx + nvx, npy = opy + nvy;

    // ---- ray directions at the new heading: lattice entry 5 k + step_deg * ray (< 720: the table goes twice around), read as
    // (cos, sin, LDS address of the direction's 1/den row) through a byte address that costs one add per slot
    float dx[RPL], dy[RPL];
    int didx[RPL];    // TAB: the LDS byte address of the slot's 1/den row
    const int k80n = 80 * k72n;                                         // 16 bytes x 5 entries per turn step
    const int m0 = k80n + fl.rs0, m_last = k80n + fl.rs_last;
    {
        int m = m0;
#pragma unroll
        for (int s = 0; s < RPL; ++s) {
            // ray(s) = min(g + G s, R - 1): only the last slot can exceed the ray count
            const f32x4 cs = *(lds_f4c)(size_t)(unsigned)(s + 1 < RPL ? m : min(m, m_last));
            dx[s] = cs.x;
            dy[s] = cs.y;
            didx[s] = (int)__float_as_uint(cs.z);
            m += fl.rstep;
        }
    }
    // ---- Car.get_passed_gate (:394-408): the four collision rays at the PREVIOUS pose against gate[next], dealt over the lanes
    const f64x4 gv = ft.gates[st.next];
    const Seg gate = {gv.x, gv.y, gv.z, gv.w};
    const int k80o = 80 * k72;
    bool gate_hit = false;
#pragma unroll
    for (int jj = 0; jj < 4 / G; ++jj) {
        const f32x4 cs = *(lds_f4c)(size_t)(unsigned)(k80o + gq[jj]);
        gate_hit |= Math<float>::cast(gate, opx, opy, cs.x, cs.y) < 10.0f;  // :387,:390
    }
    // ---- wall sweep.  More than 12 ray slots per lane (33 rays: 17) are swept in TWO passes over the vertex chain, 9 + 8
    // slots: one pass would need ~40 more registers than the 256 a wave has at two waves per SIMD (it spilled 67 of them
    // to scratch); the second pass repeats only the per-vertex position arithmetic (4 of ~50 instructions per vertex and pass).
    constexpr int R1 = RPL > 12 ? (RPL + 1) / 2 : RPL, R2 = RPL - R1;
    unsigned bb[RPL + 2];
    PC_STAMP(4)
    {
        const float(&dxa)[R1] = *reinterpret_cast<const float(*)[R1]>(&dx[0]);
        const float(&dya)[R1] = *reinterpret_cast<const float(*)[R1]>(&dy[0]);
        const int(&dia)[R1] = *reinterpret_cast<const int(*)[R1]>(&didx[0]);
        unsigned ba[2 * ((R1 + 1) / 2)];
        if (PARTS > 1)                  // small form: latency-oriented sweep over the LDS copy of the chain
            wall_sweep_lds<R1, PARTS, TAB, true>(ft.vtx, h.nV, part, npx, npy, dxa, dya, dia, ft.rden, ba);
        else if (h.nV == 28)            // (wave-uniform) big_track's chain: the unrolled sweep
            wall_sweep_unrolled<R1, TAB, 7, true>(p.vtx + h.vtx_off, h.n_chain, npx, npy, dxa, dya, dia, ft.rden, ba);
        else
            wall_sweep_f32<R1, PARTS, TAB, true>(p.vtx + h.vtx_off, h.nV, part, npx, npy, dxa, dya, dia, ft.rden, ba);
#pragma unroll
        for (int s = 0; s < R1; ++s) bb[s] = ba[s];
    }
    if constexpr (R2 > 0) {
        __builtin_amdgcn_sched_barrier(0);   // the passes one after the other
        const float(&dxb)[R2] = *reinterpret_cast<const float(*)[R2]>(&dx[R1]);
        const float(&dyb)[R2] = *reinterpret_cast<const float(*)[R2]>(&dy[R1]);
        const int(&dib)[R2] = *reinterpret_cast<const int(*)[R2]>(&didx[R1]);
        unsigned bc[2 * ((R2 + 1) / 2)];
        if (PARTS > 1)
            wall_sweep_lds<R2, PARTS, TAB, true>(ft.vtx, h.nV, part, npx, npy, dxb, dyb, dib, ft.rden, bc);
        else if (h.nV == 28)
            wall_sweep_unrolled<R2, TAB, 7, true>(p.vtx + h.vtx_off, h.n_chain, npx, npy, dxb, dyb, dib, ft.rden, bc);
        else
            wall_sweep_f32<R2, PARTS, TAB, true>(p.vtx + h.vtx_off, h.nV, part, npx, npy, dxb, dyb, dib, ft.rden, bc);
#pragma unroll
        for (int s = 0; s < R2; ++s) bb[R1 + s] = bc[s];
    }
    if constexpr (PARTS > 1) {   // the parts' minima meet in LDS (min is exact: the same bits as one wave sweeping everything)
        unsigned* ex = reinterpret_cast<unsigned*>(exch);
        int ray = g;
#pragma unroll
        for (int s = 0; s < RPL; ++s) {
            const int r = s + 1 < RPL ? ray : min(ray, p.R - 1);
            ex[r * PARTS + part] = bb[s];
            ray += G;
        }
        lds_barrier();
        ray = g;
#pragma unroll
        for (int s = 0; s < RPL; ++s) {
            const int r = s + 1 < RPL ? ray : min(ray, p.R - 1);
            unsigned m = ex[r * PARTS];
#pragma unroll
            for (int q = 1; q < PARTS; ++q) m = min(m, ex[r * PARTS + q]);
            bb[s] = m;
            ray += G;
        }
    }
    PC_STAMP(5)
    // Car.check_collision (:376-392): any collision ray closer than 10 px.  Distances are non-negative floats, so the
    // smallest one is the unsigned minimum of the bit patterns; a slot that is not a collision ray is masked to +inf.
    unsigned hm = 0x7f800000u;
#pragma unroll
    for (int s = 0; s < RPL; ++s) {
        // bit s of colmask: slot s is one of Car.check_collision's rays; (bit ? 0 : 0x7f800000) without a register per slot
        const unsigned nc = ((unsigned)__builtin_amdgcn_sbfe(fl.colmask, s, 1) & 0x7f800000u) ^ 0x7f800000u;
        hm = min(hm, bb[s] | nc);
    }
    int flags = (gate_hit ? 1 : 0) | (hm < 0x41200000u ? 2 : 0);       // 0x41200000 = 10.0f
    flags |= swap_pair(flags);                                          // any() over the env's G lanes
    if constexpr (G == 4) flags |= __builtin_amdgcn_update_dpp(0, flags, 0x4e, 0xf, 0xf, false);   // quad_perm [2, 3, 0, 1]
    static_assert(G == 2 || G == 4, "2 or 4 lanes per env");
    gate_hit = flags & 1;
    const bool destroyed = ((flags & 2) != 0) | (h.start_collides != 0);
    // ---- bookkeeping (car_env.py:694-750): float64 reward in the reference's order of accumulation
    double rw = L.fwd ? 0.01 : 0.0;                                     // 0.0 + 0.01
    const bool lap = gate_hit & (st.next == h.G - 1);                   // :730 remaining == 0
    rw = rw + (gate_hit ? 1.0 : 0.0);                                   // :727
    rw = rw + (lap ? 10.0 : 0.0);                                       // :732
    const int passed = st.passed + (gate_hit ? 1 : 0);
    const int next = gate_hit ? (lap ? 0 : st.next + 1) : st.next;      // :734-741
    const int time = st.time + 1;                                       // :745
    rw = rw + (destroyed ? -3.0 : 0.0);                                 // :748
    const bool trunc = !destroyed & (time >= 1000);                     // :749-750
    const bool done = destroyed | trunc;
    reward_f = (float)(rw * reward_scale);
    term_f = destroyed ? 1.0f : 0.0f;
    trunc_f = trunc ? 1.0f : 0.0f;
    // ---- observation row -> LDS (the reset observation of a finished env is written by the caller's fix-up)
#pragma unroll
    for (int s = 0; s < RPL; ++s) {   // ray slot s -> column 6 + ray(s): G floats apart from the lane's first; the clamped last slot apart
        const float o = Math<float>::norm_dist(__uint_as_float(bb[s]));     // :593
        if (write_row) {
            if (s + 1 < RPL) fl.lray[G * s] = o;
            else fl.llast[0] = o;
        }
    }
    if (g == 0 && write_row) {
        lrow[0] = Math<float>::norm(npx, 1280.0);  // :578-581
        lrow[1] = Math<float>::norm(npy, 720.0);
        lrow[2] = Math<float>::norm(nvx, 10.0);
        lrow[3] = Math<float>::norm(nvy, 10.0);
        lrow[4] = (float)cs1.x;                    // :584-588
        lrow[5] = (float)cs1.y;
    }
    // ---- new state (CarEnv.reset for a finished env, :677-686, is the caller's rarely taken fix-up: env_reset_fast)
    st.px = npx;
    st.py = npy;
    st.vx = nvx;
    st.vy = nvy;
    st.k += L.dk;
    k72 = k72n;
    st.time = time;
    st.next = next;
    st.passed = passed;
    return done;
}

// CarEnv.reset (car_env.py:677-686) of a finished env's registers
__device__ __forceinline__ void env_reset_fast(const TrackHdr& h, EnvRegs& st, int& k72) {
    st.px = h.start_x; st.py = h.start_y; st.vx = 0.0; st.vy = 0.0;
    st.k = 0; st.time = 0; st.next = 0; st.passed = 0;
    k72 = 0;
}

// Developer-only timing ablation of the persistent rollout kernels: a SEPARATE build (make ABLATE=n -> libppocar_ablate.so,
// never loaded by the product or the tests) compiled with -DPC_ABLATE=n skips the policy MFMAs (1), the env step (2) or
// the draw (4).  The shipped library is built with PC_ABLATE = 0: there is no run-time switch that makes a kernel do less.
#ifndef PC_ABLATE
#define PC_ABLATE 0
#endif
// K9: the whole rollout (train.py:173-195) as ONE persistent launch.
// A workgroup (8 waves) owns 256 envs for all T steps: the policy weights stay in LDS, the env state in
// registers, the observation of step t passes from the env step to the policy step through LDS; per step an
// env costs 116 B of HBM writes (its buffer rows) and no reads.  Envs never interact and the weights are fixed
// during a rollout, so there is no inter-workgroup communication at all -- and no intra-workgroup one either:
// every WAVE owns 32 envs outright (policy step as one 32-column MFMA problem, then the env step of the same 32
// envs with 2 lanes per env), so after the weight image is staged there is not a single barrier.  The two waves
// that share a SIMD are started half a step apart, so one is in its matrix-core phase while the other is in its
// VALU phase (with PREC = 1 the policy GEMMs run on the bf16 matrix pipe and leave the fp32 ALUs to the env step).
//   P(t): X^T from LDS -> policy pass -> draw -> action to LDS, (act, logprob, value) rows t to HBM
//   E(t): action from LDS -> env_step_core -> obs row t+1 to HBM and LDS, (rew, term, trunc)
// Same arithmetic, same Philox counters as the policy_kernel / env_step_kernel pair: bit-identical buffers.
// ------------------------------------------------------------------------------------------
// Persistent rollout kernels: the small per-track tables every env step GATHERS from (heading table, reward gates, ray
// table, reset observation) are copied into LDS once and the EnvParams pointers redirected, so that a gather on the
// step's critical path costs an LDS access instead of a global-memory round trip.  Single-track batches only (the
// kernels' precondition).  The caller synchronises the workgroup before the first use.
constexpr int TAB_DIR = 72 * 4 + TAB_MAX_GATES * 8, TAB_RESET = TAB_DIR + 361 * 2 + 2, TAB_FLOATS = TAB_RESET + 40;
__device__ __forceinline__ EnvParams<float> stage_tables(const EnvParams<float>& p, float* sTab, const int tid, const int nthreads) {
    if (p.track_id) return p;  // mixed-track batch: the tables are read where they lie (global memory, L2-resident)
    const TrackHdr h0 = cload(p.hdr);
    int* dst = reinterpret_cast<int*>(sTab);
    EnvParams<float> q = p;
    const int* head = reinterpret_cast<const int*>(p.headtab + h0.head_off);
    for (int i = tid; i < 72 * 4; i += nthreads) dst[i] = head[i];
    q.headtab = reinterpret_cast<const double2*>(sTab) - h0.head_off;
    if (h0.G <= TAB_MAX_GATES) {
        const int* gates = reinterpret_cast<const int*>(p.segs + h0.gate_off);
        for (int i = tid; i < h0.G * 8; i += nthreads) dst[72 * 4 + i] = gates[i];
        q.segs = reinterpret_cast<const Seg*>(sTab + 72 * 4) - h0.gate_off;   // the F32 step reads only gates from segs
    }
    {
        const int* dir = reinterpret_cast<const int*>(p.dirtab + h0.dir_off);
        for (int i = tid; i < 361 * 2; i += nthreads) dst[TAB_DIR + i] = dir[i];
        q.dirtab = reinterpret_cast<const float2*>(sTab + TAB_DIR) - h0.dir_off;
    }
    if (p.D <= 40) {
        const int* ro = reinterpret_cast<const int*>(p.reset_obs);
        for (int i = tid; i < p.D; i += nthreads) dst[TAB_RESET + i] = ro[i];
        q.reset_obs = sTab + TAB_RESET;
    }
    return q;
}

// MODE 0: the gather tables where stage_tables puts them (generic pointers; mixed-track batches read them from global
//         memory), env step = env_step_core.  MODE 1 / 2: single track, A = 9, every table in LDS behind explicit LDS
//         pointers, env step = env_step_fast (2: with the 1/den table), observation rows copied out by the wave.
template <int KS, int RPL, int PREC, int MODE>
__global__ __launch_bounds__(512) void rollout_kernel(const EnvParams<float> p, const float* __restrict__ image, const int A,
                                                      const int T, const double reward_scale, const uint64_t seed,
                                                      const uint64_t offset, const uint64_t* __restrict__ offset_dev,
                                                      float* __restrict__ obs_buf, float* __restrict__ act_buf,
                                                      float* __restrict__ rew_buf, float* __restrict__ val_buf,
                                                      float* __restrict__ term_buf, float* __restrict__ trunc_buf,
                                                      float* __restrict__ logprob_buf, float* __restrict__ next_obs,
                                                      float* __restrict__ next_term, float* __restrict__ next_trunc,
                                                      const int rden_lds, const int epw, const int vec_ok) {
    constexpr int dbg = PC_ABLATE;  // 0 in the product build (see PC_ABLATE)
    constexpr int HID = 256, NT = 2 * HID / 16, LD1 = pol_ld1(KS), LDO = 17, ET = 2;
    constexpr int NG = pol_ng(KS), KB = pol_kb(KS);
    constexpr int IMG = PREC ? polx_image_dwords(PREC, NG) : pol_image_padded(KS);
    constexpr bool FAST = MODE != 0;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* sW1 = lds;
    float* sB1 = PREC ? lds + polx_w1_dwords(PREC, NG) + polx_w2_dwords(PREC) : sW1 + 2 * HID * LD1;
    float* sW2 = sB1 + 2 * HID;
    float* sB2 = PREC ? sB1 + 512 : sW2 + NT * 4 * 64;
    const unsigned* sW1p = reinterpret_cast<const unsigned*>(lds);
    const unsigned* sW2p = sW1p + polx_w1_dwords(PREC ? PREC : 1, NG);
    const float* sW2c = sB2 + 16;                  // PREC 1: critic output weights [256]
    const int64_t N = p.N;
    constexpr int DC = RPL == 6 ? 18 : (RPL == 9 ? 23 : 39);   // FAST: 6 + the ray count the 2-lanes-per-env menu implies (12 / 17 / 33)
    const int D = FAST ? DC : p.D;
    // observation of the step in flight, [256 envs][LDX]: FAST keeps the rows dense (LDX = D, exactly the rollout buffer's
    // layout: a wave's 32 rows are one contiguous block there and here)
    const int LDX = FAST ? D : 4 * KS + 1;
    float* sObs = lds + IMG;
    int* sAct = reinterpret_cast<int*>(sObs + 256 * LDX);
    float* sTab = reinterpret_cast<float*>(sAct + 256);    // staged per-track tables
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lc = lane & 15, lk = lane >> 4;
    policy_stage_image<IMG>(image, lds, tid);
    // FAST with a mixed-track batch: the host checked that every workgroup's envs lie on ONE track, whose tables it stages
    const int trk_wg = (FAST && p.track_id)
                           ? __builtin_amdgcn_readfirstlane((int)p.track_id[min((int64_t)blockIdx.x * epw, p.N - 1)]) : 0;
    const TrackHdr h0 = cload(p.hdr + trk_wg);
    EnvParams<float> q = p;
    FastTabs ft = {};
    if constexpr (FAST) ft = stage_fast_tables(p, h0, trk_wg, sTab, tid, 512);
    else q = stage_tables(p, sTab, tid, 512);
    // the track's 1/den table, when the host found room for it (rden_lds != 0; sized for the batch's largest track)
    float* sRden = sTab + (FAST ? FT_FLOATS : TAB_FLOATS);
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(p.rden + h0.rden_off);
        const int n4 = rden_lds ? 361 * h0.nV / 4 : 0;     // nV is a multiple of 4
        for (int i = tid; i < n4; i += 512) reinterpret_cast<f32x4*>(sRden)[i] = src[i];
    }
    const lds_cfp rdl = (lds_cfp)sRden;

    // this wave's 32 envs: local rows [pbase, pbase + 32); env-step identity: 2 lanes per env
    const int pbase = wave * 32;
    const int el = pbase + (lane >> 1), g = lane & 1;
    // epw = envs per workgroup: 256 (all 8 waves) or 128 (waves 4..7 only help to stage LDS and leave: at <= 32768 envs
    // that doubles the workgroups, one wave per SIMD on all 256 CUs instead of two on half of them)
    const int64_t e_wave = (int64_t)blockIdx.x * epw + pbase;      // first env of this wave
    const int64_t e_env = e_wave + (lane >> 1);
    const bool e_valid = e_env < N;
    EnvRegs st = {};
    if (e_valid) st = env_load<float>(p, e_env);
    // mixed-track batch: this wave's envs share one track (the host checked every aligned block of 32 envs)
    const int trk = p.track_id ? (int)p.track_id[e_valid ? e_env : N - 1] : 0;
    for (int f = g; f < (FAST ? D : 4 * KS); f += 2) sObs[el * LDX + f] = (e_valid && f < D) ? next_obs[e_env * D + f] : 0.0f;
    // this wave's output tile [32 envs][LDO] lives in its own observation rows: they are dead from the policy pass's
    // operand load until the env step stores the next observation (32 * LDX >= 32 * LDO floats: D >= 17 on the host's menu)
    static_assert(4 * KS + 1 >= 17, "the output tile must fit the wave's observation rows");
    float* myOut = sObs + wave * 32 * LDX;
    const uint64_t off0 = offset + (offset_dev ? *offset_dev : 0);
    PhiloxBlock rnd = {};  // the sampling lanes' current Philox block (4 steps' draws)
    // FAST: per-lane invariants of the env step.  Ray slot s of lane g is ray min(g + 2 s, R - 1): the odd slot that 17 or
    // 33 rays leave over on lane 1 repeats that lane pair's last ray (same value, same address) instead of being masked.
    int gq[2] = {0, 0}, k72 = 0;
    FastLane fl = {};
    if constexpr (FAST) {
        fl = fast_lane<RPL, 2>(p, ft, g, sObs + el * LDX);
        gq[0] = (int)(size_t)ft.dir + 16 * g * p.q * p.step_deg;            // Car.get_passed_gate's rays j * (n // 4), j = g and g + 2,
        gq[1] = (int)(size_t)ft.dir + 16 * (g + 2) * p.q * p.step_deg;      // as byte addresses into the direction table
        k72 = Math<float>::mod72(st.k);
    }
    const lds_fp lrow = (lds_fp)(sObs + el * LDX);
    __syncthreads();  // the weight image is in place; from here on the waves never synchronise again
    if (pbase >= epw) return;
    // (no deliberate phase offset between the two waves of a SIMD: with the priorities below they fall into opposite
    // phases by themselves; a start-up stagger measured 1 % slower)
    // The loop-carried env state came from global loads.  Passed through an empty asm it is, for the compiler's s_waitcnt
    // insertion, a fresh register value: waited for HERE, once -- otherwise the first use inside the loop carries a
    // conservative `s_waitcnt vmcnt(1)` on every iteration, i.e. a wait for the wave's own global stores of the step before.
    asm volatile("" : "+v"(st.px), "+v"(st.py), "+v"(st.vx), "+v"(st.vy), "+v"(st.k), "+v"(st.time), "+v"(st.next), "+v"(st.passed), "+v"(k72));

#pragma unroll 1
    for (int t = 0; t < T; ++t) {
        PC_STAMP(0)
        {
            // ---------------- P(t)
            f32x4 out[ET];
#pragma unroll
            for (int et = 0; et < ET; ++et) out[et] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
            if constexpr (PREC == 0) {
                float x[ET][KS];
#pragma unroll
                for (int et = 0; et < ET; ++et)
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        const int f = 4 * ks + lk;
                        x[et][ks] = (!FAST || f < D) ? sObs[(pbase + 16 * et + lc) * LDX + f] : 0.0f;
                    }
                // Wave priority: the policy pass (MFMA chains, whose results it waits for anyway) runs at the lowest priority (0),
                // the env step -- dense dependent VALU work -- above it (2), the short serial draw in between highest (3).  The two waves of a
                // SIMD are in opposite phases most of the time; with equal priorities the issue arbiter interleaves them
                // instruction by instruction and both crawl, with the env-step wave preferred the matrix pipe still gets
                // its instructions in the gaps.  Measured inside the benchmark's epochs: 20.5 -> 18.7 ms per rollout.
                __builtin_amdgcn_s_setprio(0);
                if (!(dbg & 1)) policy_pass<KS>(sW1, sB1, sW2, 0, NT, x, out, lc, lk, lane);  // dbg: timing ablations only
                __builtin_amdgcn_s_setprio(3);
            } else {
                Pieces<PREC> x[ET][KB];
#pragma unroll
                for (int et = 0; et < ET; ++et) {
#pragma unroll
                    for (int kb = 0; kb < KB; ++kb) {
                        float v[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int f = 8 * (4 * kb + lk) + j;
                            v[j] = f < D ? sObs[(pbase + 16 * et + lc) * LDX + f] : 0.0f;
                            if constexpr (PREC == 2) v[j] = clamp_h(v[j] * PolScale<PREC>::sx);   // the observations' scaled domain
                        }
                        x[et][kb] = split8<PREC>(v);
                    }
                }
                float val[ET] = {0.0f, 0.0f};
                __builtin_amdgcn_s_setprio(0);
                PC_STAMP(1)
                if (!(dbg & 1)) policy_pass16<PREC, KB>(sW1p, sW2p, sB1, sW2c, 0, NT / 2, x, out, val, lc, lk);
                PC_STAMP(2)
                __builtin_amdgcn_s_setprio(3);
#pragma unroll
                for (int et = 0; et < ET; ++et) {
                    float tv = val[et];
                    tv += __shfl_xor(tv, 16, 64);
                    tv += __shfl_xor(tv, 32, 64);
                    if (A >> 2 == lk) out[et][A & 3] += tv;
                }
            }
#pragma unroll
            for (int et = 0; et < ET; ++et)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) myOut[(16 * et + lc) * LDO + 4 * lk + reg] = out[et][reg];
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // the tile is written and read by this wave only
            __builtin_amdgcn_wave_barrier();
            const int64_t e = e_wave + lane;
            if (lane < 32 && e < N) {
                float v[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(myOut[lane * LDO + i], PolScale<PREC>::so_inv, sB2[i]);   // outputs back from their scaled domain
                int act;
                float lp, val;
                const uint64_t o = off0 + (uint64_t)t;
                if (t == 0 || (o & 3) == 0) rnd = philox_block(seed, o >> 2, (uint64_t)e);  // uniform: ten rounds per 4 steps
                if constexpr (FAST) policy_tail<9>(v, 9, philox_word_uniform(rnd, (unsigned)(o & 3)), act, lp, val, nullptr);
                else policy_tail(v, A, philox_word_uniform(rnd, (unsigned)(o & 3)), act, lp, val, nullptr);
                sAct[pbase + lane] = act;
                const int64_t row = (int64_t)t * N + e;
                act_buf[row] = (float)act;     // stored as float32 like the reference (buffer.py:13)
                logprob_buf[row] = lp;
                val_buf[row] = val;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        const bool last = t + 1 == T;
        PC_STAMP(3)
        if constexpr (FAST) {
            if (!(dbg & 2)) {
                __builtin_amdgcn_s_setprio(2);
                // ---------------- E(t)
                float rw, tf, cf;
                const int a = e_valid ? sAct[el] : 8;
                const bool done = env_step_fast<RPL, MODE == 2>(p, h0, ft, fl, gq, g, st, k72, a, reward_scale, lrow, rw, tf, cf, t, lane, wave);
                PC_STAMP(6)
                // gymnasium 0.29.1 same-step auto-reset: a finished env returns its reset observation
                if (__builtin_amdgcn_ballot_w64(done) != 0) {   // wave-uniform: ~1.5 % of env steps end an episode
                    if (done) {
                        for (int f = g; f < D; f += 2) lrow[f] = ft.reset[f];
                        env_reset_fast(h0, st, k72);
                    }
                }
                if (g == 0 && e_valid) {
                    rew_buf[(int64_t)t * N + e_env] = rw;
                    float* tr = last ? next_term : term_buf + (int64_t)(t + 1) * N;    // flags that precede obs t+1
                    float* tc = last ? next_trunc : trunc_buf + (int64_t)(t + 1) * N;  // (train.py:176-177,195)
                    tr[e_env] = tf;
                    tc[e_env] = cf;
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // the rows are complete (this wave wrote them all)
                __builtin_amdgcn_wave_barrier();
                // rows -> rollout buffer: the wave's 32 rows are contiguous there (32 * D floats), 16-byte stores when aligned
                float* dstg = (last ? next_obs : obs_buf + (int64_t)(t + 1) * N * D) + e_wave * D;
                const int64_t left = N - e_wave;                       // valid envs from this wave's first on
                const int n_rows = left >= 32 ? 32 : (int)left;
                const float* srcl = sObs + pbase * LDX;
                if (vec_ok && n_rows == 32) {
#pragma unroll
                    for (int j = 0; j < (8 * DC + 63) / 64; ++j) {      // 8 * D float4s: 3 (D = 18, 23) or 5 (D = 39) stores per lane
                        const int i = lane + 64 * j;
                        if (64 * j + 63 < 8 * DC || i < 8 * DC) reinterpret_cast<f32x4*>(dstg)[i] = reinterpret_cast<const f32x4*>(srcl)[i];
                    }
                } else {
                    for (int i = lane; i < n_rows * D; i += 64) dstg[i] = srcl[i];
                }
                PC_STAMP(7)
            }
        } else if (e_valid && !(dbg & 2)) {
            __builtin_amdgcn_s_setprio(2);
            // ---------------- E(t)
            float* orow = last ? next_obs + e_env * D : obs_buf + ((int64_t)(t + 1) * N + e_env) * D;
            float rw;
            bool term, trunc;
            int passed;
            if (rden_lds)  // uniform
                env_step_core<float, RPL, 1, true>(q, trk, g, 1, st, (int64_t)sAct[el], reward_scale, orow, nullptr, sObs + el * LDX, rw,
                                                   term, trunc, passed, 0, nullptr, rdl);
            else
                env_step_core<float, RPL>(q, trk, g, 1, st, (int64_t)sAct[el], reward_scale, orow, nullptr, sObs + el * LDX, rw, term,
                                          trunc, passed);
            if (g == 0) {
                rew_buf[(int64_t)t * N + e_env] = rw;
                float* tr = last ? next_term : term_buf + (int64_t)(t + 1) * N;    // flags that precede obs t+1
                float* tc = last ? next_trunc : trunc_buf + (int64_t)(t + 1) * N;  // (train.py:176-177,195)
                tr[e_env] = term ? 1.0f : 0.0f;
                tc[e_env] = trunc ? 1.0f : 0.0f;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // obs rows in LDS are this wave's own
        __builtin_amdgcn_wave_barrier();
    }
    if (e_valid && g == 0) env_store<float>(p, e_env, st);
}

// K9s: the same persistent rollout for SMALL batches (n_envs < ~32 k): a workgroup owns only 32 envs, so that
// n_envs / 32 workgroups fill the chip.  Per step: the 8 waves split the policy's hidden tiles exactly as
// policy_kernel<SPLIT> does (partial output tiles summed through LDS, same order: bit-identical), wave 0 draws the
// 32 actions, then all 512 lanes run the env step with 16 lanes per env.  Three workgroup barriers per step.
// MODE as in rollout_kernel: 0 = generic tables, env_step_core; 1 / 2 = single track, A = 9, every table in LDS behind LDS
// pointers, env_step_fast (2: with the 1/den table), dense observation rows copied out by three waves in 16-byte stores.
template <int KS, int RPL, int PREC, int MODE, int EPW>
__global__ __launch_bounds__(512) void rollout_small_kernel(const EnvParams<float> p, const float* __restrict__ image, const int A,
                                                            const int T, const double reward_scale, const uint64_t seed,
                                                            const uint64_t offset, const uint64_t* __restrict__ offset_dev,
                                                            float* __restrict__ obs_buf, float* __restrict__ act_buf,
                                                            float* __restrict__ rew_buf, float* __restrict__ val_buf,
                                                            float* __restrict__ term_buf, float* __restrict__ trunc_buf,
                                                            float* __restrict__ logprob_buf, float* __restrict__ next_obs,
                                                            float* __restrict__ next_term, float* __restrict__ next_trunc,
                                                            const int rden_lds, const int vec_ok) {
    constexpr int dbg = PC_ABLATE;  // 0 in the product build (see PC_ABLATE)
    // EPW = envs per workgroup: 32 (two groups of 4 waves = 4 sweep parts for 16 envs each; each wave 2 env tiles of the policy
    // pass) or 16 (up to 4096 envs: twice the workgroups -- all 256 CUs at BASELINE configs[1] -- and every phase of the step
    // half as long: all 8 waves = 8 sweep parts of the same 16 envs, one env tile per wave).
    constexpr int HID = 256, NT = 2 * HID / 16, LD1 = pol_ld1(KS), LDO = 17, ET = EPW / 16;
    constexpr int NG = pol_ng(KS), KB = pol_kb(KS);
    constexpr int IMG = PREC ? polx_image_dwords(PREC, NG) : pol_image_padded(KS);
    constexpr bool FAST = MODE != 0;
    constexpr int DC = RPL == 3 ? 18 : (RPL == 5 ? 23 : 39);   // FAST: 6 + the ray count the 4-lanes-per-env menu implies (12 / 17 / 33)
    static_assert(EPW == 32 || (EPW == 16 && FAST && PREC != 0 && DC <= 23), "16 envs per workgroup: fast mode, split operand forms, <= 17 rays");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* sW1 = lds;
    float* sB1 = PREC ? lds + polx_w1_dwords(PREC, NG) + polx_w2_dwords(PREC) : sW1 + 2 * HID * LD1;
    float* sW2 = sB1 + 2 * HID;
    float* sB2 = PREC ? sB1 + 512 : sW2 + NT * 4 * 64;
    const unsigned* sW1p = reinterpret_cast<const unsigned*>(lds);
    const unsigned* sW2p = sW1p + polx_w1_dwords(PREC ? PREC : 1, NG);
    const float* sW2c = sB2 + 16;
    const int64_t N = p.N;
    const int D = FAST ? DC : p.D;
    const int LDX = FAST ? D : 4 * KS + 1;
    float* sOut = lds + IMG;                       // [8 waves][EPW envs][LDO] partial output tiles
    float* sObs = sOut + 8 * EPW * LDO;            // [EPW envs][LDX]
    int* sAct = reinterpret_cast<int*>(sObs + EPW * (FAST ? 40 : LDX));   // (FAST: room for the widest row, so the tables stay 16-byte aligned)
    float* sTab = reinterpret_cast<float*>(sAct + 32);     // staged per-track tables
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lc = lane & 15, lk = lane >> 4;
    policy_stage_image<IMG>(image, lds, tid);
    // FAST with a mixed-track batch: the workgroup's envs lie on ONE track (every aligned block of 32 does), whose tables it stages
    const int trk_wg = (FAST && p.track_id)
                           ? __builtin_amdgcn_readfirstlane((int)p.track_id[min((int64_t)blockIdx.x * EPW, p.N - 1)]) : 0;
    const TrackHdr h0 = cload(p.hdr + trk_wg);
    EnvParams<float> q = p;
    FastTabs ft = {};
    if constexpr (FAST) ft = stage_fast_tables(p, h0, trk_wg, sTab, tid, 512);
    else q = stage_tables(p, sTab, tid, 512);
    // the track's 1/den table, when the host found room for it (rden_lds != 0; sized for the batch's largest track)
    float* sRden = sTab + (FAST ? FT_FLOATS : TAB_FLOATS);
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(p.rden + h0.rden_off);
        const int n4 = rden_lds ? 361 * h0.nV / 4 : 0;     // nV is a multiple of 4
        for (int i = tid; i < n4; i += 512) reinterpret_cast<f32x4*>(sRden)[i] = src[i];
    }
    const lds_cfp rdl = (lds_cfp)sRden;

    // env step: wave w sweeps part (w % PARTS) of the wall vertices for 16 envs, 4 lanes (ray groups) per env
    constexpr int PARTS = 128 / EPW;
    const int part = __builtin_amdgcn_readfirstlane(wave % PARTS);
    const int el = (wave / PARTS) * 16 + (lane >> 2), g = lane & 3;
    constexpr int EXS = EPW == 16 ? PARTS * (DC - 6) : PARTS * 34;   // floats per env: [rays][PARTS]
    static_assert(EPW * EXS <= 8 * EPW * LDO, "the exchange area aliases the partial output tiles");
    float* exch = sOut + el * EXS;                 // [rays][PARTS] of this env; aliases the partial output tiles (idle now)
    const int64_t e_wg = (int64_t)blockIdx.x * EPW;
    const int64_t e_env = e_wg + el;
    const bool e_valid = e_env < N;
    EnvRegs st = {};
    if (e_valid) st = env_load<float>(p, e_env);
    // mixed-track batch: this wave's envs share one track (the host checked every aligned block of 32 envs)
    const int trk = p.track_id ? (int)p.track_id[e_valid ? e_env : N - 1] : 0;
    for (int f = g + 4 * part; f < (FAST ? D : 4 * KS); f += 4 * PARTS) sObs[el * LDX + f] = (e_valid && f < D) ? next_obs[e_env * D + f] : 0.0f;
    float* myOut = sOut + wave * EPW * LDO;
    const int ht0 = wave * (NT / 8), ht1 = ht0 + NT / 8;
    const uint64_t off0 = offset + (offset_dev ? *offset_dev : 0);
    PhiloxBlock rnd = {};  // the sampling lanes' current Philox block (4 steps' draws)
    int gq[2] = {0, 0}, k72 = 0;
    FastLane fl = {};
    if constexpr (FAST) {
        fl = fast_lane<RPL, 4>(p, ft, g, sObs + el * LDX);
        gq[0] = (int)(size_t)ft.dir + 16 * g * p.q * p.step_deg;            // Car.get_passed_gate's ray j * (n // 4), j = g (byte address)
        k72 = Math<float>::mod72(st.k);
    }
    const lds_fp lrow = (lds_fp)(sObs + el * LDX);
    __syncthreads();
    asm volatile("" : "+v"(st.px), "+v"(st.py), "+v"(st.vx), "+v"(st.vy), "+v"(st.k), "+v"(st.time), "+v"(st.next), "+v"(st.passed), "+v"(k72));

#pragma unroll 1
    for (int t = 0; t < T; ++t) {
        PC_STAMP(0)
        // ---------------- P(t), hidden tiles [ht0, ht1) of this wave, all 32 envs
        f32x4 out[ET];
#pragma unroll
        for (int et = 0; et < ET; ++et) out[et] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        if constexpr (PREC == 0) {
            float x[ET][KS];
#pragma unroll
            for (int et = 0; et < ET; ++et)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int f = 4 * ks + lk;
                    x[et][ks] = (!FAST || f < D) ? sObs[(16 * et + lc) * LDX + f] : 0.0f;
                }
            if constexpr (ET == 2) {
                if (!(dbg & 1)) policy_pass<KS>(sW1, sB1, sW2, ht0, ht1, x, out, lc, lk, lane);  // dbg: timing ablations only
            }
        } else {
            Pieces<PREC> x[ET][KB];
#pragma unroll
            for (int et = 0; et < ET; ++et) {
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int f = 8 * (4 * kb + lk) + j;
                        v[j] = f < D ? sObs[(16 * et + lc) * LDX + f] : 0.0f;
                        if constexpr (PREC == 2) v[j] = clamp_h(v[j] * PolScale<PREC>::sx);   // the observations' scaled domain
                    }
                    x[et][kb] = split8<PREC>(v);
                }
            }
            float val[ET] = {};
            if (!(dbg & 1)) policy_pass16<PREC, KB, ET>(sW1p, sW2p, sB1, sW2c, wave, -1, x, out, val, lc, lk);
#pragma unroll
            for (int et = 0; et < ET; ++et) {
                float tv = val[et];
                tv += __shfl_xor(tv, 16, 64);
                tv += __shfl_xor(tv, 32, 64);
                if (A >> 2 == lk) out[et][A & 3] += tv;
            }
        }
#pragma unroll
        for (int et = 0; et < ET; ++et)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) myOut[(16 * et + lc) * LDO + 4 * lk + reg] = out[et][reg];
        PC_STAMP(1)
        lds_barrier();
        PC_STAMP(2)
        if (lk < EPW / 8) {   // every wave draws for EPW / 8 of the envs, 16 lanes (= outputs) per env, exactly as policy_kernel<SPLIT>
            const int dl = wave * (EPW / 8) + lk, oi = lc;
            const int64_t e = e_wg + dl;
            float ps = 0.0f;
#pragma unroll
            for (int w = 0; w < 8; ++w) ps += sOut[(w * EPW + dl) * LDO + oi];  // fixed order
            const float tsum = __builtin_fmaf(ps, PolScale<PREC>::so_inv, sB2[oi]);   // outputs back from their scaled domain
            const uint64_t o = off0 + (uint64_t)t;
            if (t == 0 || (o & 3) == 0) rnd = philox_block(seed, o >> 2, (uint64_t)e);  // uniform: ten rounds per 4 steps
            int act;
            float lp, val;
            if (!(dbg & 4)) policy_tail_row(tsum, oi, A, philox_word_uniform(rnd, (unsigned)(o & 3)), lane, act, lp, val);
            else { act = 0; lp = tsum; val = tsum; }
            if (oi == 0 && e < N) {
                sAct[dl] = act;
                const int64_t row = (int64_t)t * N + e;
                act_buf[row] = (float)act;
                logprob_buf[row] = lp;
                val_buf[row] = val;
            }
        }
        lds_barrier();
        PC_STAMP(3)
        // ---------------- E(t): 4 waves x 4 lanes per env (one more barrier inside, where the sweep parts meet)
        const bool last = t + 1 == T;
        if constexpr (FAST) {
            if (!(dbg & 2)) {
                float rw, tf, cf;
                const int a = e_valid ? sAct[el] : 8;
                const bool done = rden_lds   // (uniform)
                    ? env_step_fast<RPL, true, 2, PARTS>(p, h0, ft, fl, gq, g, st, k72, a, reward_scale, lrow, rw, tf, cf, t, lane, wave, part, exch, part == 0)
                    : env_step_fast<RPL, false, 2, PARTS>(p, h0, ft, fl, gq, g, st, k72, a, reward_scale, lrow, rw, tf, cf, t, lane, wave, part, exch, part == 0);
                if (__builtin_amdgcn_ballot_w64(done) != 0) {
                    if (done) {
                        if (part == 0)   // (uniform) the row-writing wave: reset observation of finished envs
                            for (int f = g; f < D; f += 4) lrow[f] = ft.reset[f];
                        env_reset_fast(h0, st, k72);
                    }
                }
                if (part == 0) {   // (uniform) the row-writing wave: per-env scalars
                    if (g == 0 && e_valid) {
                        rew_buf[(int64_t)t * N + e_env] = rw;
                        float* tr = last ? next_term : term_buf + (int64_t)(t + 1) * N;
                        float* tc = last ? next_trunc : trunc_buf + (int64_t)(t + 1) * N;
                        tr[e_env] = tf;
                        tc[e_env] = cf;
                    }
                }
            }
            PC_STAMP(6)
            lds_barrier();
            PC_STAMP(7)
            // rows -> rollout buffer: the workgroup's 32 rows are contiguous there (32 * D floats): waves 0 .. 2 (.. 4) store 64 float4 each
            {
                float* dstg = (last ? next_obs : obs_buf + (int64_t)(t + 1) * N * D) + e_wg * D;
                const int64_t left = N - e_wg;
                const int n_rows = left >= EPW ? EPW : (int)left;
                if (vec_ok && n_rows == EPW) {
                    const int i = lane + 64 * wave;
                    if (i < EPW / 4 * DC) reinterpret_cast<f32x4*>(dstg)[i] = reinterpret_cast<const f32x4*>(sObs)[i];
                } else {
                    for (int i = tid; i < n_rows * D; i += 512) dstg[i] = sObs[i];
                }
            }
        } else if constexpr (EPW == 32) {
            if (!(dbg & 2)) {
                float* orow = !e_valid ? nullptr : (last ? next_obs + e_env * D : obs_buf + ((int64_t)(t + 1) * N + e_env) * D);
                float rw;
                bool term, trunc;
                int passed;
                if (rden_lds)  // uniform
                    env_step_core<float, RPL, PARTS, true>(q, trk, g, 2, st, (int64_t)sAct[el], reward_scale, orow, nullptr,
                                                           e_valid && part == 0 ? sObs + el * LDX : nullptr, rw, term, trunc, passed, part,
                                                           exch, rdl);
                else
                    env_step_core<float, RPL, PARTS>(q, trk, g, 2, st, (int64_t)sAct[el], reward_scale, orow, nullptr,
                                                     e_valid && part == 0 ? sObs + el * LDX : nullptr, rw, term, trunc, passed, part, exch);
                if (e_valid && g == 0 && part == 0) {
                    rew_buf[(int64_t)t * N + e_env] = rw;
                    float* tr = last ? next_term : term_buf + (int64_t)(t + 1) * N;
                    float* tc = last ? next_trunc : trunc_buf + (int64_t)(t + 1) * N;
                    tr[e_env] = term ? 1.0f : 0.0f;
                    tc[e_env] = trunc ? 1.0f : 0.0f;
                }
            }
            lds_barrier();
        }
    }
    if (e_valid && g == 0 && part == 0) env_store<float>(p, e_env, st);
}

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

template <int DMAX>
__device__ __forceinline__ void ppo_fwdbwd_body(const int wg, const int64_t* __restrict__ idx, const int B, const int D, const int A,
                                                const float* __restrict__ obs, const float* __restrict__ act,
                                                const float* __restrict__ old_lp, const float* __restrict__ adv,
                                                const float* __restrict__ ret, const float* __restrict__ param,
                                                const float clip, const float vf, const float ec,
                                                float* __restrict__ partial, float* __restrict__ metric_partial,
                                                const float* __restrict__ prep) {
    constexpr int H = 256, S = FB_S, LDT = DMAX + 1, LDH = H + 1;
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
    float w2a[16];
#pragma unroll
    for (int o = 0; o < 16; ++o) w2a[o] = o < A ? param[o_aW2 + o * H + u] : 0.0f;
    const float w2c = param[o_cW2 + u], b1a = param[o_ab1 + u], b1c = param[o_cb1 + u];
    const int ob = u & 15;                                                // my output index in the layer-2 epilogue
    const float b2 = ob < A ? param[o_ab2 + ob] : (ob == A ? param[o_cb2] : 0.0f);
    // W1 of both nets with 16-byte loads: the [H][D] block at parameter offset `off` is fetched as the aligned float4 window
    // [off & ~3, off + H D) -- NV4 loads per thread and net instead of D dword loads (the kernel's memory instructions were a
    // third of its time: profiles/, K10 phase stamps) -- and goes through the LDS tile in that same natural order.
    constexpr int NV4 = (H * DMAX + 3 + 1023) / 1024 + 1;
    f32x4 w1raw4[2][NV4];
    int w1_shift[2], w1_n4[2];
#pragma unroll
    for (int net = 0; net < 2; ++net) {
        const int off = net == 0 ? o_aW1 : o_cW1, b4 = off & ~3;
        w1_shift[net] = off - b4;
        w1_n4[net] = (off + H * D - b4 + 3) >> 2;
        const f32x4* __restrict__ src = reinterpret_cast<const f32x4*>(param + b4);
#pragma unroll
        for (int j = 0; j < NV4; ++j) {
            const int i4 = u + 256 * j;
            w1raw4[net][j] = i4 < w1_n4[net] ? src[i4] : (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        }
    }
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
#pragma unroll
    for (int net = 0; net < 2; ++net) {
        lds_barrier();  // (not __syncthreads(): that waits for the stores already in flight, ~2 us each time)
#pragma unroll
        for (int f = 0; f < DMAX; ++f)
            if (f < D) sT[u * D + f] = net == 0 ? g1a[f] : g1c[f];
        lds_barrier();  // (not __syncthreads(): that waits for the stores already in flight, ~2 us each time)
        f32x4* __restrict__ dst = reinterpret_cast<f32x4*>(P + net * HD);
#pragma unroll
        for (int j = 0; j < NV4; ++j) {
            const int i4 = u + 256 * j;
            if (i4 < (HD >> 2)) dst[i4] = reinterpret_cast<const f32x4*>(sT)[i4];
        }
    }
    PC_STAMP_U(7)
}

template <int DMAX>
__global__ __launch_bounds__(256) void ppo_fwdbwd_kernel(const int64_t* __restrict__ idx, const int B, const int D, const int A,
                                                         const float* __restrict__ obs, const float* __restrict__ act,
                                                         const float* __restrict__ old_lp, const float* __restrict__ adv,
                                                         const float* __restrict__ ret, const float* __restrict__ param,
                                                         const float clip, const float vf, const float ec,
                                                         float* __restrict__ partial, float* __restrict__ metric_partial,
                                                         const float* __restrict__ prep) {
    ppo_fwdbwd_body<DMAX>(blockIdx.x, idx, B, D, A, obs, act, old_lp, adv, ret, param, clip, vf, ec, partial, metric_partial, prep);
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
    float g = 0.0f;
    if (i < n) {
        // natural flat index i -> index inside a partial (ppo_fwdbwd_body's layout: both [H][D] blocks first, 16-byte aligned)
        const int pm = i < HD ? i : (i < mid_end ? i + HD : (i < mid_end + HD ? i - (mid_end - HD) : i));
        const float* __restrict__ pp = partial + pm;
        int pidx = 0;
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
            sMp[threadIdx.x] = p0 * 4 + (int)threadIdx.x < n_part * 4 ? metric_partial[p0 * 4 + threadIdx.x] : 0.0f;
            __syncthreads();
            if (threadIdx.x < 3)
                for (int pidx = 0; pidx < 64 && p0 + pidx < n_part; ++pidx) mt[threadIdx.x] += sMp[pidx * 4 + threadIdx.x];
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
__device__ __forceinline__ void adam_body(const int blk, float* __restrict__ param, float* __restrict__ grad, float* __restrict__ exp_avg,
                                                   float* __restrict__ exp_avg_sq, const float* __restrict__ step_count,
                                                   const float* __restrict__ lr_dev, const float* __restrict__ norm_partial,
                                                   const int n_norm, const int n, const float max_norm, const float beta1,
                                                   const float beta2, const float eps) {
    // all loads first (cold misses: the operands were written by other workgroups), the norm partials once per
    // workgroup through LDS; every thread then sums them in index order
    __shared__ float sNorm[256];
    const int i = blk * blockDim.x + threadIdx.x;
    const bool live = i < n;
    const float g_raw = live ? grad[i] : 0.0f, m0 = live ? exp_avg[i] : 0.0f, v0 = live ? exp_avg_sq[i] : 0.0f;
    const float p0 = live ? param[i] : 0.0f;
    const float step = step_count[0], lr = lr_dev[0];
    float ss = 0.0f;
    for (int j0 = 0; j0 < n_norm; j0 += 256) {
        __syncthreads();
        if (j0 + (int)threadIdx.x < n_norm) sNorm[threadIdx.x] = norm_partial[j0 + threadIdx.x];
        __syncthreads();
        const int cnt = n_norm - j0 < 256 ? n_norm - j0 : 256;
        for (int j = 0; j < cnt; ++j) ss += sNorm[j];
    }
    const float coef = fminf(max_norm / (sqrtf(ss) + 1e-6f), 1.0f);
    const float bc1 = 1.0f - powf(beta1, step), bc2_sqrt = sqrtf(1.0f - powf(beta2, step));
    const float step_size = lr / bc1;
    if (!live) return;
    const float g = g_raw * coef;
    grad[i] = g;
    const float m = m0 + (1.0f - beta1) * (g - m0);
    const float v = beta2 * v0 + (1.0f - beta2) * g * g;
    exp_avg[i] = m;
    exp_avg_sq[i] = v;
    param[i] = p0 - step_size * (m / (sqrtf(v) / bc2_sqrt + eps));
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ param, float* __restrict__ grad, float* __restrict__ exp_avg,
                                                   float* __restrict__ exp_avg_sq, const float* __restrict__ step_count,
                                                   const float* __restrict__ lr_dev, const float* __restrict__ norm_partial,
                                                   const int n_norm, const int n, const float max_norm, const float beta1,
                                                   const float beta2, const float eps) {
    adam_body(blockIdx.x, param, grad, exp_avg, exp_avg_sq, step_count, lr_dev, norm_partial, n_norm, n, max_norm, beta1, beta2, eps);
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
    const float total_norm = sqrtf(block_sum(ss, sh));
    const float coef = fminf(max_norm / (total_norm + 1e-6f), 1.0f);   // clip_coef_clamped
    const float bc1 = 1.0f - powf(beta1, step), bc2_sqrt = sqrtf(1.0f - powf(beta2, step));
    const float step_size = lr / bc1;
    if (!live) return;
    // (the bucket itself is left as the all-reduce delivered it: other workgroups may still be reading it for their norm)
    const float g = g_own * coef;
    const float m = m0 + (1.0f - beta1) * (g - m0);
    const float v = beta2 * v0 + (1.0f - beta2) * g * g;
    exp_avg[i] = m;
    exp_avg_sq[i] = v;
    param[i] = p0 - step_size * (m / (sqrtf(v) / bc2_sqrt + eps));
}


// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static thread_local std::string g_hip_err;

#define HIPCHK(expr)                                                                       \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            g_hip_err = std::string(#expr) + ": " + hipGetErrorString(_e);                 \
            return PC_ERR_HIP;                                                             \
        }                                                                                  \
    } while (0)

namespace {

struct DeviceGuard {  // set the handle's device for the call, restore the caller's afterwards
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
        dev_ = dev;
    }
    ~DeviceGuard() {
        if (prev >= 0 && prev != dev_) (void)hipSetDevice(prev);
    }
    int dev_;
};

constexpr int kMenu[] = {1, 2, 3, 5, 6, 9, 12, 17, 33};  // rays-per-lane instantiations of K1

int pick_rpl(int need) {
    for (int m : kMenu)
        if (m >= need) return m;
    return -1;
}

}  // namespace

struct pc_env {
    int device = 0;
    int dtype = PC_DTYPE_F32;
    int64_t N = 0;
    int n_nominal = 12, R = 12, D = 18, n_tracks = 0;
    int lanes_override = 0;
    int lg = 0, rpl = 1, blocks = 0;
    std::vector<TrackHdr> hdr_host;
    // device buffers
    double4* pv = nullptr;
    int4* iv = nullptr;
    double* rot = nullptr;
    uint8_t* track_id = nullptr;
    bool track_blocks32 = false;   // mixed tracks: every aligned block of 32 envs holds ONE track (what pc_rollout needs)
    int track_block = 0;           // ... the largest of 256 / 128 / 64 / 32 for which that holds (0: none)
    TrackHdr* hdr = nullptr;
    Seg* segs = nullptr;
    Vtx* vtx = nullptr;
    double2* headtab = nullptr;
    float2* dirtab = nullptr;
    float* rden = nullptr;
    float* reset_obs = nullptr;

    template <typename T> EnvParams<T> params() const {
        EnvParams<T> p;
        p.N = N;
        p.lg = lg;
        p.n_nominal = n_nominal;
        p.q = n_nominal / 4;
        p.step_deg = 360 / n_nominal;
        p.R = R;
        p.D = D;
        p.colbits = 0;
        for (int r = 0; r < 64 && r < n_nominal; r += n_nominal / 4) p.colbits |= 1ull << r;
        p.pv = pv;
        p.iv = iv;
        p.rot = rot;
        p.track_id = track_id;
        p.hdr = hdr;
        p.segs = segs;
        p.vtx = vtx;
        p.headtab = headtab;
        p.dirtab = dirtab;
        p.rden = rden;
        p.reset_obs = reset_obs;
        return p;
    }

    // lanes per env: fill ~4 waves per SIMD (256 CUs x 4 SIMDs x 64 lanes x 4) but never more
    // lanes than rays, and keep rays-per-lane inside the instantiated menu.
    int choose_geometry() {
        int G;
        if (lanes_override > 0) {
            G = lanes_override;
        } else {
            const int64_t target = 262144;
            G = 1;
            while (G < 64 && (int64_t)G * N < target && G < R) G <<= 1;
        }
        const int max_rpl = dtype == PC_DTYPE_F64 ? 17 : 33;  // F64 keeps 6 VGPRs per ray slot
        while (true) {
            const int need = (R + G - 1) / G;
            const int m = pick_rpl(need);
            if (m > 0 && m <= max_rpl) {
                rpl = m;
                break;
            }
            if (G >= 64) return PC_ERR_UNSUPPORTED;
            G <<= 1;
        }
        lg = 0;
        while ((1 << lg) < G) ++lg;
        const int64_t lanes = N << lg;
        blocks = (int)((lanes + 255) / 256);
        return PC_OK;
    }
};

template <typename T, int RPL>
static void launch_step(const pc_env* e, const int64_t* actions, double reward_scale, float* obs, float* reward, float* term,
                        float* trunc, int32_t* gates_passed, float* final_obs, hipStream_t st) {
    if (e->track_id)
        hipLaunchKernelGGL((env_step_kernel<T, RPL, true>), dim3(e->blocks), dim3(256), 0, st, e->params<T>(), actions,
                           reward_scale, obs, reward, term, trunc, gates_passed, final_obs);
    else
        hipLaunchKernelGGL((env_step_kernel<T, RPL, false>), dim3(e->blocks), dim3(256), 0, st, e->params<T>(), actions,
                           reward_scale, obs, reward, term, trunc, gates_passed, final_obs);
}

extern "C" {

const char* pc_strerror(int code) {
    switch (code) {
        case PC_OK: return "ok";
        case PC_ERR_INVALID_ARG: return "invalid argument";
        case PC_ERR_IO: return "track file not found or unreadable";
        case PC_ERR_PARSE: return "track JSON malformed or schema violated";
        case PC_ERR_HIP: return "HIP runtime error";
        case PC_ERR_UNSUPPORTED: return "unsupported configuration";
        case PC_ERR_NO_DEVICE: return "no usable gfx950 device";
        default: return "unknown error";
    }
}

const char* pc_last_hip_error(void) { return g_hip_err.c_str(); }

int pc_ray_count(int n) {
    if (n < 4 || n > 360) return PC_ERR_INVALID_ARG;
    const int step = 360 / n;
    return (360 + step - 1) / step;  // len(range(0, 360, 360 // n)), car_env.py:269
}

int pc_track_load_json(const char* path, pc_track** out) {
    if (!path || !out) return PC_ERR_INVALID_ARG;
    std::unique_ptr<pc_track> t(new (std::nothrow) pc_track);
    if (!t) return PC_ERR_INVALID_ARG;
    const int rc = pc_internal_parse_track(path, t.get());
    if (rc != PC_OK) return rc;
    *out = t.release();
    return PC_OK;
}

int pc_track_from_arrays(const double* walls, int n_walls, const double* gates, int n_gates, double start_x, double start_y,
                         double start_angle_deg, pc_track** out) {
    if (!walls || !gates || n_walls < 1 || n_gates < 1 || !out) return PC_ERR_INVALID_ARG;
    pc_track* t = new (std::nothrow) pc_track;
    if (!t) return PC_ERR_INVALID_ARG;
    t->walls.assign(walls, walls + 4 * (size_t)n_walls);
    t->gates.assign(gates, gates + 4 * (size_t)n_gates);
    t->start_x = start_x;
    t->start_y = start_y;
    t->start_rot = start_angle_deg;
    *out = t;
    return PC_OK;
}

int pc_track_info(const pc_track* t, int* n_walls, int* n_gates, double* start) {
    if (!t) return PC_ERR_INVALID_ARG;
    if (n_walls) *n_walls = t->n_walls();
    if (n_gates) *n_gates = t->n_gates();
    if (start) {
        start[0] = t->start_x;
        start[1] = t->start_y;
        start[2] = t->start_rot;
    }
    return PC_OK;
}

int pc_track_geometry(const pc_track* t, double* walls, double* gates) {
    if (!t) return PC_ERR_INVALID_ARG;
    if (walls) memcpy(walls, t->walls.data(), t->walls.size() * sizeof(double));
    if (gates) memcpy(gates, t->gates.data(), t->gates.size() * sizeof(double));
    return PC_OK;
}

void pc_track_destroy(pc_track* t) { delete t; }

void pc_env_destroy(pc_env* e) {
    if (!e) return;
    DeviceGuard g(e->device);
    (void)hipFree(e->pv);
    (void)hipFree(e->iv);
    (void)hipFree(e->rot);
    (void)hipFree(e->track_id);
    (void)hipFree(e->hdr);
    (void)hipFree(e->segs);
    (void)hipFree(e->vtx);
    (void)hipFree(e->headtab);
    (void)hipFree(e->dirtab);
    (void)hipFree(e->rden);
    (void)hipFree(e->reset_obs);
    delete e;
}

static int env_create_impl(pc_env* e, const pc_track* const* tracks, const uint8_t* track_id) {
    const bool f64 = e->dtype == PC_DTYPE_F64;
    // ---- host images of the track table
    std::vector<Seg> segs;
    std::vector<Vtx> vtx;
    std::vector<double2> headtab;
    std::vector<float2> dirtab;
    size_t rden_floats = 0;
    e->hdr_host.resize(e->n_tracks);
    for (int k = 0; k < e->n_tracks; ++k) {
        const pc_track* t = tracks[k];
        TrackHdr& h = e->hdr_host[k];
        h.S = t->n_walls();
        h.G = t->n_gates();
        h.wall_off = (int)segs.size();
        for (size_t i = 0; i < t->walls.size(); i += 4) segs.push_back(Seg{t->walls[i], t->walls[i + 1], t->walls[i + 2], t->walls[i + 3]});
        h.gate_off = (int)segs.size();
        for (size_t i = 0; i < t->gates.size(); i += 4) segs.push_back(Seg{t->gates[i], t->gates[i + 1], t->gates[i + 2], t->gates[i + 3]});
        // walls as vertex chains: a segment continues the chain iff it starts exactly where the previous ended
        h.vtx_off = (int)vtx.size();
        for (int w = 0; w < h.S; ++w) {
            const Seg& sg = segs[h.wall_off + w];
            const bool cont = w > 0 && segs[h.wall_off + w - 1].x2 == sg.x1 && segs[h.wall_off + w - 1].y2 == sg.y1;
            if (!cont) vtx.push_back(Vtx{sg.x1, sg.y1, 0.f, 0.f, 1, 0});
            vtx.push_back(Vtx{sg.x2, sg.y2, (float)(sg.x1 - sg.x2), (float)(sg.y1 - sg.y2), 0, 0});
        }
        h.n_chain = (int)vtx.size() - h.vtx_off;
        h.pad_ = 0;
        while ((vtx.size() - h.vtx_off) % 4)  // the sweep walks vertex groups of four: pad with chain-break sentinels
            vtx.push_back(Vtx{vtx.back().x, vtx.back().y, 0.f, 0.f, 1, 0});
        h.nV = (int)vtx.size() - h.vtx_off;
        h.dir_off = (int)dirtab.size();
        for (int j = 0; j < 360; ++j) {  // direction lattice: start_rot + j degrees, np.radians then libm cos/sin
            const double a = (t->start_rot + (double)j) * (PC_PI / 180.0);
            dirtab.push_back(make_float2((float)std::cos(a), (float)std::sin(a)));
        }
        dirtab.push_back(make_float2(0.f, 0.f));
        h.rden_off = (int)rden_floats;
        rden_floats += (size_t)361 * h.nV;
        h.head_off = (int)headtab.size();
        h.start_collides = 0;
        h.start_x = t->start_x;
        h.start_y = t->start_y;
        h.start_rot = t->start_rot;
        for (int j = 0; j < 72; ++j) {  // heading grid: start_rot + 5 j degrees, np.radians then libm cos/sin
            const double a = (t->start_rot + 5.0 * j) * (PC_PI / 180.0);
            headtab.push_back(make_double2(std::cos(a), std::sin(a)));
        }
    }
    // ---- device buffers
    const size_t N = (size_t)e->N;
    HIPCHK(hipMalloc((void**)&e->pv, N * sizeof(double4)));
    HIPCHK(hipMalloc((void**)&e->iv, N * sizeof(int4)));
    if (f64) HIPCHK(hipMalloc((void**)&e->rot, N * sizeof(double)));
    if (track_id) {
        for (int blk = 256; blk >= 32 && !e->track_block; blk >>= 1) {
            bool ok = true;
            for (size_t i = 0; i < N && ok; ++i) ok = track_id[i] == track_id[i & ~(size_t)(blk - 1)];
            if (ok) e->track_block = blk;
        }
        e->track_blocks32 = e->track_block >= 32;
        HIPCHK(hipMalloc((void**)&e->track_id, N));
        HIPCHK(hipMemcpy(e->track_id, track_id, N, hipMemcpyHostToDevice));
    }
    HIPCHK(hipMalloc((void**)&e->hdr, e->n_tracks * sizeof(TrackHdr)));
    HIPCHK(hipMemcpy(e->hdr, e->hdr_host.data(), e->n_tracks * sizeof(TrackHdr), hipMemcpyHostToDevice));
    static_assert(sizeof(Seg) == 32 && sizeof(Vtx) == 32, "segment / vertex records are 32 bytes (one s_load_dwordx8)");
    HIPCHK(hipMalloc((void**)&e->segs, segs.size() * sizeof(Seg)));
    HIPCHK(hipMemcpy(e->segs, segs.data(), segs.size() * sizeof(Seg), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void**)&e->vtx, vtx.size() * sizeof(Vtx)));
    HIPCHK(hipMemcpy(e->vtx, vtx.data(), vtx.size() * sizeof(Vtx), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void**)&e->headtab, headtab.size() * sizeof(double2)));
    HIPCHK(hipMemcpy(e->headtab, headtab.data(), headtab.size() * sizeof(double2), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void**)&e->dirtab, dirtab.size() * sizeof(float2)));
    HIPCHK(hipMemcpy(e->dirtab, dirtab.data(), dirtab.size() * sizeof(float2), hipMemcpyHostToDevice));
    if (!f64) {
        HIPCHK(hipMalloc((void**)&e->rden, rden_floats * sizeof(float)));
        hipLaunchKernelGGL(rden_build_kernel, dim3(64), dim3(256), 0, 0, e->params<float>(), e->n_tracks, e->rden);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipMalloc((void**)&e->reset_obs, (size_t)e->n_tracks * e->D * sizeof(float)));
    // ---- per-track reset observation + start_collides, computed on the device by the same arithmetic
    int* d_sc = nullptr;
    HIPCHK(hipMalloc((void**)&d_sc, e->n_tracks * sizeof(int)));
    const int rb = (e->n_tracks + 63) / 64;
    if (f64)
        hipLaunchKernelGGL(reset_obs_kernel<double>, dim3(rb), dim3(64), 0, 0, e->params<double>(), e->n_tracks, e->reset_obs, d_sc);
    else
        hipLaunchKernelGGL(reset_obs_kernel<float>, dim3(rb), dim3(64), 0, 0, e->params<float>(), e->n_tracks, e->reset_obs, d_sc);
    HIPCHK(hipGetLastError());
    std::vector<int> sc(e->n_tracks);
    HIPCHK(hipMemcpy(sc.data(), d_sc, e->n_tracks * sizeof(int), hipMemcpyDeviceToHost));
    (void)hipFree(d_sc);
    for (int k = 0; k < e->n_tracks; ++k) e->hdr_host[k].start_collides = sc[k];
    HIPCHK(hipMemcpy(e->hdr, e->hdr_host.data(), e->n_tracks * sizeof(TrackHdr), hipMemcpyHostToDevice));
    return PC_OK;
}

int pc_env_create(int device, int64_t n_envs, int num_rays_nominal, const pc_track* const* tracks, int n_tracks,
                  const uint8_t* track_id, int dtype, pc_env** out) {
    if (!out || !tracks || n_tracks < 1 || n_tracks > 256 || n_envs < 1 || (dtype != PC_DTYPE_F32 && dtype != PC_DTYPE_F64))
        return PC_ERR_INVALID_ARG;
    if (num_rays_nominal < 4 || num_rays_nominal > 360) return PC_ERR_INVALID_ARG;
    for (int k = 0; k < n_tracks; ++k)
        if (!tracks[k] || tracks[k]->n_walls() < 1 || tracks[k]->n_gates() < 1) return PC_ERR_INVALID_ARG;
    if (track_id)
        for (int64_t i = 0; i < n_envs; ++i)
            if (track_id[i] >= n_tracks) return PC_ERR_INVALID_ARG;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1 || device < 0 || device >= count) return PC_ERR_NO_DEVICE;
    DeviceGuard guard(device);
    if (!guard.ok) return PC_ERR_NO_DEVICE;
    pc_env* e = new (std::nothrow) pc_env;
    if (!e) return PC_ERR_INVALID_ARG;
    e->device = device;
    e->dtype = dtype;
    e->N = n_envs;
    e->n_nominal = num_rays_nominal;
    e->R = pc_ray_count(num_rays_nominal);
    e->D = 6 + e->R;
    e->n_tracks = n_tracks;
    int rc = e->choose_geometry();
    if (rc == PC_OK) rc = env_create_impl(e, tracks, track_id);
    if (rc != PC_OK) {
        pc_env_destroy(e);
        return rc;
    }
    *out = e;
    return PC_OK;
}

int pc_env_obs_dim(const pc_env* e) { return e ? e->D : PC_ERR_INVALID_ARG; }
int pc_env_num_actions(const pc_env* e) { return e ? 9 : PC_ERR_INVALID_ARG; }  // spaces.Discrete(9), car_env.py:525
int64_t pc_env_num_envs(const pc_env* e) { return e ? e->N : PC_ERR_INVALID_ARG; }

int pc_env_set_lanes_per_env(pc_env* e, int lanes) {
    if (!e || lanes < 0 || lanes > 64 || (lanes & (lanes - 1))) return PC_ERR_INVALID_ARG;
    const int prev = e->lanes_override;
    e->lanes_override = lanes;
    const int rc = e->choose_geometry();
    if (rc != PC_OK) {
        e->lanes_override = prev;
        (void)e->choose_geometry();
    }
    return rc;
}

int pc_env_launch_info(const pc_env* e, int* lanes_per_env, int* rays_per_lane, int* blocks, int* threads) {
    if (!e) return PC_ERR_INVALID_ARG;
    if (lanes_per_env) *lanes_per_env = 1 << e->lg;
    if (rays_per_lane) *rays_per_lane = e->rpl;
    if (blocks) *blocks = e->blocks;
    if (threads) *threads = 256;
    return PC_OK;
}

int pc_env_reset(pc_env* e, float* obs, void* stream) {
    if (!e) return PC_ERR_INVALID_ARG;
    DeviceGuard guard(e->device);
    if (!guard.ok) return PC_ERR_NO_DEVICE;
    const int blocks = (int)((e->N + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    if (e->dtype == PC_DTYPE_F64)
        hipLaunchKernelGGL(env_reset_kernel<double>, dim3(blocks), dim3(256), 0, st, e->params<double>(), obs);
    else
        hipLaunchKernelGGL(env_reset_kernel<float>, dim3(blocks), dim3(256), 0, st, e->params<float>(), obs);
    HIPCHK(hipGetLastError());
    return PC_OK;
}

int pc_env_step(pc_env* e, const int64_t* actions, double reward_scale, float* obs, float* reward, float* terminated,
                float* truncated, int32_t* gates_passed, float* final_obs, void* stream) {
    if (!e || !actions || !obs || !reward || !terminated || !truncated) return PC_ERR_INVALID_ARG;
    DeviceGuard guard(e->device);
    if (!guard.ok) return PC_ERR_NO_DEVICE;
    hipStream_t st = (hipStream_t)stream;
#define PC_CASE(T, M)                                                                                        \
    case M:                                                                                                  \
        launch_step<T, M>(e, actions, reward_scale, obs, reward, terminated, truncated, gates_passed, final_obs, st); \
        break;
    if (e->dtype == PC_DTYPE_F64) {
        switch (e->rpl) {
            PC_CASE(double, 1) PC_CASE(double, 2) PC_CASE(double, 3) PC_CASE(double, 5) PC_CASE(double, 6)
            PC_CASE(double, 9) PC_CASE(double, 12) PC_CASE(double, 17)
            default: return PC_ERR_UNSUPPORTED;
        }
    } else {
        switch (e->rpl) {
            PC_CASE(float, 1) PC_CASE(float, 2) PC_CASE(float, 3) PC_CASE(float, 5) PC_CASE(float, 6)
            PC_CASE(float, 9) PC_CASE(float, 12) PC_CASE(float, 17) PC_CASE(float, 33)
            default: return PC_ERR_UNSUPPORTED;
        }
    }
#undef PC_CASE
    HIPCHK(hipGetLastError());
    return PC_OK;
}

int pc_env_info(pc_env* e, int32_t* gates_passed, int32_t* time_passed, void* stream) {
    if (!e) return PC_ERR_INVALID_ARG;
    DeviceGuard guard(e->device);
    if (!guard.ok) return PC_ERR_NO_DEVICE;
    hipLaunchKernelGGL(env_info_kernel, dim3((unsigned)((e->N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, e->iv, e->N, gates_passed,
                       time_passed);
    HIPCHK(hipGetLastError());
    return PC_OK;
}

int pc_build_ablate(void) { return PC_ABLATE; }

#ifdef PC_STAMPS
// developer build only (not declared in ppocar.h): copy the phase stamps of the last pc_rollout launch to the host
int pc_debug_read_stamps(unsigned long long* out, int n) {
    const int total = 8 * STAMP_NT * STAMP_NPH;
    if (n < total) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), total * sizeof(unsigned long long)) != hipSuccess) return -3;
    return total;
}
int pc_debug_read_stamps_u(unsigned long long* out) {
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps_u), 16 * sizeof(unsigned long long)) != hipSuccess) return -3;
    return 16;
}
#endif

int pc_env_get_state(pc_env* e, double* px, double* py, double* vx, double* vy, double* rot, int64_t* time_step,
                     int64_t* next_gate, int64_t* passed) {
    if (!e) return PC_ERR_INVALID_ARG;
    DeviceGuard guard(e->device);
    if (!guard.ok) return PC_ERR_NO_DEVICE;
    const size_t N = (size_t)e->N;
    const bool f64 = e->dtype == PC_DTYPE_F64;
    HIPCHK(hipDeviceSynchronize());
    std::vector<int4> iv(N);
    HIPCHK(hipMemcpy(iv.data(), e->iv, N * sizeof(int4), hipMemcpyDeviceToHost));
    std::vector<double> pv(4 * N);
    HIPCHK(hipMemcpy(pv.data(), e->pv, 4 * N * sizeof(double), hipMemcpyDeviceToHost));
    std::vector<double> r(N);
    std::vector<uint8_t> tid(N, 0);
    if (f64) HIPCHK(hipMemcpy(r.data(), e->rot, N * sizeof(double), hipMemcpyDeviceToHost));
    if (e->track_id) HIPCHK(hipMemcpy(tid.data(), e->track_id, N, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < N; ++i) {
        if (px) px[i] = pv[4 * i];
        if (py) py[i] = pv[4 * i + 1];
        if (vx) vx[i] = pv[4 * i + 2];
        if (vy) vy[i] = pv[4 * i + 3];
        if (rot) rot[i] = f64 ? r[i] : e->hdr_host[tid[i]].start_rot + 5.0 * iv[i].x;
        if (time_step) time_step[i] = iv[i].y;
        if (next_gate) next_gate[i] = iv[i].z;
        if (passed) passed[i] = iv[i].w;
    }
    return PC_OK;
}

int pc_env_set_state(pc_env* e, const double* px, const double* py, const double* vx, const double* vy, const double* rot,
                     const int64_t* time_step, const int64_t* next_gate, const int64_t* passed) {
    if (!e) return PC_ERR_INVALID_ARG;
    DeviceGuard guard(e->device);
    if (!guard.ok) return PC_ERR_NO_DEVICE;
    const size_t N = (size_t)e->N;
    const bool f64 = e->dtype == PC_DTYPE_F64;
    HIPCHK(hipDeviceSynchronize());
    std::vector<int4> iv(N);
    HIPCHK(hipMemcpy(iv.data(), e->iv, N * sizeof(int4), hipMemcpyDeviceToHost));
    std::vector<double> pv(4 * N);
    HIPCHK(hipMemcpy(pv.data(), e->pv, 4 * N * sizeof(double), hipMemcpyDeviceToHost));
    std::vector<uint8_t> tid(N, 0);
    if (e->track_id) HIPCHK(hipMemcpy(tid.data(), e->track_id, N, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < N; ++i) {
        if (px) pv[4 * i] = px[i];
        if (py) pv[4 * i + 1] = py[i];
        if (vx) pv[4 * i + 2] = vx[i];
        if (vy) pv[4 * i + 3] = vy[i];
        if (rot && !f64) iv[i].x = (int)std::llround((rot[i] - e->hdr_host[tid[i]].start_rot) / 5.0);
        if (time_step) iv[i].y = (int)time_step[i];
        if (next_gate) {
            if (next_gate[i] < 0 || next_gate[i] >= e->hdr_host[tid[i]].G) return PC_ERR_INVALID_ARG;
            iv[i].z = (int)next_gate[i];
        }
        if (passed) iv[i].w = (int)passed[i];
    }
    HIPCHK(hipMemcpy(e->iv, iv.data(), N * sizeof(int4), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->pv, pv.data(), 4 * N * sizeof(double), hipMemcpyHostToDevice));
    if (f64 && rot) HIPCHK(hipMemcpy(e->rot, rot, N * sizeof(double), hipMemcpyHostToDevice));
    return PC_OK;
}

int pc_gae(int device, const float* rew, const float* val, const float* term, const float* trunc, const float* last_val,
           const float* last_term, const float* last_trunc, double gamma, double lam, int64_t T, int64_t N, float* adv,
           float* ret, void* stream) {
    if (!rew || !val || !term || !trunc || !last_val || !last_term || !last_trunc || !adv || !ret || T < 1 || N < 1)
        return PC_ERR_INVALID_ARG;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1 || device < 0 || device >= count) return PC_ERR_NO_DEVICE;
    DeviceGuard guard(device);
    if (!guard.ok) return PC_ERR_NO_DEVICE;
    const int blocks = (int)((N + 255) / 256);
    // gamma and gamma*lambda are Python floats that torch casts to float32 at the multiply
    hipLaunchKernelGGL(gae_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, rew, val, term, trunc, last_val, last_term,
                       last_trunc, (float)gamma, (float)(gamma * lam), T, N, adv, ret);
    HIPCHK(hipGetLastError());
    return PC_OK;
}

int pc_sample(int device, const float* logits, int64_t N, int A, uint64_t seed, uint64_t offset, int64_t* actions,
              float* logprob, float* entropy, void* stream) {
    if (!logits || !actions || !logprob || N < 1 || A < 1) return PC_ERR_INVALID_ARG;
    if (A > 16) return PC_ERR_UNSUPPORTED;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1 || device < 0 || device >= count) return PC_ERR_NO_DEVICE;
    DeviceGuard guard(device);
    if (!guard.ok) return PC_ERR_NO_DEVICE;
    const int blocks = (int)((N + 255) / 256);
    hipLaunchKernelGGL(sample_kernel<16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, logits, N, A, seed, offset, actions,
                       logprob, entropy);
    HIPCHK(hipGetLastError());
    return PC_OK;
}

static int policy_ks(int D) { return D <= 20 ? 5 : (D <= 24 ? 6 : 10); }
static int g_policy_split_mode = -1;  // -1 auto (split below 32768 envs), 0 never, 1 always
static int g_rollout_form = -1;       // pc_rollout: -1 auto, 0 = 256 envs per workgroup, 1 = 32 envs per workgroup
// Batches up to this size take the forms that cut the work of 32 envs over a whole workgroup (policy_kernel<SPLIT>,
// rollout_small_kernel): n_envs / 32 workgroups, so 16384 envs are two rounds of 256 -- about what the 128-env big form
// needs for anything up to 32768 envs.  The same bound for both kernels keeps the default per-step and persistent paths
// bit-identical.
#define PC_SPLIT_MAX_ENVS 16384
static int64_t g_rollout_epw128_max = 32768;  // big form at or below this many envs: 128 envs (4 waves) per workgroup
static int g_rollout_epw_override = 0;  // pc_rollout_set_epw: 0 = automatic, 128 / 256 = force (test knob)
static int g_rollout_fast = 1;        // pc_rollout: the big form's fast mode (LDS tables behind LDS pointers) when the shape allows it (0: never; A/B knob)
static int g_rollout_rden = 1;        // pc_rollout: stage the 1/den table in LDS when it fits (0: never; test / tuning knob)
static int g_policy_precision = 2;    // 0 = fp32-input MFMA; split forms on the 16-bit matrix cores (need D <= 24, A <= 9): 1 = bf16 x 3, 2 = fp16 x 2

int pc_rollout_set_form(int form) {
    if (form < -1 || form > 3) return PC_ERR_INVALID_ARG;
    g_rollout_rden = form >= 2 ? 0 : 1;              // forms 2 / 3 = forms 0 / 1 without the LDS 1/den table
    g_rollout_form = form >= 2 ? form - 2 : form;
    return PC_OK;
}

int pc_rollout_set_fast(int on) {
    if (on != 0 && on != 1) return PC_ERR_INVALID_ARG;
    g_rollout_fast = on;
    return PC_OK;
}

int pc_rollout_set_epw(int envs_per_workgroup) {
    if (envs_per_workgroup != 0 && envs_per_workgroup != 16 && envs_per_workgroup != 32 && envs_per_workgroup != 128 &&
        envs_per_workgroup != 256)
        return PC_ERR_INVALID_ARG;
    g_rollout_epw_override = envs_per_workgroup;
    return PC_OK;
}

int pc_policy_set_precision(int mode) {
    if (mode < 0 || mode > 2) return PC_ERR_INVALID_ARG;
    g_policy_precision = mode;
    return PC_OK;
}

static int policy_prec(int D, int A) { return (g_policy_precision >= 1 && D <= 40 && A <= 9) ? g_policy_precision : 0; }
int pc_policy_precision(int D, int H, int A) {
    if (H != 256 || A < 1 || A > 15 || D < 1 || D > 40) return PC_ERR_UNSUPPORTED;
    return policy_prec(D, A);
}

int pc_policy_set_split(int mode) {
    if (mode < -1 || mode > 1) return PC_ERR_INVALID_ARG;
    g_policy_split_mode = mode;
    return PC_OK;
}

int64_t pc_policy_image_floats(int D, int H, int A) {
    if (H != 256 || A < 1 || A > 15 || D < 1 || D > 40) return PC_ERR_UNSUPPORTED;
    const int prec = policy_prec(D, A);
    return prec ? polx_image_dwords(prec, pol_ng(policy_ks(D))) : pol_image_padded(policy_ks(D));
}

int pc_policy_pack(int device, int D, int H, int A, const float* aW1, const float* ab1, const float* aW2, const float* ab2,
                   const float* cW1, const float* cb1, const float* cW2, const float* cb2, float* image, void* stream) {
    if (!aW1 || !ab1 || !aW2 || !ab2 || !cW1 || !cb1 || !cW2 || !cb2 || !image) return PC_ERR_INVALID_ARG;
    if (H != 256 || A < 1 || A > 15 || D < 1 || D > 40) return PC_ERR_UNSUPPORTED;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1 || device < 0 || device >= count) return PC_ERR_NO_DEVICE;
    DeviceGuard guard(device);
    if (!guard.ok) return PC_ERR_NO_DEVICE;
    const int prec = policy_prec(D, A);
#define PC_PACK(PRC, NGV)                                                                                                \
    hipLaunchKernelGGL((policy_pack16_kernel<PRC, NGV>), dim3(64), dim3(256), 0, (hipStream_t)stream, D, A, aW1, ab1, aW2, ab2, cW1, \
                       cb1, cW2, cb2, reinterpret_cast<unsigned*>(image))
    const int ng = pol_ng(policy_ks(D));
    if (prec == 1) { if (ng == 5) PC_PACK(1, 5); else PC_PACK(1, 3); }
    else if (prec == 2) { if (ng == 5) PC_PACK(2, 5); else PC_PACK(2, 3); }
#undef PC_PACK
    else
        hipLaunchKernelGGL(policy_pack_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, policy_ks(D), D, A, aW1, ab1, aW2, ab2,
                           cW1, cb1, cW2, cb2, image);
    HIPCHK(hipGetLastError());
    return PC_OK;
}

int pc_policy_act(int device, const float* obs, int64_t N, int D, int H, int A, const float* image, uint64_t seed,
                  uint64_t offset, const uint64_t* offset_dev, int64_t* action, float* action_f32, float* logprob, float* value,
                  float* logits_out, void* stream) {
    if (!obs || !image || !action || !logprob || !value || N < 1) return PC_ERR_INVALID_ARG;
    if (H != 256 || A < 1 || A > 15 || D < 1 || D > 40) return PC_ERR_UNSUPPORTED;  // the caller falls back to its own GEMMs
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1 || device < 0 || device >= count) return PC_ERR_NO_DEVICE;
    DeviceGuard guard(device);
    if (!guard.ok) return PC_ERR_NO_DEVICE;
    const int KS = policy_ks(D);
    const int prec = policy_prec(D, A);
    const size_t lds = (size_t)((prec ? polx_image_dwords(prec, pol_ng(KS)) : pol_image_padded(KS)) + 8 * 32 * 17) * sizeof(float);
    static int n_cu[64] = {0};
    if (device < 64 && n_cu[device] == 0) {
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, device));
        n_cu[device] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int cus = device < 64 ? n_cu[device] : 256;
    // too few 256-env workgroups to fill the chip: split the hidden tiles over the waves instead
    const bool split = g_policy_split_mode < 0 ? N <= PC_SPLIT_MAX_ENVS : g_policy_split_mode == 1;
    const int64_t chunks = split ? (N + 31) / 32 : (N + 255) / 256;
    const int blocks = (int)(chunks < cus ? chunks : cus);  // one ~100-KB-LDS workgroup per CU, persistent over env chunks
    hipStream_t st = (hipStream_t)stream;
#define PC_POL(KSV, SPL, PRC)                                                                                            \
    do {                                                                                                                 \
        static bool attr_set[64] = {false};                                                                              \
        if (device < 64 && !attr_set[device]) {                                                                          \
            HIPCHK(hipFuncSetAttribute((const void*)policy_kernel<KSV, SPL, PRC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
            attr_set[device] = true;                                                                                     \
        }                                                                                                                \
        hipLaunchKernelGGL((policy_kernel<KSV, SPL, PRC>), dim3(blocks), dim3(512), lds, st, obs, N, D, A, image, seed, offset, offset_dev, \
                           action, action_f32, logprob, value, logits_out);                                              \
    } while (0)
    if (prec == 1) {
        if (split) { if (KS == 5) PC_POL(5, true, 1); else if (KS == 6) PC_POL(6, true, 1); else PC_POL(10, true, 1); }
        else { if (KS == 5) PC_POL(5, false, 1); else if (KS == 6) PC_POL(6, false, 1); else PC_POL(10, false, 1); }
    } else if (prec == 2) {
        if (split) { if (KS == 5) PC_POL(5, true, 2); else if (KS == 6) PC_POL(6, true, 2); else PC_POL(10, true, 2); }
        else { if (KS == 5) PC_POL(5, false, 2); else if (KS == 6) PC_POL(6, false, 2); else PC_POL(10, false, 2); }
    } else if (split) {
        if (KS == 5) PC_POL(5, true, 0);
        else if (KS == 6) PC_POL(6, true, 0);
        else PC_POL(10, true, 0);
    } else {
        if (KS == 5) PC_POL(5, false, 0);
        else if (KS == 6) PC_POL(6, false, 0);
        else PC_POL(10, false, 0);
    }
#undef PC_POL
    HIPCHK(hipGetLastError());
    return PC_OK;
}

int pc_ppo_gather(int device, const int64_t* idx, int B, int D, const float* obs, const float* act, const float* logprob,
                  const float* adv, const float* ret, float* o_obs, float* o_act, float* o_logprob, float* o_adv, float* o_ret,
                  void* stream) {
    if (!idx || !obs || !act || !logprob || !adv || !ret || !o_obs || !o_act || !o_logprob || !o_adv || !o_ret || B < 1 || D < 1)
        return PC_ERR_INVALID_ARG;
    DeviceGuard guard(device);
    if (!guard.ok) return PC_ERR_NO_DEVICE;
    const int total = B * (D + 4);
    hipLaunchKernelGGL(ppo_gather_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, idx, B, D, obs, act,
                       logprob, adv, ret, o_obs, o_act, o_logprob, o_adv, o_ret);
    HIPCHK(hipGetLastError());
    return PC_OK;
}

int pc_ppo_loss(int device, const float* logits, const float* values, const float* act, const float* old_logprob,
                const float* adv, const float* ret, int B, int A, double clip_ratio, double vf_coef, double ent_coef,
                float* dlogits, float* dvalues, float* metrics, void* stream) {
    if (!logits || !values || !act || !old_logprob || !adv || !ret || !dlogits || !dvalues || !metrics) return PC_ERR_INVALID_ARG;
    if (B < 2 || B > 1024 || A < 1 || A > 16) return PC_ERR_UNSUPPORTED;
    DeviceGuard guard(device);
    if (!guard.ok) return PC_ERR_NO_DEVICE;
    const int threads = ((B + 63) / 64) * 64;
    hipLaunchKernelGGL(ppo_loss_kernel<16>, dim3(1), dim3(threads), 0, (hipStream_t)stream, logits, values, act, old_logprob, adv,
                       ret, B, A, (float)clip_ratio, (float)vf_coef, (float)ent_coef, dlogits, dvalues, metrics);
    HIPCHK(hipGetLastError());
    return PC_OK;
}

int pc_clip_adam(int device, float* param, float* grad, float* exp_avg, float* exp_avg_sq, float* step_count, const float* lr_dev,
                 int64_t n, double max_norm, double grad_scale, double beta1, double beta2, double eps, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !step_count || !lr_dev || n < 1 || n > (1 << 26)) return PC_ERR_INVALID_ARG;
    DeviceGuard guard(device);
    if (!guard.ok) return PC_ERR_NO_DEVICE;
    hipLaunchKernelGGL(clip_adam_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, step_count,
                       lr_dev, (int)n, (float)max_norm, (float)grad_scale, (float)beta1, (float)beta2, (float)eps);
    HIPCHK(hipGetLastError());
    return PC_OK;
}

int pc_clip_adam_advanced(int device, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const float* step_count,
                          const float* lr_dev, int64_t n, double max_norm, double grad_scale, double beta1, double beta2, double eps,
                          void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !step_count || !lr_dev || n < 1 || n > (1 << 26)) return PC_ERR_INVALID_ARG;
    DeviceGuard guard(device);
    if (!guard.ok) return PC_ERR_NO_DEVICE;
    hipLaunchKernelGGL(clip_adam_mb_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, param, const_cast<float*>(grad),
                       exp_avg, exp_avg_sq, step_count, lr_dev, (int)n, (float)max_norm, (float)grad_scale, (float)beta1, (float)beta2, (float)eps);
    HIPCHK(hipGetLastError());
    return PC_OK;
}

int pc_rollout(pc_env* e, const float* image, int A, int64_t T, double reward_scale, uint64_t seed, uint64_t offset,
               const uint64_t* offset_dev, float* obs_buf, float* act_buf, float* rew_buf, float* val_buf, float* term_buf,
               float* trunc_buf, float* logprob_buf, float* next_obs, float* next_term, float* next_trunc, void* stream) {
    if (!e || !image || !obs_buf || !act_buf || !rew_buf || !val_buf || !term_buf || !trunc_buf || !logprob_buf || !next_obs ||
        !next_term || !next_trunc || T < 1 || T > (1 << 24))
        return PC_ERR_INVALID_ARG;
    // mixed tracks: a wave (big form) / a workgroup (small form) steps 32 consecutive envs, which must share one track
    if (e->dtype != PC_DTYPE_F32 || (e->track_id && !e->track_blocks32) || A < 1 || A > 15) return PC_ERR_UNSUPPORTED;
    const int KS = policy_ks(e->D);
    DeviceGuard guard(e->device);
    if (!guard.ok) return PC_ERR_NO_DEVICE;
    const int prec = policy_prec(e->D, A);
    const int img = prec ? polx_image_dwords(prec, pol_ng(KS)) : pol_image_padded(KS);
    // large batches: 256 envs per workgroup, every wave independent; small batches: 32 envs per workgroup, hidden tiles and
    // wall-sweep parts split over the waves
    const bool small = g_rollout_form == 1 || (g_rollout_form < 0 && e->N <= PC_SPLIT_MAX_ENVS);
    const int epw = g_rollout_epw_override >= 128 ? g_rollout_epw_override
                                                  : ((!small && e->N <= g_rollout_epw128_max) ? 128 : 256);   // big form: envs per workgroup
    int max_G = 0, max_nV = 0;
    for (const TrackHdr& h : e->hdr_host) { max_G = std::max(max_G, h.G); max_nV = std::max(max_nV, h.nV); }
    // fast mode: Discrete(9), every gather table in LDS behind LDS pointers, dense observation rows (big form: each wave's output
    // tile aliases its own 32 observation rows -- dead between the policy pass's operand load and the env step's store of the
    // next observation: needs D >= 17).  A workgroup stages ONE track's tables: single-track batches, or mixed ones in which
    // every workgroup's block of envs lies on one track.
    const bool fast_shape = A == 9 && e->D >= 17 && e->D <= 40 && max_G <= TAB_MAX_GATES && g_rollout_fast;
    const bool fast = !small && fast_shape && (!e->track_id || e->track_block >= epw);
    const size_t lds_big = fast ? (size_t)(img + 256 * e->D + 256 + FT_FLOATS) * sizeof(float)
                                : (size_t)(img + 256 * (4 * KS + 1) + 256 + TAB_FLOATS) * sizeof(float);
    const bool fast_small = small && fast_shape && max_nV <= FT_VTX_MAX;     // (a small-form workgroup is 16 or 32 envs)
    const size_t lds_small = fast_small ? (size_t)(img + 8 * 32 * 17 + 32 * 40 + 32 + FT_FLOATS) * sizeof(float)
                                        : (size_t)(img + 8 * 32 * 17 + 32 * (4 * KS + 1) + 32 + TAB_FLOATS) * sizeof(float);
    size_t lds = small ? lds_small : lds_big;
    if (lds > 160 * 1024) return PC_ERR_UNSUPPORTED;
    // the track's 1/den table rides along in LDS when it fits (big_track: 361 x 28 floats = 40 KB); else the sweep forms
    // den and its reciprocal itself -- same bits either way.  Mixed batches: in the fast modes only (room for the largest track).
    // (the big form at 33 rays has 4 KB left: no closed track's table fits, so that shape is built without the table mode)
    int rden_lds = 361 * max_nV;
    if (g_rollout_rden == 0 || (e->track_id && !(small ? fast_small : fast)) || lds + (size_t)rden_lds * sizeof(float) > 160 * 1024 ||
        (!small && KS == 10))
        rden_lds = 0;
    lds += (size_t)rden_lds * sizeof(float);
    const int rpl = small ? (e->R + 3) / 4 : (e->R + 1) / 2;  // 4 (x 4 sweep parts) or 2 lanes per env
    // small form: 16 envs per workgroup up to 4096 envs (<= 256 workgroups: one per CU), else 32
    const int epw_small = (g_rollout_epw_override == 16 || g_rollout_epw_override == 32) ? g_rollout_epw_override
                          : ((fast_small && prec != 0 && e->R <= 17 && e->N <= 4096) ? 16 : 32);
    if (small && epw_small == 16 && !(fast_small && prec != 0 && e->R <= 17)) return PC_ERR_UNSUPPORTED;
    const int blocks = (int)(small ? (e->N + epw_small - 1) / epw_small : (e->N + epw - 1) / epw);
    const int vec_ok = ((e->N * e->D) % 4 == 0) ? 1 : 0;      // the waves' 32-row blocks are 16-byte aligned in the buffers
    const int mode = (fast || fast_small) ? (rden_lds ? 2 : 1) : 0;
    hipStream_t st = (hipStream_t)stream;
    EnvParams<float> prm = e->params<float>();
    prm.lg = small ? 2 : 1;
#define PC_ROLL_M(KSV, RPLV, PRC, MD)                                                                                    \
    do {                                                                                                                 \
        static bool attr_set[64] = {false};                                                                              \
        if (e->device < 64 && !attr_set[e->device]) {                                                                    \
            HIPCHK(hipFuncSetAttribute((const void*)rollout_kernel<KSV, RPLV, PRC, MD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
            attr_set[e->device] = true;                                                                                  \
        }                                                                                                                \
        hipLaunchKernelGGL((rollout_kernel<KSV, RPLV, PRC, MD>), dim3(blocks), dim3(512), lds, st, prm, image, A, (int)T, reward_scale, seed, \
                           offset, offset_dev, obs_buf, act_buf, rew_buf, val_buf, term_buf, trunc_buf, logprob_buf, next_obs,  \
                           next_term, next_trunc, rden_lds, epw, vec_ok);                                                \
    } while (0)
#define PC_ROLL(KSV, RPLV, PRC)                                                                                          \
    do {                                                                                                                 \
        if (mode == 2) PC_ROLL_M(KSV, RPLV, PRC, 2);                                                                     \
        else if (mode == 1) PC_ROLL_M(KSV, RPLV, PRC, 1);                                                                \
        else PC_ROLL_M(KSV, RPLV, PRC, 0);                                                                               \
    } while (0)
#define PC_ROLLS_M(KSV, RPLV, PRC, MD, EPWV)                                                                             \
    do {                                                                                                                 \
        static bool attr_set[64] = {false};                                                                              \
        if (e->device < 64 && !attr_set[e->device]) {                                                                    \
            HIPCHK(hipFuncSetAttribute((const void*)rollout_small_kernel<KSV, RPLV, PRC, MD, EPWV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
            attr_set[e->device] = true;                                                                                  \
        }                                                                                                                \
        hipLaunchKernelGGL((rollout_small_kernel<KSV, RPLV, PRC, MD, EPWV>), dim3(blocks), dim3(512), lds, st, prm, image, A, (int)T, reward_scale, \
                           seed, offset, offset_dev, obs_buf, act_buf, rew_buf, val_buf, term_buf, trunc_buf, logprob_buf, next_obs, \
                           next_term, next_trunc, rden_lds, vec_ok);                                                             \
    } while (0)
#define PC_ROLLS(KSV, RPLV, PRC)                                                                                         \
    do {                                                                                                                 \
        if constexpr (PRC != 0 && RPLV <= 5) {                                                                           \
            if (mode && epw_small == 16) { PC_ROLLS_M(KSV, RPLV, PRC, 1, 16); break; }                                   \
        }                                                                                                                \
        if (mode) PC_ROLLS_M(KSV, RPLV, PRC, 1, 32);    /* (the small form takes the 1/den table as a run-time branch) */   \
        else PC_ROLLS_M(KSV, RPLV, PRC, 0, 32);                                                                          \
    } while (0)
    if (small) {
        if (KS == 5 && rpl == 3) { if (prec == 2) PC_ROLLS(5, 3, 2); else if (prec) PC_ROLLS(5, 3, 1); else PC_ROLLS(5, 3, 0); }        // 12 rays
        else if (KS == 6 && rpl == 5) { if (prec == 2) PC_ROLLS(6, 5, 2); else if (prec) PC_ROLLS(6, 5, 1); else PC_ROLLS(6, 5, 0); }   // 16 -> 17 rays
        else if (KS == 10 && rpl == 9) { if (prec == 2) PC_ROLLS(10, 9, 2); else if (prec) PC_ROLLS(10, 9, 1); else PC_ROLLS(10, 9, 0); }   // 32 -> 33 rays
        else return PC_ERR_UNSUPPORTED;
    } else if (KS == 5 && rpl == 6) { if (prec == 2) PC_ROLL(5, 6, 2); else if (prec) PC_ROLL(5, 6, 1); else PC_ROLL(5, 6, 0); }       // 12 rays, D = 18
    else if (KS == 6 && rpl == 9) { if (prec == 2) PC_ROLL(6, 9, 2); else if (prec) PC_ROLL(6, 9, 1); else PC_ROLL(6, 9, 0); }          // 16 -> 17 rays, D = 23
    else if (KS == 10 && rpl == 17 && prec) {                                                                                             // 32 -> 33 rays, D = 39
        if (prec == 2) { if (mode) PC_ROLL_M(10, 17, 2, 1); else PC_ROLL_M(10, 17, 2, 0); }
        else { if (mode) PC_ROLL_M(10, 17, 1, 1); else PC_ROLL_M(10, 17, 1, 0); }
    }
    else return PC_ERR_UNSUPPORTED;
#undef PC_ROLL_M
#undef PC_ROLLS_M
#undef PC_ROLLS
#undef PC_ROLL
    HIPCHK(hipGetLastError());
    return PC_OK;
}

int64_t pc_ppo_workspace_floats(int B, int D, int H, int A) {
    if (H != 256 || A < 1 || A > 15 || D < 1 || D > 40 || B < 2 || B > 1024) return PC_ERR_UNSUPPORTED;
    const int64_t n_param = 2 * ((int64_t)H * D + H) + (int64_t)A * H + A + H + 1;
    const int64_t n_part = (B + FB_S - 1) / FB_S;
    return n_part * ((n_param + 3) & ~(int64_t)3) + n_part * 4 + (n_param + 255) / 256;
}

static int ppo_minibatch_impl(int device, const int64_t* idx, const float* prep, int B, int D, int H, int A, const float* obs,
                              const float* act, const float* old_logprob, const float* adv, const float* ret, float* param, float* grad,
                              float* exp_avg, float* exp_avg_sq, float* step_count, const float* lr_dev, double clip_ratio,
                              double vf_coef, double ent_coef, double max_norm, double beta1, double beta2, double eps, float* metrics,
                              float* workspace, int apply, void* stream) {
    if (!param || !grad || !metrics || !workspace) return PC_ERR_INVALID_ARG;
    if (!prep && (!idx || !obs || !act || !old_logprob || !adv || !ret)) return PC_ERR_INVALID_ARG;
    if (apply == 1 && (!exp_avg || !exp_avg_sq || !step_count || !lr_dev)) return PC_ERR_INVALID_ARG;
    if (apply == 2 && !step_count) return PC_ERR_INVALID_ARG;
    if (apply < 0 || apply > 2) return PC_ERR_INVALID_ARG;
    if (H != 256 || A < 1 || A > 15 || D < 1 || D > 40 || B < 2 || B > 1024) return PC_ERR_UNSUPPORTED;
    DeviceGuard guard(device);
    if (!guard.ok) return PC_ERR_NO_DEVICE;
    const int n_param = 2 * (H * D + H) + A * H + A + H + 1;
    const int n_part = (B + FB_S - 1) / FB_S;
    const int n_blk = (n_param + 255) / 256;
    const int n_pad = (n_param + 3) & ~3;          // a partial's row stride: 16-byte aligned rows
    const int HD = H * D, mid_end = HD + H + A * H + A;   // natural offset of critic.0.weight (ppo_fwdbwd_body's o_cW1)
    float* partial = workspace;
    float* metric_partial = partial + (size_t)n_part * n_pad;
    float* norm_partial = metric_partial + n_part * 4;
    hipStream_t st = (hipStream_t)stream;
    if (D <= 24)
        hipLaunchKernelGGL(ppo_fwdbwd_kernel<24>, dim3(n_part), dim3(256), 0, st, idx, B, D, A, obs, act, old_logprob, adv, ret, param,
                           (float)clip_ratio, (float)vf_coef, (float)ent_coef, partial, metric_partial, prep);
    else
        hipLaunchKernelGGL(ppo_fwdbwd_kernel<40>, dim3(n_part), dim3(256), 0, st, idx, B, D, A, obs, act, old_logprob, adv, ret, param,
                           (float)clip_ratio, (float)vf_coef, (float)ent_coef, partial, metric_partial, prep);
    hipLaunchKernelGGL(grad_reduce_kernel, dim3(n_blk), dim3(256), 0, st, partial, n_part, n_param, HD, mid_end, n_pad, grad, norm_partial, metric_partial, B,
                       (float)vf_coef, (float)ent_coef, metrics, apply ? step_count : nullptr);
    if (apply == 1)
        hipLaunchKernelGGL(adam_kernel, dim3(n_blk), dim3(256), 0, st, param, grad, exp_avg, exp_avg_sq, step_count, lr_dev, norm_partial,
                           n_blk, n_param, (float)max_norm, (float)beta1, (float)beta2, (float)eps);
    HIPCHK(hipGetLastError());
    return PC_OK;
}

int pc_ppo_minibatch(int device, const int64_t* idx, int B, int D, int H, int A, const float* obs, const float* act,
                     const float* old_logprob, const float* adv, const float* ret, float* param, float* grad, float* exp_avg,
                     float* exp_avg_sq, float* step_count, const float* lr_dev, double clip_ratio, double vf_coef, double ent_coef,
                     double max_norm, double beta1, double beta2, double eps, float* metrics, float* workspace, int apply,
                     void* stream) {
    return ppo_minibatch_impl(device, idx, nullptr, B, D, H, A, obs, act, old_logprob, adv, ret, param, grad, exp_avg, exp_avg_sq,
                              step_count, lr_dev, clip_ratio, vf_coef, ent_coef, max_norm, beta1, beta2, eps, metrics, workspace, apply,
                              stream);
}

int64_t pc_ppo_prepared_floats(int B, int D) {
    if (D < 1 || D > 40 || B < 2 || B > 1024) return PC_ERR_UNSUPPORTED;
    return (int64_t)B * (D + 4) + 4;
}

int pc_ppo_prepare(int device, const int64_t* idx, int64_t idx_ld, int n_mb, int B, int D, const float* obs, const float* act,
                   const float* old_logprob, const float* adv, const float* ret, float* prepared, void* stream) {
    if (!idx || !obs || !act || !old_logprob || !adv || !ret || !prepared || n_mb < 1 || idx_ld < B) return PC_ERR_INVALID_ARG;
    if (D < 1 || D > 40 || B < 2 || B > 1024) return PC_ERR_UNSUPPORTED;
    DeviceGuard guard(device);
    if (!guard.ok) return PC_ERR_NO_DEVICE;
    hipLaunchKernelGGL(ppo_prepare_kernel, dim3(n_mb), dim3(256), 0, (hipStream_t)stream, idx, idx_ld, B, D, obs, act, old_logprob, adv, ret,
                       prepared, (int64_t)B * (D + 4) + 4);
    HIPCHK(hipGetLastError());
    return PC_OK;
}

int pc_ppo_minibatch_prepared(int device, const float* prepared_mb, int B, int D, int H, int A, float* param, float* grad, float* exp_avg,
                              float* exp_avg_sq, float* step_count, const float* lr_dev, double clip_ratio, double vf_coef,
                              double ent_coef, double max_norm, double beta1, double beta2, double eps, float* metrics,
                              float* workspace, int apply, void* stream) {
    if (!prepared_mb) return PC_ERR_INVALID_ARG;
    return ppo_minibatch_impl(device, nullptr, prepared_mb, B, D, H, A, nullptr, nullptr, nullptr, nullptr, nullptr, param, grad, exp_avg,
                              exp_avg_sq, step_count, lr_dev, clip_ratio, vf_coef, ent_coef, max_norm, beta1, beta2, eps, metrics,
                              workspace, apply, stream);
}

}  // extern "C"
