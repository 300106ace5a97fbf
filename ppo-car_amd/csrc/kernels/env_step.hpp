// env_step.hpp -- part of the single translation unit ppocar.hip (included there, in order; not a stand-alone header).
// K1 env_step_kernel (the whole CarEnv.step transition, wall sweeps included) and K2 env_reset_kernel / reset_obs_kernel.
#pragma once

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
