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

// ------------------------------------------------------------------------------------------
// The float32 wall sweep = the SELECTOR of Car.get_distances (car_env.py:360-374): for the RPL ray slots of one lane, which
// wall segment is hit first.  The distances themselves are recomputed in float64 afterwards (refine_fast / refine_careful,
// env_math.hpp).
//
// Per ray slot and chain vertex k (segment (k-1, k)) the sweep forms a CANDIDATE and keeps the unsigned minimum of the
// candidates' bit patterns (for non-negative floats unsigned order is value order):
//   c_k   = cross(a_k, dir), a_k = p_k - pos formed in float64 then rounded: one per VERTEX, shared by the two segments that meet
//           there, so that a float32 ray cannot slip between two adjacent walls through their common corner;
//   P     = clamp01(c_{k-1} * c_k): exactly 0 iff the endpoints do not lie strictly on the same side of the ray line -- the
//           reference's 0 < t < 1 (car_env.py:178) up to the sign of a zero; one v_pk_mul_f32 with the clamp modifier per two slots;
//   cand  = fma(un', 1/den, P), un' = 2^-40 * cross(e, a_{k-1}) with e the wall's unit vector (the reference's u numerator, :176,
//           over the wall's length): u * 2^-40 for a crossed segment, >= 2^-40 ... 1 otherwise (any real distance is <=
//           1000 * 2^-40 ~ 1e-9, and two same-side values whose product is below 1e-9 have raised a flag, below), negative (sign
//           bit: above every non-negative pattern) for a hit behind the car (u < 0), NaN for a chain start (0 * inf); ONE v_fma_f32;
//   bits  = (bits(cand) & ~idx_mask) | k: the vertex index rides in the low mantissa bits (5 for <= 32 vertices: the selector
//           compares distances to 2^-19 relative; equal within that, the lower index wins); v_and_or_b32.
// Beside the minimum the sweep keeps, per PAIR of ray slots, the smallest |c_k| it saw and, per lane, the smallest |un'|: a ray
// that passes a vertex closer than the float32 rounding of c (tau_c), or a car closer to a wall's line than the rounding of un
// (tau_u), is FLAGGED -- its side tests / the sign of u cannot be trusted -- and gets selection 0.
// What comes out per slot: the index of the nearest crossed segment, or 0 = nothing certified (flagged, or nothing within
// 1000.5 px).  For an unflagged ray every side test and every sign of u is the exact one, so every wall the reference hits was
// a candidate and the selected one is a real hit; float32 can still ORDER two hits wrongly, but only if they lie within its
// rounding of each other, i.e. where two walls meet: the refinement re-examines the chain neighbours whenever the refined hit
// lies within 0.05 px of a segment end.  Selection 0 is resolved by a float64 scan of the whole chain under the reference's
// strict test.  What is assumed of a track: two walls come within ~0.01 px of each other only at a shared vertex, and the
// track fits 2000 px (the flag thresholds are priced from the car's distance to the track's bounding box).
// ------------------------------------------------------------------------------------------
constexpr unsigned SEL_INIT = 0x307a2000u;       // bits(1000.5f * 2^-40): "nothing selected" (vertex index 0 is a chain start)
constexpr float SEL_SCALE = 0x1p-40f;

__device__ __forceinline__ unsigned sgpr_const(const unsigned v) {   // a constant in an SGPR that the optimiser cannot fold into a literal
    unsigned m;
    asm("s_mov_b32 %0, %1" : "=s"(m) : "i"(v));
    return m;
}
// (a & m) | k with the vertex index as an inline constant (unrolled sweeps) or in a VGPR (loops)
template <int K> __device__ __forceinline__ unsigned and_or_k(unsigned a, unsigned m) {
    static_assert(K >= 0 && K <= 64, "inline constant");
    unsigned d;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(m), "n"(K));
    return d;
}
// v_pk_mul_f32 with the clamp modifier: clamp01(a * b) on two ray slots at once
__device__ __forceinline__ f32x2 pk_mul_clamp(const f32x2 a, const f32x2 b) {
    f32x2 d;
    asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// The flag thresholds for a car at (npx, npy).  With u = 2^-24: dir and the unit edge are rounded to float32 (relative u each), a
// cross product is a multiply and an fma; a_k = p_k - pos is the float32 difference of the vertex and the car, both taken
// relative to the track's anchor and rounded (each within u of a half-extent of the bounding box).  With M_k = |a_k|_inf <= R =
// the largest |coordinate difference| between the car and the track's bounding box (>= its half-extents), |c~_k - c_k| and
// |un~ - un| stay below 14 u R.  tau = 32 u R = 2^-19 R.
__device__ __forceinline__ float flag_threshold(const TrackHdr& h, const double npx, const double npy) {
    const float px = (float)npx, py = (float)npy;
    const float rx = fmaxf(fabsf(h.bx0 - px), fabsf(h.bx1 - px)), ry = fmaxf(fabsf(h.by0 - py), fabsf(h.by1 - py));
    return fmaxf(rx, ry) * 0x1p-19f;
}

// The per-lane sweep state shared by the three forms of the sweep (scalar-load loop, unrolled, LDS copy of the chain)
template <int RPL, bool TAB> struct Sweep {
    static constexpr int NP = (RPL + 1) / 2;
    f32x2 dx2[NP], dy2[NP];
    float cm[NP];     // per pair of ray slots: the smallest |c_k| over the vertices seen
    float um;         // the smallest |un'| over the segments seen
    unsigned keep;
    __device__ __forceinline__ void init(const float (&dx)[RPL], const float (&dy)[RPL], const unsigned idx_mask, unsigned (&bb)[2 * NP]) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {   // ray slots in PAIRS (packed fp32); an odd last slot is padded with a direction-0 ray
            dx2[j] = (f32x2){dx[2 * j], 2 * j + 1 < RPL ? dx[2 * j + 1] : 0.0f};
            dy2[j] = (f32x2){dy[2 * j], 2 * j + 1 < RPL ? dy[2 * j + 1] : 0.0f};
            bb[2 * j] = bb[2 * j + 1] = SEL_INIT;
            cm[j] = 1e30f;
        }
        um = 1e30f;
        keep = ~idx_mask;
    }
    // side values of a vertex at (vx, vy) for a car at (px, py), both relative to the track's anchor: a = p - pos, c = cross(a, dir)
    // per ray slot
    __device__ __forceinline__ void side(const float vx, const float vy, const float px, const float py, float& ax, float& ay,
                                         f32x2 (&c)[NP]) {
        ax = vx - px;
        ay = vy - py;
        const f32x2 ax2 = {ax, ax}, ay2 = {ay, ay};
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            c[j] = __builtin_elementwise_fma(ay2, dx2[j], -(ax2 * dy2[j]));
            if (2 * j + 1 < RPL) cm[j] = __builtin_fminf(__builtin_fminf(cm[j], __builtin_fabsf(c[j].x)), __builtin_fabsf(c[j].y));   // v_min3_f32 |.|
            else cm[j] = __builtin_fminf(cm[j], __builtin_fabsf(c[j].x));
        }
    }
    // candidate values (before the index is merged in) of the segment closed by a vertex with unit edge (ex, ey) / scaled unit edge
    // (exs, eys): (axp, ayp, cp) belong to the segment's first endpoint, c to the closing vertex; rd[s][I] = the slots' 1/den (TAB)
    __device__ __forceinline__ void cand(const float ex, const float ey, const float exs, const float eys, const float axp,
                                         const float ayp, const f32x2 (&cp)[NP], const f32x2 (&c)[NP], const f32x4 (&rd)[2 * NP],
                                         const int I, float (&u)[2 * NP]) {
        const float un = __builtin_fmaf(eys, axp, -(exs * ayp));
        // (a chain start's record carries (exs, eys) = (1, 0) beside its zero edge: |un'| = |ayp| there -- never small -- and its
        // candidate is still +-inf or NaN, (un' or 0) * (1 / 0))
        um = __builtin_fminf(um, __builtin_fabsf(un));
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const f32x2 P = pk_mul_clamp(cp[j], c[j]);
            float r0, r1 = 0.0f;
            if constexpr (TAB) {
                r0 = rd[2 * j][I];
                if (2 * j + 1 < RPL) r1 = rd[2 * j + 1][I];
            } else {
                const f32x2 ex2 = {ex, ex}, ey2 = {ey, ey};
                const f32x2 den = __builtin_elementwise_fma(ey2, dx2[j], -(ex2 * dy2[j]));  // = rden_build_kernel's
                r0 = __builtin_amdgcn_rcpf(den.x);
                r1 = __builtin_amdgcn_rcpf(den.y);
            }
            u[2 * j] = __builtin_fmaf(un, r0, P.x);
            if (2 * j + 1 < RPL) u[2 * j + 1] = __builtin_fmaf(un, r1, P.y);   // (odd RPL: the last slot is padding)
        }
    }
    // after the sweep: selection 0 ("nothing certified") for the slots of a flagged pair / of a flagged lane
    __device__ __forceinline__ void apply_flags(const float tau, unsigned (&bb)[2 * NP]) const {
        const bool lane_bad = um < tau * SEL_SCALE;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const bool bad = lane_bad | (cm[j] < tau);
            bb[2 * j] = bad ? 0u : bb[2 * j];
            if (2 * j + 1 < RPL) bb[2 * j + 1] = bad ? 0u : bb[2 * j + 1];
        }
    }
};

// Scalar-load loop form: vertex chain `vt` (nV vertices, a multiple of 4) read through wave-uniform scalar loads, part `part` of
// PARTS.  dx / dy = the slots' directions, didx = their lattice indices (TAB: rows of the 1/den table `rdl` in LDS; ADDR: didx
// holds the LDS BYTE ADDRESSES of those rows instead).  bb = the slots' selections (candidate bit patterns).
template <int RPL, int PARTS, bool TAB, bool ADDR = false>
__device__ __forceinline__ void wall_sweep_f32(const Vtx* vt, const int nV, const int part, const float pxr, const float pyr,
                                               const float (&dx)[RPL], const float (&dy)[RPL], const int (&didx)[RPL], lds_cfp rdl,
                                               const float tau, const unsigned idx_mask, unsigned (&bb)[2 * ((RPL + 1) / 2)]) {
    constexpr int NP = (RPL + 1) / 2;
    Sweep<RPL, TAB> sw;
    sw.init(dx, dy, idx_mask, bb);
    // vertex k closes the segment (k-1, k): (axp, ayp, cp) belong to k-1, c to k; rd = the slots' 1/den rows (TAB)
    auto close = [&](const float ex, const float ey, const float exs, const float eys, const int k, const float axp, const float ayp,
                     const f32x2 (&cp)[NP], const f32x2 (&c)[NP], const f32x4 (&rd)[2 * NP], const int I) {
        float u[2 * NP];
        sw.cand(ex, ey, exs, eys, axp, ayp, cp, c, rd, I, u);
        unsigned kv;
        asm("v_mov_b32 %0, %1" : "=v"(kv) : "s"(k));
#pragma unroll
        for (int s = 0; s < RPL; ++s) bb[s] = min(bb[s], and_or(__float_as_uint(u[s]), sw.keep, kv));
    };
    // Vertex GROUPS of four (the host pads every track's chain to a multiple of 4 with chain-start sentinels);
    // this part's groups [gbeg, gend).  The vertex before the range supplies the chain's previous side values.
    const int ngrp = nV >> 2;
    const int gbeg = PARTS > 1 ? ngrp * part / PARTS : 0;
    const int gend = PARTS > 1 ? ngrp * (part + 1) / PARTS : ngrp;
    // The "previous vertex" registers alternate between sets A and B (no copies); wave-uniform vertex records ->
    // s_load_dwordx8, prefetched one vertex ahead under the VALU work.
    float axA = 1.0f, ayA = 1.0f, axB = 1.0f, ayB = 1.0f;   // (nonzero: the first chain start's |un'| must not look like a car on a wall line)
    f32x2 cA[NP], cB[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) cA[j] = cB[j] = (f32x2){0.0f, 0.0f};
    if (PARTS > 1 && gbeg > 0) {
        const Vtx v = cload(vt + 4 * gbeg - 1);
        sw.side(v.xr, v.yr, pxr, pyr, axA, ayA, cA);
    }
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
#define PC_VERTEX(RD, I, PAX, PAY, PC, NAX, NAY, NC)                                                                     \
    {                                                                                                                \
        const Vtx v = nxt;                                                                                           \
        nxt = cload(vt + (k + I + 1 < 4 * gend ? k + I + 1 : k + I));                                                \
        sw.side(v.xr, v.yr, pxr, pyr, NAX, NAY, NC);                                                                   \
        if (!vtx_brk(v)) close(v.ex, v.ey, v.exs, v.eys, k + I, PAX, PAY, PC, NC, RD, I);                                                \
    }
    for (int gq = gbeg; gq < gend; ++gq) {
        f32x4 rd[2 * NP];
#pragma unroll
        for (int s = 0; s < 2 * NP; ++s) {
            rd[s] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
            if constexpr (TAB) rd[s] = rrow[s][gq - gbeg];
        }
        const int k = 4 * gq;
        PC_VERTEX(rd, 0, axA, ayA, cA, axB, ayB, cB)
        PC_VERTEX(rd, 1, axB, ayB, cB, axA, ayA, cA)
        PC_VERTEX(rd, 2, axA, ayA, cA, axB, ayB, cB)
        PC_VERTEX(rd, 3, axB, ayB, cB, axA, ayA, cA)
    }
#undef PC_VERTEX
    sw.apply_flags(tau, bb);
}

// The same sweep for a track whose chain has exactly NGRP groups of four vertices, fully unrolled and WITHOUT a branch per
// vertex (persistent big-form kernel in modes 1 / 2, a chain of 28 whose layout is not known at compile time; a track that is
// known to be two equal chains takes wall_sweep_loops in modes 3 / 4):
//   * the 1/den rows are read with immediate offsets (no address arithmetic per group), the vertex index is an inline constant;
//   * two consecutive vertices share one v_min3_u32 per ray slot instead of two v_min_u32;
//   * a chain-start vertex is not skipped but computed: its edge is (0, 0), so un' = 0 and 1/den = +-inf (what
//     rden_build_kernel's v_rcp_f32 of 0 stores, too), 0 * inf + P = NaN, whose bit pattern lies above every finite distance:
//     the candidate can never win the unsigned minimum.  Only the trailing padding pair(s) are skipped (n_chain).
// The minimum is exact, so the result is the very same bits as wall_sweep_f32's.
template <int RPL, bool TAB, int NGRP, bool ADDR = false>
__device__ __forceinline__ void wall_sweep_unrolled(const Vtx* vt, const int n_chain, const float pxr, const float pyr,
                                                    const float (&dx)[RPL], const float (&dy)[RPL], const int (&didx)[RPL], lds_cfp rdl,
                                                    const float tau, unsigned (&bb)[2 * ((RPL + 1) / 2)]) {
    constexpr int NP = (RPL + 1) / 2, nV = 4 * NGRP;
    static_assert(nV <= 32, "five index bits");
    Sweep<RPL, TAB> sw;
    sw.init(dx, dy, 31u, bb);
    const unsigned keep = sgpr_const(0xffffffe0u);
    float axA = 1.0f, ayA = 1.0f, axB = 1.0f, ayB = 1.0f;   // (nonzero: the first chain start's |un'| must not look like a car on a wall line)
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
    auto pair = [&](auto KC, const f32x4 (&rd)[2 * NP]) {   // vertices K, K + 1 (K even): sets A -> B -> A
        constexpr int K = decltype(KC)::value, I = K & 3;
        const Vtx v0 = cload(vt + K), v1 = cload(vt + K + 1);
        float u0[2 * NP], u1[2 * NP];
        sw.side(v0.xr, v0.yr, pxr, pyr, axB, ayB, cB);
        sw.cand(v0.ex, v0.ey, v0.exs, v0.eys, axA, ayA, cA, cB, rd, I, u0);
        sw.side(v1.xr, v1.yr, pxr, pyr, axA, ayA, cA);
        sw.cand(v1.ex, v1.ey, v1.exs, v1.eys, axB, ayB, cB, cA, rd, I + 1, u1);
#pragma unroll
        for (int s = 0; s < RPL; ++s)
            bb[s] = min(min(bb[s], and_or_k<K>(__float_as_uint(u0[s]), keep)), and_or_k<K + 1>(__float_as_uint(u1[s]), keep));   // v_min3_u32
    };
    auto group = [&](auto GC) {
        constexpr int gq = decltype(GC)::value;
        f32x4 rd[2 * NP];
#pragma unroll
        for (int s = 0; s < 2 * NP; ++s) {
            rd[s] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
            if constexpr (TAB) { if (s < RPL) rd[s] = rrow[s][gq]; }
        }
        pair(std::integral_constant<int, 4 * gq>{}, rd);
        if (gq < NGRP - 1 || 4 * gq + 2 < n_chain)   // (wave-uniform; only the last group can hold a padding pair)
            pair(std::integral_constant<int, 4 * gq + 2>{}, rd);
        // one scheduling region per group: left alone, the scheduler hoists every group's table rows and vertex records
        // to the top of the 1300-instruction block and spills
        __builtin_amdgcn_sched_barrier(0);
    };
    [&]<int... Gs>(std::integer_sequence<int, Gs...>) { (group(std::integral_constant<int, Gs>{}), ...); }(std::make_integer_sequence<int, NGRP>{});
    sw.apply_flags(tau, bb);
}

// The same sweep for a track whose walls are exactly TWO chains of L vertices each (TrackHdr::vtxp_off; big_track.json: L = 13,
// the outer and the inner loop), packed BY CHAIN: every packed-fp32 instruction carries position i of chain 0 in its low half
// and position i of chain 1 (vertex L + i) in its high half.  Against wall_sweep_unrolled -- which packs two RAY SLOTS per
// instruction --
//   * the per-vertex work that does not depend on the ray slot (a = p - pos, un', its flag) is packed as well: half the count;
//   * the candidate u = fma(un', 1/den, P) is ONE v_pk_fma_f32 per slot and pair of vertices, its 1/den operand one aligned
//     register pair of the slot's table row -- the table's LDS copy holds each row in the order (0, L, 1, L + 1, ...)
//     (rden_stage_loops);
//   * an odd slot count wastes nothing (5 + 4 slots used to cost 6 + 4), and the flags are per slot instead of per slot pair;
//   * both chain starts are position 0: their candidates are not formed, no sentinel is computed.
// Every candidate is the same arithmetic on the same operands as in wall_sweep_f32: the same bits, and the minimum is exact.
// Vertex records: wave-uniform scalar loads of VtxP (48 bytes).  rdl rows: [4 * ((L + 1) / 2)] floats.
template <int RPL, bool TAB, int L, bool ADDR = false>
__device__ __forceinline__ void wall_sweep_loops(const VtxP* vp, const float pxr, const float pyr, const float (&dx)[RPL],
                                                 const float (&dy)[RPL], const int (&didx)[RPL], lds_cfp rdl, const float tau,
                                                 unsigned (&bb)[2 * ((RPL + 1) / 2)]) {
    static_assert(2 * L <= 32, "five index bits");
    constexpr int NG = (L + 1) / 2, ROW = 4 * NG;
    const unsigned keep = sgpr_const(0xffffffe0u);
    const f32x2 px2 = {pxr, pxr}, py2 = {pyr, pyr};
    f32x2 axA, ayA, axB, ayB, cA[RPL], cB[RPL];
    float cm[RPL], um = 1e30f;
#pragma unroll
    for (int s = 0; s < 2 * ((RPL + 1) / 2); ++s) bb[s] = SEL_INIT;
#pragma unroll
    for (int s = 0; s < RPL; ++s) cm[s] = 1e30f;
    typedef const __attribute__((address_space(3))) f32x4* lds_row;
    lds_row rrow[RPL];
    if constexpr (TAB) {
#pragma unroll
        for (int s = 0; s < RPL; ++s) {
            if constexpr (ADDR) rrow[s] = (lds_row)(size_t)(unsigned)didx[s];
            else rrow[s] = (lds_row)(rdl + __umul24(didx[s], ROW));
        }
    }
    // side values of position i of both chains: a = p - pos, c[s] = cross(a, dir_s), and the smallest |c| per slot
    auto side = [&](const VtxP& v, f32x2& ax, f32x2& ay, f32x2(&c)[RPL]) {
        ax = v.xr - px2;
        ay = v.yr - py2;
#pragma unroll
        for (int s = 0; s < RPL; ++s) {
            const f32x2 dxs = {dx[s], dx[s]}, dys = {dy[s], dy[s]};
            c[s] = __builtin_elementwise_fma(ay, dxs, -(ax * dys));
            if constexpr (!(PC_ABLATE & 32)) cm[s] = __builtin_fminf(__builtin_fminf(cm[s], __builtin_fabsf(c[s].x)), __builtin_fabsf(c[s].y));   // v_min3_f32 |.|
        }
    };
    // the two segments closed by position I (vertices I and L + I): (axp, ayp, cp) belong to position I - 1, c to I
    auto cand = [&](auto IC, const VtxP& v, const f32x2 axp, const f32x2 ayp, const f32x2(&cp)[RPL], const f32x2(&c)[RPL],
                    const f32x4(&rd)[RPL]) {
        constexpr int I = decltype(IC)::value;
        const f32x2 un = __builtin_elementwise_fma(v.eys, axp, -(v.exs * ayp));
        um = __builtin_fminf(__builtin_fminf(um, __builtin_fabsf(un.x)), __builtin_fabsf(un.y));
#pragma unroll
        for (int s = 0; s < RPL; ++s) {
            const f32x2 P = pk_mul_clamp(cp[s], c[s]);
            f32x2 r;
            if constexpr (TAB) {
                r = (I & 1) ? (f32x2){rd[s][2], rd[s][3]} : (f32x2){rd[s][0], rd[s][1]};
            } else {
                const f32x2 dxs = {dx[s], dx[s]}, dys = {dy[s], dy[s]};
                const f32x2 den = __builtin_elementwise_fma(v.ey, dxs, -(v.ex * dys));   // = rden_build_kernel's
                r = (f32x2){__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
            }
            const f32x2 u = __builtin_elementwise_fma(un, r, P);
            bb[s] = min(min(bb[s], and_or_k<I>(__float_as_uint(u.x), keep)), and_or_k<L + I>(__float_as_uint(u.y), keep));   // v_min3_u32
        }
    };
    auto group = [&](auto GC) {      // positions 2 gq (set A) and 2 gq + 1 (set B)
        constexpr int gq = decltype(GC)::value;
        f32x4 rd[RPL];
#pragma unroll
        for (int s = 0; s < RPL; ++s) {
            rd[s] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
            if constexpr (TAB) rd[s] = rrow[s][gq];
        }
        {
            const VtxP v = cload(vp + 2 * gq);
            side(v, axA, ayA, cA);
            if constexpr (gq > 0) cand(std::integral_constant<int, 2 * gq>{}, v, axB, ayB, cB, cA, rd);
        }
        if constexpr (2 * gq + 1 < L) {
            const VtxP v = cload(vp + 2 * gq + 1);
            side(v, axB, ayB, cB);
            cand(std::integral_constant<int, 2 * gq + 1>{}, v, axA, ayA, cA, cB, rd);
        }
        __builtin_amdgcn_sched_barrier(0);   // one scheduling region per group (see wall_sweep_unrolled)
    };
    [&]<int... Gs>(std::integer_sequence<int, Gs...>) { (group(std::integral_constant<int, Gs>{}), ...); }(std::make_integer_sequence<int, NG>{});
    // selection 0 ("nothing certified") for a flagged slot / every slot of a flagged lane
    const bool lane_bad = um < tau * SEL_SCALE;
#pragma unroll
    for (int s = 0; s < RPL; ++s) bb[s] = (lane_bad | (cm[s] < tau)) ? 0u : bb[s];
}

// One CarEnv.step (car_env.py:693-760) + TransformReward + same-step auto-reset for the env whose state the
// 2^lg lanes of this group hold in `st` (updated in place, identically in every lane).  Lane g sweeps rays
// g, g + G, ...  Observation entries go to orow (global row), frow (pre-reset obs, optional) and lrow (an LDS
// copy for the persistent rollout kernel, optional).  The per-env scalars come back in registers.
// SEL (T = double, the per-step kernel): on a track inside the selector's limits (TrackHdr::sel_ok) the walls are not tested pair by
// pair: the float32 sweep selects each ray's wall and the literal arithmetic measures it (lit_fast / lit_careful, env_math.hpp) --
// the same bits as the filter form below, which stays for every other track and for the persistent filter kernel (K9d).
// (SEL 0: never; 1: where the track allows it, decided at run time -- both forms compiled in; 2: always -- the host has checked every
// track of the handle: the persistent generic kernel, which has no registers for both forms)
template <typename T, int RPL, int PARTS = 1, bool TAB = false, bool TWOPASS = false, int SEL = 0>
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
    if constexpr (sizeof(T) == 8) {      // F64: k indexes the track's rotation table (Math<double>): follow its transition entry
        k_new = st.k;
        if (turned) k_new = Math<double>::turn(p, h, st.k, left);
    }
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
    T dx[RPL], dy[RPL];
    double best[RPL];   // min(1000, distance) per ray slot (F32: the float64 refinement of the float32 sweep's selection)
    int didx[RPL];      // F32: the slots' direction-lattice indices
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
        best[s] = 1000.0;  // Ray.get_distance :198
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
        // Car.get_passed_gate (:394-408): the collision rays j * (n // 4), j < nc (four when 4 divides n; range(0, n, n // 4) has up
        // to seven otherwise) at the PREVIOUS pose against gate[next], cast in float64 (cast_d: one segment, strict test).  Any
        // lane can cast any ray (directions come from the lattice table), so the casts are dealt round-robin to the env's lanes
        // instead of falling on whichever lane owns those rays (with rays strided over the lanes: all on lane 0).
        const int k5o = 5 * Math<float>::mod72(st.k);
#pragma unroll 1
        for (int jj = 0; jj * G < p.nc; ++jj) {  // (uniform trip count)
            const int j = g + jj * G;
            const unsigned m = (unsigned)(k5o + (j < p.nc ? j : 0) * p.q * p.step_deg);
            const double2 cs = p.dirtab64[h.dir_off + (int)min(m, m - 360u)];
            const bool hit = cast_d(gate, opx, opy, cs.x, cs.y) < 10.0;  // :387,:390
            gate_hit |= hit & (j < p.nc);
        }
    }

    // ---- wall sweep: Car.get_distances (:360-374) -- also serves Car.check_collision (E2)
    if constexpr (sizeof(T) == 4) {
        unsigned bb[2 * ((RPL + 1) / 2) + 2];
        if constexpr (TWOPASS && RPL >= 9) {
            // inside a persistent kernel's generic mode (TWOPASS): the slots in two passes over the chain, as in the fast mode at 33
            // rays -- one pass keeps ~40 more registers alive than the kernel has beside its policy state (it spilled); the second
            // pass repeats only the per-vertex position arithmetic.  Same candidates, same minima: same bits.
            constexpr int R1 = (RPL + 1) / 2, R2 = RPL - R1;
            const float pxr = (float)(npx - h.ax0), pyr = (float)(npy - h.ay0), tau = flag_threshold(h, npx, npy);
            {
                unsigned ba[2 * ((R1 + 1) / 2)];
                wall_sweep_f32<R1, PARTS, TAB>(p.vtx + h.vtx_off, h.nV, part, pxr, pyr, *reinterpret_cast<const float(*)[R1]>(&dx[0]),
                                               *reinterpret_cast<const float(*)[R1]>(&dy[0]), *reinterpret_cast<const int(*)[R1]>(&didx[0]), rdl,
                                               tau, h.idx_mask, ba);
#pragma unroll
                for (int s = 0; s < R1; ++s) bb[s] = ba[s];
            }
            __builtin_amdgcn_sched_barrier(0);   // the passes one after the other
            {
                unsigned bc[2 * ((R2 + 1) / 2)];
                wall_sweep_f32<R2, PARTS, TAB>(p.vtx + h.vtx_off, h.nV, part, pxr, pyr, *reinterpret_cast<const float(*)[R2]>(&dx[R1]),
                                               *reinterpret_cast<const float(*)[R2]>(&dy[R1]), *reinterpret_cast<const int(*)[R2]>(&didx[R1]), rdl,
                                               tau, h.idx_mask, bc);
#pragma unroll
                for (int s = 0; s < R2; ++s) bb[R1 + s] = bc[s];
            }
        } else {
            unsigned b0[2 * ((RPL + 1) / 2)];
            wall_sweep_f32<RPL, PARTS, TAB>(p.vtx + h.vtx_off, h.nV, part, (float)(npx - h.ax0), (float)(npy - h.ay0), dx, dy, didx, rdl,
                                            flag_threshold(h, npx, npy), h.idx_mask, b0);
#pragma unroll
            for (int s = 0; s < RPL; ++s) bb[s] = b0[s];
        }
        if constexpr (PARTS > 1) {   // the parts' selections meet in LDS (the minimum is exact: the same bits as one wave sweeping everything)
            unsigned* ex = reinterpret_cast<unsigned*>(exch);
#pragma unroll
            for (int s = 0; s < RPL; ++s) {
                const int ray = g + s * G;
                if (ray < p.R) ex[ray * PARTS + part] = bb[s];
            }
            lds_barrier();
#pragma unroll
            for (int s = 0; s < RPL; ++s) {
                const int ray = g + s * G;
                if (ray < p.R) {
                    unsigned m = ex[ray * PARTS];
#pragma unroll
                    for (int q = 1; q < PARTS; ++q) m = min(m, ex[ray * PARTS + q]);
                    bb[s] = m;
                }
            }
        }
        // the float64 refinement of every slot's selection (refine_fast / refine_careful, env_math.hpp)
        const SegD* sg64 = p.seg64 + h.vtx_off;
        const auto segs = [sg64](const int k) { return sg64[k]; };
        uint64_t todo = 0;   // bit s: slot s needs the careful path (RPL <= 33)
#pragma unroll
        for (int s = 0; s < RPL; ++s) {
            if (g + s * G < p.R) {
                const double2 d64 = p.dirtab64[h.dir_off + didx[s]];
                bool ok;
                best[s] = refine_fast(sg64[bb[s] & h.idx_mask], npx, npy, d64.x, d64.y, ok);
                todo |= ok ? 0ull : 1ull << s;
            }
        }
        // the rare rest, one slot of one lane at a time through ONE copy of the careful code (select chains instead of
        // dynamically indexed registers)
        while (__builtin_expect(__builtin_amdgcn_ballot_w64(todo != 0) != 0, 0)) {
            const int s0 = todo ? __builtin_ctzll(todo) : -1;
            unsigned sel = 0;
            int di = 360;
#pragma unroll
            for (int s = 0; s < RPL; ++s) {
                sel = s == s0 ? bb[s] : sel;
                di = s == s0 ? didx[s] : di;
            }
            if (s0 >= 0) {
                const double2 d64 = p.dirtab64[h.dir_off + di];
                const double d = refine_careful((int)(sel & h.idx_mask), segs, h.nV, npx, npy, d64.x, d64.y);
#pragma unroll
                for (int s = 0; s < RPL; ++s) best[s] = s == s0 ? d : best[s];
                todo &= todo - 1;
            }
        }
    } else if (SEL == 2 || (SEL == 1 && h.sel_ok)) {      // (wave-uniform)
        // F64 on a track the float32 selector handles: the sweep on the literal directions rounded to float32 (what it is priced for:
        // flag_threshold), then the reference's literal arithmetic on each selection.  A slot without a ray sweeps slot 0's direction
        // (a zero direction would flag its pair partner at every step) and is not looked at.
        unsigned bb[2 * ((RPL + 1) / 2) + 2];
        const float pxr = (float)(npx - h.ax0), pyr = (float)(npy - h.ay0), tau = flag_threshold(h, npx, npy);
        // one pass over the chain for the slots [S0, S0 + RN): their float32 directions are formed here, not kept beside the float64 ones
        auto sweep_pass = [&](auto S0C, auto RNC) {
            constexpr int S0 = decltype(S0C)::value, RN = decltype(RNC)::value;
            float dxf[RN], dyf[RN];
            int di[RN];
#pragma unroll
            for (int s = 0; s < RN; ++s) {
                const bool valid = g + (S0 + s) * G < p.R;
                dxf[s] = valid ? (float)dx[S0 + s] : (float)dx[0];
                dyf[s] = valid ? (float)dy[S0 + s] : (float)dy[0];
                di[s] = 0;
            }
            unsigned ba[2 * ((RN + 1) / 2)];
            wall_sweep_f32<RN, 1, false>(p.vtx + h.vtx_off, h.nV, 0, pxr, pyr, dxf, dyf, di, nullptr, tau, h.idx_mask, ba);
#pragma unroll
            for (int s = 0; s < RN; ++s) bb[S0 + s] = ba[s];
        };
        if constexpr (TWOPASS && RPL >= 9) {      // inside a persistent kernel: the slots in two passes over the chain (see the float32 branch)
            constexpr int R1 = (RPL + 1) / 2;
            sweep_pass(std::integral_constant<int, 0>{}, std::integral_constant<int, R1>{});
            __builtin_amdgcn_sched_barrier(0);
            sweep_pass(std::integral_constant<int, R1>{}, std::integral_constant<int, RPL - R1>{});
        } else {
            sweep_pass(std::integral_constant<int, 0>{}, std::integral_constant<int, RPL>{});
        }
        const SegD* sg64 = p.seg64 + h.vtx_off;      // (F64 handles: (ex, ey) carry the wall's second endpoint)
        const auto segs = [sg64](const int k) { return sg64[k]; };
        uint64_t todo = 0;
#pragma unroll
        for (int s = 0; s < RPL; ++s) {
            if (g + s * G < p.R) {
                bool ok;
                const double d = lit_fast(sg64[bb[s] & h.idx_mask], npx, npy, dx[s], dy[s], ok);
                best[s] = d < 1000.0 ? d : 1000.0;      // :198
                todo |= ok ? 0ull : 1ull << s;
            }
            __builtin_amdgcn_sched_barrier(0);      // one literal cast at a time (see cast_ref_t)
        }
        while (__builtin_expect(__builtin_amdgcn_ballot_w64(todo != 0) != 0, 0)) {
            const int s0 = todo ? __builtin_ctzll(todo) : -1;
            unsigned sel = 0;
            double ddx = 0.0, ddy = 0.0;
#pragma unroll
            for (int s = 0; s < RPL; ++s) {
                sel = s == s0 ? bb[s] : sel;
                ddx = s == s0 ? (double)dx[s] : ddx;
                ddy = s == s0 ? (double)dy[s] : ddy;
            }
            if (s0 >= 0) {
                const double d = lit_careful((int)(sel & h.idx_mask), segs, p.segs + h.wall_off, h.S, npx, npy, ddx, ddy);
#pragma unroll
                for (int s = 0; s < RPL; ++s) best[s] = s == s0 ? (d < 1000.0 ? d : 1000.0) : best[s];
                todo &= todo - 1;
            }
        }
    } else if constexpr (SEL != 2) {
        // F64: Ray.get_distance (:186-213) over all walls, the reference's own arithmetic -- in two passes per block of 32 walls.
        //   Pass 1 FILTERS: Ray.cast's numerators and denominator formed exactly as cast_ref forms them (the rounded differences
        //   x3 - x4, y3 - y4 included, :166-176) and the hit test `0 < t < 1 and u > 0` (:178) decided on them WITHOUT a division --
        //   with IEEE division 0 < fl(n / d) <=> n, d of equal sign, fl(n / d) < 1 <=> |n| < |d| (see cast_exact) -- in its
        //   NON-strict form (signs by the sign bits, |tn| <= |den|): a superset of the hits, never a miss of one.  ~16 instructions
        //   per ray and wall where the two float64 divisions and the square root of the literal form cost ~70.
        //   Pass 2 runs the LITERAL cast_ref on the filtered (ray, wall) pairs only -- a ray crosses 1-3 of a track's walls -- and it
        //   alone decides and measures: a pair that is no hit after all returns 1000.0 as it always did.  Same bits as the loop
        //   over all pairs (min is order-independent), by construction.
        const Seg* walls = p.segs + h.wall_off;
        // (the ray slots in groups of at most five: a group's rounded differences and hit masks are alive together, not all RPL of
        // them -- 9 or 17 float64 slots per lane beside a persistent kernel's other state do not fit the register file)
        constexpr int SG = RPL <= 6 ? RPL : (RPL <= 10 ? (RPL + 1) / 2 : 5);
        constexpr unsigned EST_NONE = 0x7f800000u;      // +inf: no filtered pair
        auto slot_group = [&](auto G0C, auto CNTC) {
            constexpr int g0 = decltype(G0C)::value, SGN = decltype(CNTC)::value;     // slots g0 .. g0 + SGN - 1
            double mx[SGN], my[SGN];
#pragma unroll
            for (int q = 0; q < SGN; ++q) {
                const double x4 = npx + dx[g0 + q], y4 = npy + dy[g0 + q];     // :169
                mx[q] = npx - x4;                                               // (x3 - x4)
                my[q] = npy - y4;                                               // (y3 - y4)
            }
            for (int w0 = 0; w0 < h.S; w0 += 32) {
                const int wn = h.S - w0 < 32 ? h.S - w0 : 32;
                // Beside the filter: a float32 ESTIMATE of every filtered pair's distance, u = |unn| / |den| (the reference's own
                // quotient, car_env.py:176, good to ~3e-7 here), wall index in the five low mantissa bits, and per slot the smallest
                // and the second smallest of them (unsigned order = value order for non-negative floats).  Pass 2 then measures the
                // estimated-nearest pair with the literal cast_ref -- ONE pair per slot instead of every filtered one, of which a
                // wave's 64 lanes held up to seven -- and falls back to EVERY wall of the block only where the runner-up's estimate is
                // not clearly (1e-4 relative: 300 x the estimate's error, 1e11 x the literal arithmetic's) beyond the measured
                // distance: two walls hit at the same place, i.e. a corner.  The minimum of the literal distances is unchanged.
                unsigned e1[SGN], e2[SGN];
#pragma unroll
                for (int q = 0; q < SGN; ++q) { e1[q] = EST_NONE; e2[q] = EST_NONE; }
                Seg nxt = cload(walls + w0);
                for (int j = 0; j < wn; ++j) {
                    const Seg sg = nxt;  // wave-uniform -> s_load_dwordx8
                    nxt = cload(walls + w0 + (j + 1 < wn ? j + 1 : j));
                    const double ex = sg.x1 - sg.x2, ey = sg.y1 - sg.y2;      // (x1 - x2), (y1 - y2)
                    const double ax = sg.x1 - npx, ay = sg.y1 - npy;          // (x1 - x3), (y1 - y3)
                    const double unn = ex * ay - ey * ax;                     // u = -unn / den (:176)
                    const int un_hi = __double2hiint(unn);
                    const float unf = __builtin_fabsf((float)unn);
                    unsigned jv;
                    asm("v_mov_b32 %0, %1" : "=v"(jv) : "s"(j));
#pragma unroll
                    for (int q = 0; q < SGN; ++q) {
                        const double den = ex * my[q] - ey * mx[q];           // :171
                        const double tn = ax * my[q] - ay * mx[q];            // :175 numerator
                        const int den_hi = __double2hiint(den);
                        // 0 <= t: equal sign bits; t <= 1: |tn| <= |den|; u >= 0: -unn and den of equal sign bits, i.e. unn and den of different ones
                        const bool maybe = ((__double2hiint(tn) ^ den_hi) >= 0) & (__builtin_fabs(tn) <= __builtin_fabs(den)) & ((un_hi ^ den_hi) < 0);
                        // (an estimate that overflowed or is not a number -- a denominator below float32's range -- becomes 3e38: beyond every
                        // distance that matters, Ray.get_distance caps at 1000 px, car_env.py:198)
                        const float est = __builtin_fminf(unf * __builtin_amdgcn_rcpf(__builtin_fabsf((float)den)), 3.0e38f);
                        const unsigned cand = maybe ? ((__float_as_uint(est) & ~31u) | jv) : EST_NONE;
                        e2[q] = min(e2[q], max(e1[q], cand));
                        e1[q] = min(e1[q], cand);
                    }
                }
#pragma unroll
                for (int q = 0; q < SGN; ++q) {
                    const int s = g0 + q;
                    // the estimated-nearest pair, unless even it lies clearly beyond what an earlier block of walls gave
                    const bool need1 = e1[q] < EST_NONE && __uint_as_float(e1[q] & ~31u) <= (float)best[s] * 1.0001f;
                    if (__builtin_amdgcn_ballot_w64(need1) != 0) {
                        const Seg sg = walls[w0 + (int)(e1[q] & 31u)];       // (per-lane wall: a vector load of the 32-byte record)
                        const double d = cast_ref(sg.x1, sg.y1, sg.x2, sg.y2, npx, npy, dx[s], dy[s]);
                        if (need1 && d < best[s]) best[s] = d;  // :203-207
                    }
                    // the runner-up within 1e-4 of the measured minimum: the literal loop over every wall of the block for those lanes
                    const bool need_all = e2[q] != EST_NONE && !(__uint_as_float(e2[q] & ~31u) > (float)best[s] * 1.0001f);
                    if (__builtin_expect(__builtin_amdgcn_ballot_w64(need_all) != 0, 0)) {
                        for (int j = 0; j < wn; ++j) {
                            const Seg sg = cload(walls + w0 + j);
                            const double d = cast_ref(sg.x1, sg.y1, sg.x2, sg.y2, npx, npy, dx[s], dy[s]);
                            if (need_all && d < best[s]) best[s] = d;
                        }
                    }
                }
            }
        };
        // (the ray slots in groups of at most five: a group's rounded differences and estimates are alive together, not all RPL of
        // them -- 9 or 17 float64 slots per lane beside a persistent kernel's other state do not fit the register file)
        [&]<int... Is>(std::integer_sequence<int, Is...>) {
            (slot_group(std::integral_constant<int, Is * SG>{}, std::integral_constant<int, (Is * SG + SG <= RPL ? SG : RPL - Is * SG)>{}), ...);
        }(std::make_integer_sequence<int, (RPL + SG - 1) / SG>{});
    }
    const bool store = PARTS == 1 || part == 0;
    bool wall_hit = false;
#pragma unroll
    for (int s = 0; s < RPL; ++s) wall_hit |= ((colmask >> s) & 1) & (best[s] < 10.0);  // :390

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
    env_step_core<T, RPL, 1, false, false, sizeof(T) == 8 ? 1 : 0>(p, trk, g, p.lg, st, actions[e], reward_scale, obs + (size_t)e * p.D,
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
        if (h.rden_off < 0) continue;      // (no table for this track: see env_create_impl)
        const int total = 361 * h.nV;
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
            const int idx = i / h.nV, k = i - idx * h.nV;
            const Vtx v = p.vtx[h.vtx_off + k];
            const float2 d = p.dirtab[h.lat_off + idx];
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
        double best = 1000.0;
        if constexpr (sizeof(T) == 4) {   // F32: the float64 chain scan the step's refinement falls back to (same arithmetic per segment)
            const double2 d64 = p.dirtab64[h.dir_off + Math<float>::dir_index(p, 0, ray)];
            const SegD* sg64 = p.seg64 + h.vtx_off;
            best = scan_chain_d([sg64](const int k) { return sg64[k]; }, h.nV, npx, npy, d64.x, d64.y);
        } else {
            T dx, dy;
            Math<T>::ray_dir(p, h, ray, 0, h.start_rot, dx, dy);
            for (int w = 0; w < h.S; ++w) {
                const T d = Math<T>::cast(p.segs[h.wall_off + w], npx, npy, dx, dy);
                if (d < best) best = d;
            }
        }
        o[6 + ray] = Math<T>::norm_dist(best);
        if (ray < p.n_nominal && ray % p.q == 0 && best < 10.0) hit = true;
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
