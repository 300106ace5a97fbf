// rollout.hpp -- part of the single translation unit ppocar.hip (included there, in order; not a stand-alone header).
// K9 rollout_kernel / K9s rollout_small_kernel: the whole rollout as one persistent launch; the table-driven fast env step; developer stamps.
#pragma once

// ------------------------------------------------------------------------------------------
// Developer-only phase timeline of the big-form rollout kernel (make stamps -> libppocar_stamps.so, -DPC_STAMPS): workgroup 0's
// eight waves write s_memtime at every phase boundary of steps 64..71 into a device array that tools/k9_timeline.py reads back.
// The product build contains none of this.
#ifdef PC_STAMPS
constexpr int STAMP_T0 = 64, STAMP_NT = 8, STAMP_NPH = 12;
__device__ unsigned long long g_stamps[8 * STAMP_NT * STAMP_NPH];
#define PC_STAMP(ph)                                                                                                      \
    if (blockIdx.x == 0 && t >= STAMP_T0 && t < STAMP_T0 + STAMP_NT && lane == 0)                                        \
        g_stamps[((wave * STAMP_NT) + (t - STAMP_T0)) * STAMP_NPH + (ph)] = __builtin_amdgcn_s_memtime();
// the two stamps inside the env step: small form only (in the big form they cost registers the kernel does not have -- with
// them it spilled 400 VGPRs and the timeline measured the spills)
#define PC_STAMP_E(ph) if constexpr (PARTS > 1) { PC_STAMP(ph) }
__device__ unsigned long long g_stamps_u[16];    // the minibatch kernel (K10), workgroup 0, thread 0, of the last launch
#define PC_STAMP_U(ph) if (wg == 0 && threadIdx.x == 0) g_stamps_u[ph] = __builtin_amdgcn_s_memtime();
#else
#define PC_STAMP(ph)
#define PC_STAMP_E(ph)
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
    lds_f4c dir;         // [720] direction lattice, twice around: (cos, sin, LDS byte address of the direction's 1/den row,
                         //       LDS byte address of the direction's float64 (cos, sin) in dir64)
    lds_cfp reset;       // [D] the track's reset observation
    lds_cd2 vtx;         // [nV] the wall vertex chain, 32-byte Vtx records: (xr, yr, ex, ey), (exs, eys, -, -) -- small form only (nV <= 64)
    lds_cd2 seg;         // [nV] the chain for the float64 refinement: SegD records, 48 bytes each (nV <= 64)
    lds_cd2 dir64;       // [360] (cos, sin) float64 of the direction lattice
    lds_cfp rden;        // [361][nV] or unused
};
// Region offsets in floats.  The big form sweeps its chain through scalar loads and never reads `vtx`: its `seg` table takes that
// region; the small form keeps both.  The float64 direction lattice sits at a FIXED distance behind the float32 one, entry for
// entry, twice around like it (TWICE: a slot's float64 direction is one LDS read at its float32 entry's address + FT_D64_BYTES);
// the 33-ray big form has no 5.6 KB to spare and keeps one turn (the address is reduced mod 360 entries with two integer ops).
constexpr int FT_HEAD = 0, FT_WRAP = FT_HEAD + 72 * 4, FT_ACT = FT_WRAP + 76, FT_GATES = FT_ACT + 16 * 8,
              FT_DIR = FT_GATES + TAB_MAX_GATES * 8, FT_RESET = FT_DIR + 720 * 4, FT_VTX = FT_RESET + 40,
              FT_VTX_MAX = 64, FT_DIR64 = FT_VTX + FT_VTX_MAX * 12;   // (the vtx region holds 32-byte Vtx or 48-byte SegD records)
constexpr int FT_D64_BYTES = (FT_DIR64 - FT_DIR) * 4;
__host__ __device__ constexpr int ft_seg_small(bool twice) { return FT_DIR64 + (twice ? 720 : 360) * 4; }
__host__ __device__ constexpr int ft_floats(bool small, bool twice) { return ft_seg_small(twice) + (small ? FT_VTX_MAX * 12 : 0); }
// Dynamic LDS of rollout_kernel's fast modes in floats: weight image, the workgroup's 8 x (envs per wave) dense observation rows, the
// action slots, the gather tables, the 1/den table.  The ONE expression the kernel's carve-up and the host's launch size share.
constexpr int K9_ACT_SLOTS = 256;
__host__ __device__ constexpr int k9_fast_lds_floats(int img, int envs_per_wave, int D, bool twice, int rden_floats) {
    return img + 8 * envs_per_wave * D + K9_ACT_SLOTS + ft_floats(false, twice) + rden_floats;
}
static_assert(FT_ACT % 4 == 0 && FT_GATES % 4 == 0 && FT_DIR % 4 == 0 && FT_VTX % 4 == 0 && FT_DIR64 % 4 == 0,
              "16-byte aligned records");

// LIT (PC_DTYPE_F64 handles): the float64 heading table and direction lattice are not staged -- those kernels take headings and
// directions from the track's rotation table (glibc's values of the very angles, Math<double>) -- and the chain records carry the
// walls' second endpoints (lit_fast).
template <bool SMALL, bool TWICE, bool LIT = false>
__device__ __forceinline__ FastTabs stage_fast_tables(const EnvParams<float>& p, const TrackHdr& h0, const int trk, float* sTab,
                                                      const int tid, const int nthreads) {
    int* dst = reinterpret_cast<int*>(sTab);
    if constexpr (!LIT) {
        const int* head = reinterpret_cast<const int*>(p.headtab + h0.head_off);
        for (int i = tid; i < 72 * 4; i += nthreads) dst[FT_HEAD + i] = head[i];
    }
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
    constexpr int FT_SEG = SMALL ? ft_seg_small(TWICE) : FT_VTX;
    const float2* dir = p.dirtab + h0.lat_off;
    const unsigned rden_base = (unsigned)(size_t)(lds_cfp)(sTab + ft_floats(SMALL, TWICE));
    for (int i = tid; i < 720; i += nthreads) {
        const int j = i < 360 ? i : i - 360;
        const float2 cs = dir[j];
        *reinterpret_cast<f32x4*>(sTab + FT_DIR + 4 * i) = (f32x4){cs.x, cs.y, __uint_as_float(rden_base + (unsigned)(j * h0.nV) * 4u), 0.0f};
    }
    if constexpr (!LIT) {
        const int* d64 = reinterpret_cast<const int*>(p.dirtab64 + h0.dir_off);
        for (int i = tid; i < (TWICE ? 720 : 360) * 4; i += nthreads) dst[FT_DIR64 + i] = d64[i < 360 * 4 ? i : i - 360 * 4];
    }
    const int* ro = reinterpret_cast<const int*>(p.reset_obs + (size_t)trk * p.D);
    for (int i = tid; i < p.D; i += nthreads) dst[FT_RESET + i] = ro[i];
    if (h0.nV <= FT_VTX_MAX) {
        if constexpr (SMALL) {
            const int* vs = reinterpret_cast<const int*>(p.vtx + h0.vtx_off);
            for (int i = tid; i < h0.nV * 8; i += nthreads) dst[FT_VTX + i] = vs[i];
        }
        const int* sg = reinterpret_cast<const int*>(p.seg64 + h0.vtx_off);
        for (int i = tid; i < h0.nV * 12; i += nthreads) dst[FT_SEG + i] = sg[i];
    }
    FastTabs ft;
    ft.head = (lds_cd2)(sTab + FT_HEAD);
    ft.wrap = (lds_ci)(sTab + FT_WRAP);
    ft.act = (lds_cd2)(sTab + FT_ACT);
    ft.gates = (lds_cd4)(sTab + FT_GATES);
    ft.dir = (lds_f4c)(sTab + FT_DIR);
    ft.reset = (lds_cfp)(sTab + FT_RESET);
    ft.vtx = (lds_cd2)(sTab + FT_VTX);
    ft.seg = (lds_cd2)(sTab + FT_SEG);
    ft.dir64 = (lds_cd2)(sTab + FT_DIR64);
    ft.rden = (lds_cfp)(sTab + ft_floats(SMALL, TWICE));
    return ft;
}

// the same tables `floats` further on in LDS (MODE 6: the second track's block sits at a fixed distance behind the first's)
__device__ __forceinline__ FastTabs ft_shift(const FastTabs& a, const int floats) {
    FastTabs b;
    const int by = 4 * floats;
#define PC_FT_SH(m) b.m = (decltype(b.m))(size_t)((unsigned)(size_t)a.m + (unsigned)by)
    PC_FT_SH(head); PC_FT_SH(wrap); PC_FT_SH(act); PC_FT_SH(gates); PC_FT_SH(dir); PC_FT_SH(reset); PC_FT_SH(vtx); PC_FT_SH(seg); PC_FT_SH(dir64); PC_FT_SH(rden);
#undef PC_FT_SH
    return b;
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
// one observation entry as the fp16 x 2 policy pass wants it (policy.hpp: split_pair_h's arithmetic on a single value): scaled domain
// x 16, saturating at fp16's range, h = fp16(v), l = fp16(v - h) -- into row[0][col] and row[1][col] of an env's [2][32] halves
typedef __attribute__((address_space(3))) _Float16* lds_hp;
__device__ __forceinline__ void write_pieces(lds_hp row, const int col, const float v) {
    const float vs = clamp_h(v * PolScale<2>::sx);
    const _Float16 h = (_Float16)vs;
    const _Float16 l = (_Float16)__builtin_fmaf((float)h, opaque_neg_one(), vs);
    row[col] = h;
    row[32 + col] = l;
}
// exchange with the neighbouring lane (the other lane of the env): DPP quad_perm [1, 0, 3, 2], one VALU instruction
__device__ __forceinline__ int swap_pair(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xb1, 0xf, 0xf, false); }

// The float32 wall sweep (selector: see env_step.hpp) of the SMALL persistent form: part `part` of PARTS of the vertex chain, read
// from its LDS copy (ft.vtx) instead of through scalar loads.  A part is only one or two groups of four vertices, so what counts
// is latency, not issue slots: a group's vertex records and 1/den rows are all requested at its top, the four vertices' side
// values are independent instruction chains, chain-start vertices are computed rather than branched around (their candidates
// are NaN: see wall_sweep_unrolled), and two vertices share a v_min3_u32.  Same bits as wall_sweep_f32<RPL, PARTS, TAB>.
// VPART: `part` differs between the lanes of the wave (env_step_wave: the parts are lane groups, not waves).
template <int RPL, int PARTS, bool TAB, bool ADDR = false, bool VPART = false>
__device__ __forceinline__ void wall_sweep_lds(lds_cd2 vt, const int nV, const int part, const float pxr, const float pyr,
                                               const float (&dx)[RPL], const float (&dy)[RPL], const int (&didx)[RPL], lds_cfp rdl,
                                               const float tau, const unsigned idx_mask, unsigned (&bb)[2 * ((RPL + 1) / 2)]) {
    constexpr int NP = (RPL + 1) / 2;
    typedef const __attribute__((address_space(3))) f32x4* lds_f4;
    Sweep<RPL, TAB> sw;
    sw.init(dx, dy, idx_mask, bb);
    const int ngrp = nV >> 2;
    const int gbeg = PARTS > 1 ? ngrp * part / PARTS : 0;
    const int gend = PARTS > 1 ? ngrp * (part + 1) / PARTS : ngrp;
    float axA = 1.0f, ayA = 1.0f, axB = 1.0f, ayB = 1.0f;   // (nonzero: see wall_sweep_unrolled)
    f32x2 cA[NP], cB[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) cA[j] = cB[j] = (f32x2){0.0f, 0.0f};
    if (PARTS > 1 && gbeg > 0) {   // the vertex before the range: the chain's previous side values
        const f32x4 xy = *(lds_f4)(vt + 2 * (4 * gbeg - 1));
        sw.side(xy.x, xy.y, pxr, pyr, axA, ayA, cA);
    }
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
#pragma unroll
        for (int s = 0; s < 2 * NP; ++s) {
            rd[s] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
            if constexpr (TAB) { if (s < RPL) rd[s] = rrow[s][gq]; }
        }
        f32x4 xy[4];    // (xr, yr, ex, ey)
        f32x2 es[4];    // (exs, eys)
#pragma unroll
        for (int I = 0; I < 4; ++I) {
            xy[I] = *(lds_f4)(vt + 2 * (4 * gq + I));
            es[I] = *(lds_cf2)(vt + 2 * (4 * gq + I) + 1);
        }
        unsigned kv;
        if constexpr (VPART) kv = 4u * (unsigned)gq;
        else asm("v_mov_b32 %0, %1" : "=v"(kv) : "s"(4 * gq));
#pragma unroll
        for (int I = 0; I < 4; I += 2) {
            float u0[2 * NP], u1[2 * NP];
            sw.side(xy[I].x, xy[I].y, pxr, pyr, axB, ayB, cB);
            sw.cand(xy[I].z, xy[I].w, es[I].x, es[I].y, axA, ayA, cA, cB, rd, I, u0);
            sw.side(xy[I + 1].x, xy[I + 1].y, pxr, pyr, axA, ayA, cA);
            sw.cand(xy[I + 1].z, xy[I + 1].w, es[I + 1].x, es[I + 1].y, axB, ayB, cB, cA, rd, I + 1, u1);
#pragma unroll
            for (int s = 0; s < RPL; ++s)   // (index = 4 gq + I: the group's base in a register, I as an inline constant; v_min3_u32)
                bb[s] = min(min(bb[s], and_or(__float_as_uint(u0[s]), sw.keep, kv + I)), and_or(__float_as_uint(u1[s]), sw.keep, kv + I + 1));
        }
    }
    sw.apply_flags(tau, bb);
}

// LG = log2 of the lanes per env (1: K9, a wave owns 32 envs; 2: K9s, 16 envs per wave).  PARTS > 1 (K9s): the wall sweep is split
// over PARTS waves of the workgroup -- this wave sweeps vertex part `part`, the per-ray selections meet in LDS (`exch`: the env's
// [rays][PARTS] words) across a workgroup barrier, the float64 refinement is DEALT over the same waves (ray slot s is refined by
// part s % PARTS, which also writes that ray's column of the observation row), the parts' collision verdicts meet in LDS (`hitw`:
// the env's [PARTS] words) across a second barrier (every thread of the workgroup must make the call), and all waves finish
// the step on identical values; only `write_row` waves store the observation row.
// LIT (PC_DTYPE_F64 handles, big form only): the same step with the reference's LITERAL float64 arithmetic wherever a value is
// produced -- headings and ray directions are glibc's, read from the track's rotation table by the rotation's row st.k
// (Math<double>; *hcar carries the heading of the current rotation from step to step), the gate casts and the selected walls'
// distances are cast_ref's (lit_fast / lit_careful), the observation is normalised by division (Math<double>::norm) -- while
// the float32 sweep only SELECTS, on the float32 lattice directions, exactly as for F32 handles.  Bit for bit what
// env_step_core<double> computes (tests/test_rollout_f64_gpu.py).
template <int RPL, bool TAB, int LG = 1, int PARTS = 1, int SWP = 0, bool TWICE = true, bool LIT = false, bool COOP = false>
__device__ __forceinline__ bool env_step_fast(const EnvParams<float>& p, const TrackHdr& h, const FastTabs& ft, const FastLane& fl,
                                              const int (&gq)[2], const int g,
                                              EnvRegs& st, int& k72, const int a, const double reward_scale, lds_fp lrow,
                                              float& reward_f, float& term_f, float& trunc_f, const int t = 0, const int lane = 0,
                                              const int wave = 0, const int part = 0, float* exch = nullptr, const bool write_row = true,
                                              int* hitw = nullptr, f64x2* hcar = nullptr) {
    constexpr int G = 1 << LG;
    static_assert(!LIT || PARTS != 8, "the literal form: the big form and the four-part small form (wave-owned envs take env_step_wave)");
    const double2* rot_tab = p.dirtab64 + h.rot_off;      // LIT: row k = the R rays' (cos, sin) at rotation k, then (row of rot - 5, row of rot + 5), (rot, -)
    const int rot_ld = p.R + 2;
    // the float64 twin of the lattice entry at LDS byte address m (see FT_D64_BYTES)
    const unsigned dir_b = (unsigned)(size_t)ft.dir;
    const auto dir64_at = [&](const int m) {
        if constexpr (TWICE) return *(lds_cd2)(size_t)(unsigned)(m + FT_D64_BYTES);
        else {
            const unsigned off = (unsigned)m - dir_b;
            return *(lds_cd2)(size_t)(dir_b + (unsigned)FT_D64_BYTES + min(off, off - 5760u));
        }
    };
    // ---- action, heading before and after the turn (car_env.py:698-722, :440-442)
    const f64x2 Lf = ft.act[2 * a];                                     // (thrust, fric)
    const i32x2 Li = *(lds_ci2)(ft.act + 2 * a + 1);                    // (dk, fwd)
    struct { double thrust, fric; int dk, fwd; } L = {Lf.x, Lf.y, Li.x, Li.y};
    const int k72n = ft.wrap[k72 + L.dk + 1];
    f64x2 cs0, cs1;
    int kid_new = st.k;           // LIT: the rotation table's row after the turn (the only thing of it that stays in registers across the sweep)
    f64x2 gdir[LIT ? 4 / G : 1];  // LIT: the gate rays' directions at the PREVIOUS pose
    if constexpr (LIT) {
        cs0 = *hcar;
        const double2* row = rot_tab + st.k * rot_ld;
        const double2 lr = row[p.R];
#pragma unroll
        for (int jj = 0; jj < 4 / G; ++jj) {
            const double2 e = row[(g + G * jj) * p.q];          // Car.check_collision's rays j * (n // 4), dealt over the env's lanes
            gdir[jj] = (f64x2){e.x, e.y};
        }
        if (L.dk < 0) kid_new = (int)lr.x;      // rot - 5.0, :440
        if (L.dk > 0) kid_new = (int)lr.y;      // rot + 5.0, :442
    } else {
        cs0 = ft.head[k72];
        cs1 = ft.head[k72n];
    }
    // ---- Car.update physics (car_env.py:452-461), float64: thrust with the PRE-turn heading, friction without thrust, clip
    double nvx = (st.vx + cs0.x * L.thrust) * L.fric, nvy = (st.vy + cs0.y * L.thrust) * L.fric;
    nvx = fmin(fmax(nvx, -10.0), 10.0);      // np.clip per component (:457); the velocity is never NaN
    nvy = fmin(fmax(nvy, -10.0), 10.0);
    const double opx = st.px, opy = st.py;
    const double npx = opx + nvx, npy = opy + nvy;

    // ---- ray directions at the new heading: lattice entry 5 k + step_deg * ray (< 720: the table goes twice around), read as
    // (cos, sin, LDS address of the direction's 1/den row) through a byte address that costs one add per slot
    // (more than 12 slots per lane -- 33 rays -- are swept in passes: each pass reads its own slots' entries, so that no more
    // than one pass's directions are alive at a time)
    constexpr bool DIR_PER_PASS = RPL > 12;
    float dx[DIR_PER_PASS ? 1 : RPL], dy[DIR_PER_PASS ? 1 : RPL];
    int didx[DIR_PER_PASS ? 1 : RPL];    // TAB: the LDS byte address of the slot's 1/den row
    const int k80n = 80 * k72n;                                         // 16 bytes x 5 entries per turn step
    const int m0 = k80n + fl.rs0, m_last = k80n + fl.rs_last;
    if constexpr (!DIR_PER_PASS) {
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
    // ---- Car.get_passed_gate (:394-408): the four collision rays at the PREVIOUS pose against gate[next], dealt over the lanes,
    // cast in float64 (cast_d: the lattice entry names its float64 twin; the reference's own verdict at a tie)
    const f64x4 gv = ft.gates[st.next];
    const Seg gate = {gv.x, gv.y, gv.z, gv.w};
    const int k80o = 80 * k72;
    bool gate_hit = false;
    if constexpr (!(PC_ABLATE & 8)) {
#pragma unroll
        for (int jj = 0; jj < 4 / G; ++jj) {
            if constexpr (LIT) {
                gate_hit |= cast_ref(gate.x1, gate.y1, gate.x2, gate.y2, opx, opy, gdir[jj].x, gdir[jj].y) < 10.0;  // :387,:390 (the branchy form: the
                                                                                                                          // straight-line one costs the mixed-track kernel 14 spilled registers)
            } else {
                const f64x2 cs = dir64_at(k80o + gq[jj]);
                gate_hit |= cast_d(gate, opx, opy, cs.x, cs.y) < 10.0;  // :387,:390
            }
        }
    }
    // ---- wall sweep (float32 selector, env_step.hpp).  More than 12 ray slots per lane (33 rays: 17) are swept in TWO passes over
    // the vertex chain, 9 + 8 slots: one pass would need ~40 more registers than the 256 a wave has at two waves per SIMD (it
    // spilled 67 of them to scratch); the second pass repeats only the per-vertex position arithmetic (4 of ~50 instructions per
    // vertex and pass).
    // (33 rays, 17 slots per lane: two passes, 9 + 8.  The chain-packed sweep at 33 rays in three passes of 6 + 6 + 5 slots was built in
    // round 5 -- 3-4 % faster because hipcc shared the per-vertex work between the passes -- and was wrong about once in 1e9 entries:
    // exactly that sharing, see the RULE in sweep_pass and profiles/r6_cfg2_packed_rootcause.md.  Without it the variant gains 1 %: not built.)
    constexpr int NPASS = RPL > 12 ? 2 : 1;
    constexpr int R1 = (RPL + NPASS - 1) / NPASS;
    unsigned bb[RPL + 2];
    const float tau = flag_threshold(h, npx, npy);
    const float pxr = (float)(npx - h.ax0), pyr = (float)(npy - h.ay0);   // the car relative to the track's anchor (see Vtx)
    PC_STAMP_E(4)
    if constexpr (PC_ABLATE & 64) {
#pragma unroll
        for (int s = 0; s < RPL; ++s) bb[s] = 1u + (unsigned)s;
    } else {
        // one pass over the chain for the slots [S0, S0 + RN)
        auto sweep_pass = [&](auto S0C, auto RNC) {
            constexpr int S0 = decltype(S0C)::value, RN = decltype(RNC)::value;
            float dxp[RN], dyp[RN];
            int dip[RN];
            if constexpr (DIR_PER_PASS) {
#pragma unroll
                for (int s = 0; s < RN; ++s) {
                    const int m = m0 + (S0 + s) * fl.rstep;
                    const f32x4 cs = *(lds_f4c)(size_t)(unsigned)(S0 + s + 1 < RPL ? m : min(m, m_last));
                    dxp[s] = cs.x;
                    dyp[s] = cs.y;
                    dip[s] = (int)__float_as_uint(cs.z);
                }
            }
            const float(&dxa)[RN] = DIR_PER_PASS ? dxp : *reinterpret_cast<const float(*)[RN]>(&dx[DIR_PER_PASS ? 0 : S0]);
            const float(&dya)[RN] = DIR_PER_PASS ? dyp : *reinterpret_cast<const float(*)[RN]>(&dy[DIR_PER_PASS ? 0 : S0]);
            const int(&dia)[RN] = DIR_PER_PASS ? dip : *reinterpret_cast<const int(*)[RN]>(&didx[DIR_PER_PASS ? 0 : S0]);
            unsigned ba[2 * ((RN + 1) / 2)];
            // RULE (profiles/r6_cfg2_packed_rootcause.md): a sweep in several passes keeps NOTHING per-vertex from one pass to the next.
            // Left alone, hipcc computes the slot-independent per-vertex values (a = p - pos, un') in the first pass and keeps them in
            // VGPRs for the later ones; packed-fp32 instructions that read such long-lived registers while the SIMD's other wave is active
            // returned a wrong candidate about once in 1e9 entries (round 5's three-pass variant).  Every pass therefore works from its own
            // opaque copy of the car's position and of the chain's address: it loads and derives everything itself.
            float pxp = pxr, pyp = pyr;
            const Vtx* vtp = p.vtx + h.vtx_off;
            const VtxP* vpp = p.vtxp + h.vtxp_off;
            if constexpr (NPASS > 1) asm volatile("" : "+v"(pxp), "+v"(pyp), "+s"(vtp), "+s"(vpp));
            if (PARTS > 1)                  // small form: latency-oriented sweep over the LDS copy of the chain
                wall_sweep_lds<RN, PARTS, TAB, true>(ft.vtx, h.nV, part, pxp, pyp, dxa, dya, dia, ft.rden, tau, h.idx_mask, ba);
            else if constexpr (SWP == 7)       // the host guarantees big_track's layout (two loops of 13 vertices, packed: TrackHdr::vtxp_off)
                wall_sweep_loops<RN, TAB, 13, true>(vpp, pxp, pyp, dxa, dya, dia, ft.rden, tau, ba);
            else if constexpr (SWP == 5) {     // ... or two loops of 13 or of 9 vertices per track (track.json: 8 walls each): a mixed batch, workgroup-uniform
                if (h.brk2 == 13) wall_sweep_loops<RN, TAB, 13, true>(vpp, pxp, pyp, dxa, dya, dia, ft.rden, tau, ba);
                else wall_sweep_loops<RN, TAB, 9, true>(vpp, pxp, pyp, dxa, dya, dia, ft.rden, tau, ba);
            }
            else if (h.nV == 28)               // (wave-uniform) a chain of 28: the unrolled sweep
                wall_sweep_unrolled<RN, TAB, 7, true>(vtp, h.n_chain, pxp, pyp, dxa, dya, dia, ft.rden, tau, ba);   // chain length
            else if constexpr (SWP == 0)       // and the generic loop is not even compiled in (1 % from the shorter kernel alone)
                wall_sweep_f32<RN, PARTS, TAB, true>(vtp, h.nV, part, pxp, pyp, dxa, dya, dia, ft.rden, tau, h.idx_mask, ba);
#pragma unroll
            for (int s = 0; s < RN; ++s) bb[S0 + s] = ba[s];
        };
        sweep_pass(std::integral_constant<int, 0>{}, std::integral_constant<int, R1>{});
        if constexpr (NPASS >= 2) {
            __builtin_amdgcn_sched_barrier(0);   // the passes one after the other
            constexpr int RB = NPASS == 2 ? RPL - R1 : R1;
            sweep_pass(std::integral_constant<int, R1>{}, std::integral_constant<int, RB>{});
        }
        if constexpr (NPASS >= 3) {
            __builtin_amdgcn_sched_barrier(0);
            sweep_pass(std::integral_constant<int, 2 * R1>{}, std::integral_constant<int, RPL - 2 * R1>{});
        }
    }
    if constexpr (PARTS > 1) {   // the parts' selections meet in LDS (min is exact: the same bits as one wave sweeping everything)
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
            if (s % PARTS == part) {     // (wave-uniform) this part refines slot s
                const int r = s + 1 < RPL ? ray : min(ray, p.R - 1);
                unsigned m = ex[r * PARTS];
#pragma unroll
                for (int q = 1; q < PARTS; ++q) m = min(m, ex[r * PARTS + q]);
                bb[s] = m;
            }
            ray += G;
        }
    }
    PC_STAMP_E(5)
    // ---- float64 refinement of every slot's selection (refine_fast / refine_careful, env_math.hpp): the distance the observation
    // reports and Car.check_collision (:376-392) tests against 10 px.  The chain tables are read from LDS; a slot's float64
    // direction is found through its lattice entry (re-read: nine addresses are cheaper to keep than nine more live registers in
    // the sweep).
    const lds_cd2 sgl = ft.seg;
    const auto segs = [sgl](const int k) {     // one 48-byte record = three 16-byte LDS reads
        const f64x2 a = sgl[3 * k], b = sgl[3 * k + 1], c = sgl[3 * k + 2];
        return SegD{a.x, a.y, b.x, b.y, c.x, (int)(unsigned)__double_as_longlong(c.y), 0};
    };
    const double2* rot_row_new = rot_tab + kid_new * rot_ld;
    const auto ray_of = [&](const int s) { return min(g + G * s, p.R - 1); };
    bool wall_hit = false;
    unsigned todo = 0;   // bit s: slot s needs the careful path
    // Car.check_collision's slots as WAVE masks built on the scalar unit: the lanes g, g + G, ... of the wave share one colmask
    // (fast_lane), so slot s is a collision slot on the lane set (bit s of that colmask ? those lanes : none).
    constexpr uint64_t LANES_G0 = G == 2 ? 0x5555555555555555ull : 0x1111111111111111ull;
    int colm_g[G];
#pragma unroll
    for (int gg = 0; gg < G; ++gg) colm_g[gg] = __builtin_amdgcn_readlane(fl.colmask, gg);
    uint64_t hit_mask = 0;
    // Slots in batches: every LDS read of a batch -- the selected segments' records, the slots' float64 directions -- is issued
    // before the first float64 instruction, so that the wave pays the LDS round trip once per batch, not once per slot.
    // (LIT: three slots per batch, and one slot's literal cast -- two float64 divisions and a square root -- at a time: interleaved,
    // five of them held 165 registers more than the wave has)
    constexpr int NB = LIT ? 3 : (RPL <= 5 ? RPL : (RPL <= 9 ? (RPL + 1) / 2 : (RPL + 3) / 4));   // (14 registers per slot in flight: all nine of the 17-ray kernel at once spill)
#pragma unroll
    for (int s0 = 0; s0 < RPL; s0 += NB) {
        SegD sg[NB];
        f64x2 d64[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int s = s0 + j;
            if (s < RPL && (PARTS == 1 || s % PARTS == part)) {
                const int m = m0 + s * fl.rstep;
                sg[j] = segs((int)(bb[s] & h.idx_mask));
                if constexpr (LIT) {
                    const double2 e = rot_row_new[ray_of(s)];
                    d64[j] = (f64x2){e.x, e.y};
                } else {
                    d64[j] = dir64_at(s + 1 < RPL ? m : min(m, m_last));
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int s = s0 + j;
            if (s < RPL && (PARTS == 1 || s % PARTS == part)) {
                bool ok = true;
                double d = 500.0 + sg[j].x1 * 1e-9 + d64[j].x * 1e-9;
                if constexpr (LIT) d = lit_fast(sg[j], npx, npy, d64[j].x, d64[j].y, ok);
                else if constexpr (!(PC_ABLATE & 16)) d = refine_fast(sg[j], npx, npy, d64[j].x, d64[j].y, ok);
                todo |= ok ? 0u : 1u << s;
                uint64_t col_lanes = 0;
#pragma unroll
                for (int gg = 0; gg < G; ++gg) col_lanes |= ((colm_g[gg] >> s) & 1) ? LANES_G0 << gg : 0ull;
                hit_mask |= __builtin_amdgcn_ballot_w64(ok & (d < 10.0)) & col_lanes;           // :387-390 on Car.check_collision's rays
                const float o = LIT ? Math<double>::norm_dist(d < 1000.0 ? d : 1000.0) : obs_dist(d);    // :198, :593
                if (write_row || PARTS > 1) {   // ray slot s -> column 6 + ray(s): G floats apart from the lane's first; the clamped last slot apart
                    if (s + 1 < RPL) fl.lray[G * s] = o;
                    else fl.llast[0] = o;
                }
                if constexpr (LIT) __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    wall_hit |= __builtin_amdgcn_inverse_ballot_w64(hit_mask);
    PC_STAMP_E(8)
    // the rare rest.  COOP (every lane of the wave is active here: the caller's promise): one slot of ONE lane at a time as a job of the
    // whole wave (careful_wave, env_math.hpp: lane j measures chain segment j) -- ~100 instructions per slot where the owning lane alone
    // needs up to ~900
    if constexpr (COOP) {
        // (every branch of these loops is on wave-uniform SCALAR values -- the mask of lanes with work, taken once, and the owning lane's
        // slot bits -- and no ballot sits behind a divergent branch: hipcc threads a `while (ballot(todo))` through an earlier
        // `if (lane-condition)` and then runs the loop body with part of the wave masked off, which a per-lane body tolerates and a
        // whole-wave job does not)
        const int ln = threadIdx.x & 63;
        uint64_t tm = __builtin_amdgcn_ballot_w64(todo != 0);
        asm volatile("" : "+s"(tm));
        while (__builtin_expect(tm != 0, 0)) {
            const int L = __builtin_ctzll(tm);                       // the job's lane
            unsigned tl = (unsigned)__builtin_amdgcn_readlane((int)todo, L);      // ... and its careful slots
            do {
                const int s0 = __builtin_ctz(tl);
                unsigned sel = 0;
#pragma unroll
                for (int s = 0; s < RPL; ++s) sel = s == s0 ? bb[s] : sel;
                const int ms = m0 + s0 * fl.rstep;
                f64x2 dd;
                if constexpr (LIT) {
                    const double2 e = rot_row_new[ray_of(s0)];
                    dd = (f64x2){e.x, e.y};
                } else {
                    dd = dir64_at(s0 + 1 < RPL ? ms : min(ms, m_last));
                }
                const double d = careful_wave<LIT>(__builtin_amdgcn_readlane((int)(sel & h.idx_mask), L), segs, h.nV, p.segs + h.wall_off, h.S,
                                                   bcast_d(npx, L), bcast_d(npy, L), bcast_d(dd.x, L), bcast_d(dd.y, L), ln);
                if (ln == L) {
                    wall_hit |= (bool)((fl.colmask >> s0) & 1) & (d < 10.0);
                    if (write_row || PARTS > 1) {
                        const lds_fp dst = s0 + 1 < RPL ? fl.lray + G * s0 : fl.llast;
                        dst[0] = LIT ? Math<double>::norm_dist(d < 1000.0 ? d : 1000.0) : obs_dist(d);
                    }
                }
                tl &= tl - 1;
            } while (tl != 0);
            tm &= tm - 1;
        }
    }
    // ... else one slot of one lane at a time through ONE copy of the per-lane careful code (a select chain picks the slot's selection)
    while (!COOP && __builtin_expect(__builtin_amdgcn_ballot_w64(todo != 0) != 0, 0)) {
        const int s0 = todo ? __builtin_ctz(todo) : -1;
        unsigned sel = 0;
#pragma unroll
        for (int s = 0; s < RPL; ++s) sel = s == s0 ? bb[s] : sel;
        if (s0 >= 0) {
            const int ms = m0 + s0 * fl.rstep;
            double d;
            if constexpr (LIT) {
                const double2 e = rot_row_new[ray_of(s0)];
                d = lit_careful((int)(sel & h.idx_mask), segs, p.segs + h.wall_off, h.S, npx, npy, e.x, e.y);
            } else {
                const f64x2 d64 = dir64_at(s0 + 1 < RPL ? ms : min(ms, m_last));
                d = refine_careful((int)(sel & h.idx_mask), segs, h.nV, npx, npy, d64.x, d64.y);
            }
            wall_hit |= (bool)((fl.colmask >> s0) & 1) & (d < 10.0);
            if (write_row || PARTS > 1) {
                const lds_fp dst = s0 + 1 < RPL ? fl.lray + G * s0 : fl.llast;
                dst[0] = LIT ? Math<double>::norm_dist(d < 1000.0 ? d : 1000.0) : obs_dist(d);
            }
            todo &= todo - 1;
        }
    }
    PC_STAMP_E(9)
    int flags = (gate_hit ? 1 : 0) | (wall_hit ? 2 : 0);
    flags |= swap_pair(flags);                                          // any() over the env's G lanes
    if constexpr (G == 4) flags |= __builtin_amdgcn_update_dpp(0, flags, 0x4e, 0xf, 0xf, false);   // quad_perm [2, 3, 0, 1]
    static_assert(G == 2 || G == 4, "2 or 4 lanes per env");
    // PARTS == 8 (16 envs per workgroup: every wave holds ALL of the workgroup's envs): the barrier at which the parts' verdicts
    // meet is also the one behind which the observation rows are complete -- the step's LAST barrier (the caller adds one more
    // only in a step in which some env of the workgroup finished: that is the same decision in every wave).  So the six
    // kinematic columns, which depend on no verdict, are written before it.
    constexpr bool MERGED = PARTS == 8;
    if (MERGED && g == 0 && write_row) {
        lrow[0] = Math<float>::norm(npx, 1280.0);  // :578-581
        lrow[1] = Math<float>::norm(npy, 720.0);
        lrow[2] = Math<float>::norm(nvx, 10.0);
        lrow[3] = Math<float>::norm(nvy, 10.0);
        lrow[4] = (float)cs1.x;                    // :584-588
        lrow[5] = (float)cs1.y;
    }
    if constexpr (PARTS > 1) {   // every part has judged its own ray slots: the env's verdict is the OR over the parts
        if (g == 0) hitw[part] = flags;
        lds_barrier();
#pragma unroll
        for (int q = 0; q < PARTS; ++q) flags |= hitw[q];
    }
    PC_STAMP_E(10)
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
    // ---- observation row -> LDS: the ray columns were written by the refinement loop (the reset observation of a finished env is
    // written by the caller's fix-up)
    if constexpr (LIT) {      // the heading after the turn (rot + 0 * step is rot itself), read only now: nothing of it lives across the sweep
        const double2 e1 = rot_tab[kid_new * rot_ld];
        cs1 = (f64x2){e1.x, e1.y};
    }
    if (!MERGED && g == 0 && write_row) {
        using M = Math<std::conditional_t<LIT, double, float>>;
        lrow[0] = M::norm(npx, 1280.0);  // :578-581
        lrow[1] = M::norm(npy, 720.0);
        lrow[2] = M::norm(nvx, 10.0);
        lrow[3] = M::norm(nvy, 10.0);
        lrow[4] = (float)cs1.x;                    // :584-588
        lrow[5] = (float)cs1.y;
    }
    // ---- new state (CarEnv.reset for a finished env, :677-686, is the caller's rarely taken fix-up: env_reset_fast)
    st.px = npx;
    st.py = npy;
    st.vx = nvx;
    st.vy = nvy;
    if constexpr (LIT) {
        st.k = kid_new;
        *hcar = cs1;
    } else {
        st.k += L.dk;
    }
    k72 = k72n;
    st.time = time;
    st.next = next;
    st.passed = passed;
    return done;
}

// CarEnv.reset (car_env.py:677-686) of a finished env's registers (F64 handles: row 0 of the rotation table is start_rot)
__device__ __forceinline__ void env_reset_fast(const TrackHdr& h, EnvRegs& st, int& k72) {
    st.px = h.start_x; st.py = h.start_y; st.vx = 0.0; st.vy = 0.0; st.rot = h.start_rot;
    st.k = 0; st.time = 0; st.next = 0; st.passed = 0;
    k72 = 0;
}

// The env step of K9s at 16 envs per workgroup: a WAVE owns two envs outright, 32 lanes per env = 8 sweep parts x 4 ray groups --
// env_step_fast<RPL, TAB, 2, 8>'s decomposition with the parts as lane groups of ONE wave instead of the eight waves of the
// workgroup.  Same per-lane work (a part sweeps at most one group of four vertices of big_track's chain for its RPL ray slots),
// but nothing crosses a wave: the parts' selections meet through two DPP row rotations and one ds_swizzle per slot (min is
// exact: the same bits), part s refines ray slot s and writes its observation column, Car.check_collision's verdict is one
// ballot -- where the eight-wave form paid two LDS exchanges and two workgroup barriers per step (the selections, the verdicts)
// plus the one between the draw and the env step (a wave draws for exactly the two envs it steps: the action never leaves it).
// lane = 32 (env of the wave) + 4 part + g.
// LIT: the literal form for PC_DTYPE_F64 handles, as in env_step_fast (rotation table rows, cast_ref_t, lit_fast / lit_careful).
template <int RPL, bool TAB, bool LIT = false>
__device__ __forceinline__ bool env_step_wave(const EnvParams<float>& p, const TrackHdr& h, const FastTabs& ft, const FastLane& fl,
                                              const int gq0, const int g, const int part, EnvRegs& st, int& k72, const int a,
                                              const double reward_scale, lds_fp lrow, float& reward_f, float& term_f, float& trunc_f,
                                              const int lane, const int t = 0, const int wave = 0, f64x2* hcar = nullptr,
                                              lds_hp xrow = nullptr) {   // (t, wave: the developer stamps)
    // xrow (fp16 x 2 policy arithmetic): this env's two rows of PRE-SPLIT policy operands [2 pieces][32 features] in LDS.  The lane that
    // produces an observation entry also writes its scaled-domain pieces h = fp16(16 v), l = fp16(16 v - h) there -- once per entry,
    // where the eight waves of the workgroup used to split all 16 x 32 entries each at the head of their policy pass (split8's bits).
    constexpr int G = 4, PARTS = 8, NP = (RPL + 1) / 2;
    const double2* rot_tab = p.dirtab64 + h.rot_off;      // LIT: the track's rotation table (see env_step_fast)
    const int rot_ld = p.R + 2;
    static_assert(RPL <= PARTS && RPL <= 12, "one refinement per lane, one sweep pass");
    // ---- action, heading, Car.update physics: env_step_fast's, instruction for instruction
    const f64x2 Lf = ft.act[2 * a];
    const i32x2 Li = *(lds_ci2)(ft.act + 2 * a + 1);
    struct { double thrust, fric; int dk, fwd; } L = {Lf.x, Lf.y, Li.x, Li.y};
    const int k72n = ft.wrap[k72 + L.dk + 1];
    f64x2 cs0, cs1, gdir = {0.0, 0.0};
    int kid_new = st.k;
    if constexpr (LIT) {
        cs0 = *hcar;
        const double2* row = rot_tab + st.k * rot_ld;
        const double2 lr = row[p.R], e = row[g * p.q];       // (rot - 5, rot + 5)'s rows; collision ray j = g at the previous pose
        gdir = (f64x2){e.x, e.y};
        if (L.dk < 0) kid_new = (int)lr.x;      // :440
        if (L.dk > 0) kid_new = (int)lr.y;      // :442
    } else {
        cs0 = ft.head[k72];
        cs1 = ft.head[k72n];
    }
    double nvx = (st.vx + cs0.x * L.thrust) * L.fric, nvy = (st.vy + cs0.y * L.thrust) * L.fric;
    nvx = fmin(fmax(nvx, -10.0), 10.0);
    nvy = fmin(fmax(nvy, -10.0), 10.0);
    const double opx = st.px, opy = st.py;
    const double npx = opx + nvx, npy = opy + nvy;
    float dx[RPL], dy[RPL];
    int didx[RPL];
    const int k80n = 80 * k72n;
    const int m0 = k80n + fl.rs0, m_last = k80n + fl.rs_last;
    {
        int m = m0;
#pragma unroll
        for (int s = 0; s < RPL; ++s) {
            const f32x4 cs = *(lds_f4c)(size_t)(unsigned)(s + 1 < RPL ? m : min(m, m_last));
            dx[s] = cs.x;
            dy[s] = cs.y;
            didx[s] = (int)__float_as_uint(cs.z);
            m += fl.rstep;
        }
    }
    // ---- Car.get_passed_gate: collision ray j = g at the previous pose against gate[next], float64 (every part: the same values)
    const f64x4 gv = ft.gates[st.next];
    const Seg gate = {gv.x, gv.y, gv.z, gv.w};
    bool gate_hit;
    if constexpr (LIT) {
        gate_hit = cast_ref_t(gate.x1, gate.y1, gate.x2, gate.y2, opx, opy, gdir.x, gdir.y).d < 10.0;
    } else {
        const f64x2 cs = *(lds_cd2)(size_t)(unsigned)(80 * k72 + gq0 + FT_D64_BYTES);
        gate_hit = cast_d(gate, opx, opy, cs.x, cs.y) < 10.0;
    }
    // ---- wall sweep: this lane's part of the chain for its RPL slots, then the minimum over the env's 8 parts (lanes of equal g)
    unsigned bb[2 * NP];
    const float tau = flag_threshold(h, npx, npy);
    const float pxr = (float)(npx - h.ax0), pyr = (float)(npy - h.ay0);
    PC_STAMP(4)
    wall_sweep_lds<RPL, PARTS, TAB, true, true>(ft.vtx, h.nV, part, pxr, pyr, dx, dy, didx, ft.rden, tau, h.idx_mask, bb);
#pragma unroll
    for (int s = 0; s < RPL; ++s) {
        unsigned m = bb[s];
        m = min(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x124, 0xf, 0xf, false));      // row_ror:4  } the four parts of
        m = min(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x128, 0xf, 0xf, false));      // row_ror:8  } the 16-lane row
        m = min(m, (unsigned)__builtin_amdgcn_ds_swizzle((int)m, 0x401f));                         // lane ^ 16: the env's other row
        bb[s] = m;
    }
    PC_STAMP(5)
    // ---- float64 refinement: part s takes ray slot s (parts >= RPL repeat the last slot and are masked out)
    const lds_cd2 sgl = ft.seg;
    const auto segs = [sgl](const int k) {
        const f64x2 a = sgl[3 * k], b = sgl[3 * k + 1], c = sgl[3 * k + 2];
        return SegD{a.x, a.y, b.x, b.y, c.x, (int)(unsigned)__double_as_longlong(c.y), 0};
    };
    const bool active = part < RPL;
    const int slot = active ? part : RPL - 1;
    unsigned sel = bb[0];
#pragma unroll
    for (int s = 1; s < RPL; ++s) sel = slot == s ? bb[s] : sel;
    const int ms = m0 + slot * fl.rstep;
    const bool is_last = slot == RPL - 1;
    f64x2 d64;
    if constexpr (LIT) {
        const double2 e = rot_tab[kid_new * rot_ld + min(g + G * slot, p.R - 1)];
        d64 = (f64x2){e.x, e.y};
    } else {
        d64 = *(lds_cd2)(size_t)(unsigned)((is_last ? min(ms, m_last) : ms) + FT_D64_BYTES);
    }
    const SegD sg = segs((int)(sel & h.idx_mask));
    bool ok = true;
    double d = LIT ? lit_fast(sg, npx, npy, d64.x, d64.y, ok) : refine_fast(sg, npx, npy, d64.x, d64.y, ok);
    const auto obs_of = [](const double dd) { return LIT ? Math<double>::norm_dist(dd < 1000.0 ? dd : 1000.0) : obs_dist(dd); };   // :198, :593
    bool todo = active & !ok;
    // the lanes with a careful job, taken HERE: ahead of every lane-dependent branch (see env_step_fast's whole-wave jobs)
    uint64_t tm = __builtin_amdgcn_ballot_w64(todo);
    asm volatile("" : "+s"(tm));
    const bool col = (bool)((fl.colmask >> slot) & 1) & active;
    bool hit = col & ok & (d < 10.0);                                                              // :387-390 on Car.check_collision's rays
    const lds_fp dst = is_last ? fl.llast : fl.lray + G * slot;
    if (active) dst[0] = obs_of(d);
    PC_STAMP(8)
    // the rare rest: the careful path, one ray at a time as a job of the whole wave (careful_wave, env_math.hpp; every lane is active here).
    // The owning lane alone needs up to ~900 instructions for it -- with the workgroup's other seven waves waiting at the step's barrier
    while (__builtin_expect(tm != 0, 0)) {
        const int L = __builtin_ctzll(tm);       // the job's lane (wave-uniform)
        const double dj = careful_wave<LIT>(__builtin_amdgcn_readlane((int)(sel & h.idx_mask), L), segs, h.nV, p.segs + h.wall_off, h.S, bcast_d(npx, L),
                                            bcast_d(npy, L), bcast_d(d64.x, L), bcast_d(d64.y, L), (int)(threadIdx.x & 63));
        if ((int)(threadIdx.x & 63) == L) {
            d = dj;
            hit = col & (d < 10.0);
            dst[0] = obs_of(d);
        }
        tm &= tm - 1;
    }
    PC_STAMP(9)
    const uint64_t hit_mask = __builtin_amdgcn_ballot_w64(hit);
    const bool wall_hit = (unsigned)(hit_mask >> (lane & 32)) != 0u;                               // any() over the env's 32 lanes
    int gflag = gate_hit ? 1 : 0;
    gflag |= swap_pair(gflag);                                                                     // any() over the four ray groups
    gflag |= __builtin_amdgcn_update_dpp(0, gflag, 0x4e, 0xf, 0xf, false);                         // quad_perm [2, 3, 0, 1]
    gate_hit = gflag != 0;
    PC_STAMP(10)
    const bool destroyed = wall_hit | (h.start_collides != 0);
    // ---- bookkeeping (car_env.py:694-750): env_step_fast's
    double rw = L.fwd ? 0.01 : 0.0;
    const bool lap = gate_hit & (st.next == h.G - 1);
    rw = rw + (gate_hit ? 1.0 : 0.0);
    rw = rw + (lap ? 10.0 : 0.0);
    const int passed = st.passed + (gate_hit ? 1 : 0);
    const int next = gate_hit ? (lap ? 0 : st.next + 1) : st.next;
    const int time = st.time + 1;
    rw = rw + (destroyed ? -3.0 : 0.0);
    const bool trunc = !destroyed & (time >= 1000);
    const bool done = destroyed | trunc;
    reward_f = (float)(rw * reward_scale);
    term_f = destroyed ? 1.0f : 0.0f;
    trunc_f = trunc ? 1.0f : 0.0f;
    if constexpr (LIT) {      // the heading after the turn, read only now
        const double2 e1 = rot_tab[kid_new * rot_ld];
        cs1 = (f64x2){e1.x, e1.y};
    }
    if (xrow == nullptr) {
        if (g == 0 && part == 0) {
            using M = Math<std::conditional_t<LIT, double, float>>;
            lrow[0] = M::norm(npx, 1280.0);
            lrow[1] = M::norm(npy, 720.0);
            lrow[2] = M::norm(nvx, 10.0);
            lrow[3] = M::norm(nvy, 10.0);
            lrow[4] = (float)cs1.x;
            lrow[5] = (float)cs1.y;
        }
    } else {
        // every lane holds at most ONE entry of the row: parts 0 .. RPL - 1 their ray slot's (written above, final after the careful path),
        // parts 5 and 6 -- idle in the refinement -- the six kinematic entries (part 5: columns 0 .. 3, part 6: columns 4, 5)
        using M = Math<std::conditional_t<LIT, double, float>>;
        static_assert(RPL <= 5, "parts 5 and 6 carry the kinematic columns");
        const float k01 = g & 1 ? M::norm(npy, 720.0) : M::norm(npx, 1280.0), k23 = g & 1 ? M::norm(nvy, 10.0) : M::norm(nvx, 10.0);
        const float k45 = g & 1 ? (float)cs1.y : (float)cs1.x;
        const float kin = part == 5 ? (g & 2 ? k23 : k01) : k45;
        const bool mine_kin = (part == 5) | ((part == 6) & (g < 2));
        const int col = active ? 6 + min(g + G * slot, p.R - 1) : (part == 5 ? g : 4 + g);
        const float val = active ? obs_of(d) : kin;
        if (mine_kin) lrow[col] = val;
        if (active | mine_kin) write_pieces(xrow, col, val);
    }
    st.px = npx;
    st.py = npy;
    st.vx = nvx;
    st.vy = nvy;
    if constexpr (LIT) {
        st.k = kid_new;
        *hcar = cs1;
    } else {
        st.k += L.dk;
    }
    k72 = k72n;
    st.time = time;
    st.next = next;
    st.passed = passed;
    return done;
}

// Developer-only timing ablation of the persistent rollout kernels: a SEPARATE build (make ABLATE=n -> libppocar_ablate.so,
// never loaded by the product or the tests) compiled with -DPC_ABLATE=n skips the policy MFMAs (1), the env step (2), the draw
// (4), the gate casts (8), the float64 refinement (16), the sweep's flag minima (32), the whole sweep (64) or the copy-out (256).
// The shipped library is built with PC_ABLATE = 0 (env_math.hpp): there is no run-time switch that makes a kernel do less.
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
//         memory), env step = env_step_core.  MODE 1 / 2: one track per workgroup, A = 9, every table in LDS behind explicit LDS
//         pointers, env step = env_step_fast (2: with the 1/den table), observation rows copied out by the wave.
//         MODE 4 / 3 = 1 / 2 for batches whose tracks all have a padded wall chain of 28 vertices (big_track.json): only the
//         unrolled sweep is compiled in (built for 17 rays and the default arithmetic: ~1 % from the shorter kernel; the 33-ray
//         kernel schedules worse without the generic branch -- it spills -- and keeps it).
//         MODE 5 = 2 for batches whose tracks are all two equal chains of 13 OR of 9 vertices (big_track.json and track.json
//         mixed: BASELINE configs[4]): the chain-packed sweep for both lengths, chosen per workgroup.
//         MODE 6 = 1 for a batch of TWO such tracks INTERLEAVED inside the waves (both tracks' tables staged, the env step once per track of a wave).
//         LIT (the fast modes): the handle is PC_DTYPE_F64 -- env_step_fast's literal form; state with the float64 rotation and
//         its row of the rotation table (env_load<double>).
//         LGE = log2 of the lanes per env: 1 (default) = a wave owns 32 envs, two policy column tiles; 2 (round 5, fast modes at 17
//         rays) = 16 envs per wave, 4 lanes per env, ONE column tile, five ray slots per lane -- the form for 8193 .. 32768 envs,
//         where 32-env waves leave every SIMD with a single wave (9.8 us per step whatever the batch): two 16-env waves per SIMD
//         hide each other's latency.  More vector instructions per env (the per-wave work -- physics, draw, bookkeeping -- is shared
//         by 16 envs, not 32), so it loses again where 32-env waves already come in pairs (above 32768 envs).  Same bits.
template <int KS, int RPL, int PREC, int MODE, bool LIT = false, int LGE = 1>
__global__ __launch_bounds__(512) void rollout_kernel(const EnvParams<float> p, const float* __restrict__ image, const int A,
                                                      const int T, const double reward_scale, const uint64_t seed,
                                                      const uint64_t offset, const uint64_t* __restrict__ offset_dev,
                                                      float* __restrict__ obs_buf, float* __restrict__ act_buf,
                                                      float* __restrict__ rew_buf, float* __restrict__ val_buf,
                                                      float* __restrict__ term_buf, float* __restrict__ trunc_buf,
                                                      float* __restrict__ logprob_buf, float* __restrict__ next_obs,
                                                      float* __restrict__ next_term, float* __restrict__ next_trunc,
                                                      const int rden_lds, const int epw, const int vec_ok,
                                                      float* __restrict__ last_val, float* __restrict__ rew_sum) {
    constexpr int dbg = PC_ABLATE;  // 0 in the product build (see PC_ABLATE)
    constexpr int GE = 1 << LGE, EPWV = 64 / GE;      // lanes per env, envs per wave
    constexpr int HID = 256, NT = 2 * HID / 16, LD1 = pol_ld1(KS), ET = EPWV / 16;
    static_assert(LGE == 1 || (LGE == 2 && MODE != 0 && PREC != 0), "16 envs per wave: fast modes, split operand forms");
    constexpr int NG = pol_ng(KS), KB = pol_kb(KS);
    constexpr int IMG = PREC ? polx_image_dwords(PREC, NG) : pol_image_padded(KS);
    constexpr bool FAST = MODE != 0;
    // row stride of a wave's output tile [32 envs][LDO]: 16-byte rows (one ds_write_b128 per env tile, 16-byte reads in the draw);
    // the tile aliases the wave's 32 observation rows, so at 12 rays (18 floats per dense row) the stride is 16, else 20
    constexpr int LDO = (FAST && RPL == 6 && LGE == 1) ? 16 : 20;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* sW1 = lds;
    float* sB1 = PREC ? lds + polx_w1_dwords(PREC, NG) + polx_w2_dwords(PREC) : sW1 + 2 * HID * LD1;
    float* sW2 = sB1 + 2 * HID;
    float* sB2 = PREC ? sB1 + 512 : sW2 + NT * 4 * 64;
    const unsigned* sW1p = reinterpret_cast<const unsigned*>(lds);
    const unsigned* sW2p = sW1p + polx_w1_dwords(PREC ? PREC : 1, NG);
    const float* sW2c = sB2 + 16;                  // PREC 1: critic output weights [256]
    const int64_t N = p.N;
    // FAST: 6 + the ray count the lanes-per-env menu implies (12 / 17 / 33 rays: 6 / 9 / 17 slots on two lanes, 3 / 5 / 9 on four)
    constexpr int DC = LGE == 1 ? (RPL == 6 ? 18 : (RPL == 9 ? 23 : 39)) : (RPL == 3 ? 18 : (RPL == 5 ? 23 : 39));
    const int D = FAST ? DC : p.D;
    // observation of the step in flight, [256 envs][LDX]: FAST keeps the rows dense (LDX = D, exactly the rollout buffer's
    // layout: a wave's 32 rows are one contiguous block there and here)
    const int LDX = FAST ? D : 4 * KS + 1;
    float* sObs = lds + IMG;
    int* sAct = reinterpret_cast<int*>(sObs + 8 * EPWV * LDX);
    float* sTab = reinterpret_cast<float*>(sAct + K9_ACT_SLOTS);    // staged per-track tables (k9_fast_lds_floats: the host's size)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform by construction: everything derived from it lives in SGPRs)
    const int lc = lane & 15, lk = lane >> 4;
    policy_stage_image<IMG>(image, lds, tid);
    // FAST with a mixed-track batch: the host checked that every workgroup's envs lie on ONE track, whose tables it stages
    const int trk_wg = (FAST && p.track_id)
                           ? __builtin_amdgcn_readfirstlane((int)p.track_id[min((int64_t)blockIdx.x * epw, p.N - 1)]) : 0;
    const TrackHdr h0 = cload(p.hdr + trk_wg);
    EnvParams<float> q = p;
    FastTabs ft = {};
    static_assert(!LIT || FAST, "the literal form: the fast modes (LDS tables)");
    // MODE 6: a batch of TWO tracks interleaved inside the waves (car_env.py:621-628: every env may sit on its own track; configs[4]'s
    // `track_id = i & 1` variant).  Both tracks' gather tables are staged, track 1's block TS6 floats behind track 0's (where the other
    // fast modes keep the 1/den table: the sweep forms 1 / den itself), and the env step runs once per track present in the wave, each
    // pass with that track's header and tables -- wave-uniform, as in every fast mode.
    constexpr int TS6 = (ft_floats(false, RPL != 17) + 3) & ~3;
    if constexpr (MODE == 6 || MODE == 7) {
        ft = stage_fast_tables<false, RPL != 17, LIT>(p, cload(p.hdr), 0, sTab, tid, 512);
        (void)stage_fast_tables<false, RPL != 17, LIT>(p, cload(p.hdr + 1), 1, sTab + TS6, tid, 512);
    } else if constexpr (FAST) ft = stage_fast_tables<false, RPL != 17, LIT>(p, h0, trk_wg, sTab, tid, 512);
    else q = stage_tables(p, sTab, tid, 512);
    // the track's 1/den table, when the host found room for it (rden_lds != 0; sized for the batch's largest track)
    float* sRden = sTab + (FAST ? ft_floats(false, RPL != 17) : TAB_FLOATS);
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(p.rden + h0.rden_off);
        const int n4 = rden_lds ? 361 * h0.nV / 4 : 0;     // nV is a multiple of 4
        if constexpr (MODE == 3 || MODE == 5) {
            // the two-loop sweep (wall_sweep_loops) reads a row in the order (0, L, 1, L + 1, ...): entry 2 i = vertex i, entry
            // 2 i + 1 = vertex L + i (L = 13: rows of 28, the last two entries are the padding vertices'; L = 9: rows of 20, two padding entries)
            const float* g = p.rden + h0.rden_off;
            const int L = h0.brk2, nv = h0.nV;
            for (int i = tid; i < 4 * n4; i += 512) {
                const int row = i / nv, e = i - nv * row;
                const int k = e < 2 * L ? (e >> 1) + ((e & 1) ? L : 0) : e;
                sRden[i] = g[nv * row + k];
            }
        } else {
            for (int i = tid; i < n4; i += 512) reinterpret_cast<f32x4*>(sRden)[i] = src[i];
        }
    }
    const lds_cfp rdl = (lds_cfp)sRden;

    // this wave's 32 envs: local rows [pbase, pbase + 32); env-step identity: 2 lanes per env
    const int pbase = wave * EPWV;
    const int el = pbase + (lane >> LGE), g = lane & (GE - 1);
    // epw = envs per workgroup: 256 (all 8 waves) or 128 (waves 4..7 only help to stage LDS and leave: at <= 32768 envs
    // that doubles the workgroups, one wave per SIMD on all 256 CUs instead of two on half of them)
    const int64_t e_wave = (int64_t)blockIdx.x * epw + pbase;      // first env of this wave
    // MODE 7: two tracks interleaved env by env with every aligned block of 2 x EPWV envs split evenly (the host checked; `i & 1` is the
    // plainest case): the block's two waves DE-INTERLEAVE it -- wave 2k takes its track-0 envs, wave 2k + 1 its track-1 envs, slot s the
    // s-th of them -- so that every wave steps ONE track: one pass of the table-driven step with that track's header and tables, careful
    // rays as whole-wave jobs, where mode 6 runs a pass per track with the other track's lanes masked off.  Only the addresses change:
    // an env's rows, state and Philox counter are the env's (same bits as every other form); the observation rows leave by row.
    const int wtrk = MODE == 7 ? (wave & 1) : 0;
    const int64_t e_block = (int64_t)blockIdx.x * epw + (wave >> 1) * (2 * EPWV);
    int lj = lane >> LGE;      // MODE 7: this lane's env inside the block
    if constexpr (MODE == 7) {
        const int64_t ei = e_block + lane;
        const bool is1 = lane < 2 * EPWV && ei < N && p.track_id[ei] != 0;
        const bool is0 = lane < 2 * EPWV && ei < N && p.track_id[ei] == 0;
        uint64_t m = wtrk ? __builtin_amdgcn_ballot_w64(is1) : __builtin_amdgcn_ballot_w64(is0);
        for (int k = lane >> LGE; k > 0; --k) m &= m - 1;      // drop the slot's predecessors (once per launch)
        lj = m ? __builtin_ctzll(m) : 2 * EPWV;                // (an unbalanced tail cannot happen: the host's check)
        if ((lane & (GE - 1)) == 0) sAct[pbase + (lane >> LGE)] = lj;
    }
    const int64_t e_env = MODE == 7 ? e_block + lj : e_wave + (lane >> LGE);
    const bool e_valid = e_env < N && (MODE != 7 || lj < 2 * EPWV);
    using StateT = std::conditional_t<LIT, double, float>;
    const EnvParams<StateT> ps = p.template as<StateT>();
    EnvRegs st = {};
    if (e_valid) st = env_load<StateT>(ps, e_env);
    // mixed-track batch: this wave's envs share one track (the host checked every aligned block of 32 envs)
    const int trk = p.track_id ? (int)p.track_id[e_valid ? e_env : N - 1] : 0;
    for (int f = g; f < (FAST ? D : 4 * KS); f += GE) sObs[el * LDX + f] = (e_valid && f < D) ? next_obs[e_env * D + f] : 0.0f;
    // this wave's output tile [32 envs][LDO] lives in its own observation rows: they are dead from the policy pass's
    // operand load until the env step stores the next observation (32 * LDX >= 32 * LDO floats: D >= 17 on the host's menu)
    static_assert(4 * KS + 1 >= 20, "the output tile must fit the wave's observation rows");
    static_assert(LGE == 1 || DC >= LDO, "the output tile [16 envs][LDO] must fit the wave's 16 observation rows");
    float* myOut = sObs + wave * EPWV * LDX;
    const uint64_t off0 = offset + (offset_dev ? *offset_dev : 0);
    PhiloxBlock rnd = {};  // the sampling lanes' current Philox block (4 steps' draws)
    // FAST: per-lane invariants of the env step.  Ray slot s of lane g is ray min(g + 2 s, R - 1): the odd slot that 17 or
    // 33 rays leave over on lane 1 repeats that lane pair's last ray (same value, same address) instead of being masked.
    int gq[2] = {0, 0}, k72 = 0;
    FastLane fl = {};
    if constexpr (FAST) {
        fl = fast_lane<RPL, GE>(p, ft, g, sObs + el * LDX);
        gq[0] = (int)(size_t)ft.dir + 16 * g * p.q * p.step_deg;            // Car.get_passed_gate's rays j * (n // 4), j = g and (two lanes per env) g + 2,
        gq[1] = (int)(size_t)ft.dir + 16 * (g + 2) * p.q * p.step_deg;      // as byte addresses into the direction table
        k72 = Math<float>::mod72(st.k);
        if constexpr (MODE == 7) {      // the wave's track: its block of tables, TS6 floats behind track 0's
            const int by = 4 * TS6 * wtrk;
            fl.rs0 += by;
            fl.rs_last += by;
            gq[0] += by;
            gq[1] += by;
        }
    }
    const TrackHdr hw = MODE == 7 ? cload(p.hdr + wtrk) : h0;      // MODE 7: the wave's own track
    f64x2 hcar = {1.0, 0.0};      // LIT: (cos, sin) of the env's current rotation (row st.k of the rotation table)
    const int rot_off_l = (LIT && MODE == 6) ? (trk ? cload(p.hdr + 1).rot_off : cload(p.hdr).rot_off) : hw.rot_off;     // the lane's track's rotation table
    if constexpr (LIT) {
        // the lattice index of the heading: rot = start_rot after k turns of +-5.0 (each sum rounded; the quotient is within 1e-10 of k)
        // (MODE 6: the lane's own track -- two tracks -- not the workgroup's)
        const double start_rot_l = MODE == 6 ? (trk ? cload(p.hdr + 1).start_rot : cload(p.hdr).start_rot) : hw.start_rot;
        k72 = Math<float>::mod72((int)__builtin_rint((st.rot - start_rot_l) / 5.0));
        if (e_valid) {
            const double2 e0 = p.dirtab64[rot_off_l + st.k * (p.R + 2)];
            hcar = (f64x2){e0.x, e0.y};
        }
        st.rot = 0.0;     // (not kept: the rotation is the row's last entry, read again when the state is stored)
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
    if constexpr (LIT) asm volatile("" : "+v"(hcar));

    // pc_rollout_ex: one more policy pass after the last env step gives the critic's value of the FINAL observation (the
    // bootstrap value of Buffer.calculate_advantages, train.py:200) -- tail iteration t == T: no draw stored, no env step --,
    // and every env's reward total rides along in a register (train.py:272's average reward without re-reading rew_buf).
    const int TT = last_val ? T + 1 : T;
    float rsum = 0.0f;
    int act_reg = 8;     // the action drawn for this lane pair's env (the pair draw leaves it in both lanes)
    // FAST: the per-env rows a lane stores every step as per-lane pointers that advance by one buffer row per step, instead of
    // base + (t N + e) formed from kernel-argument SGPRs the kernel does not have (spilled to vector lanes: a v_readlane pair per use;
    // rollout_small_kernel: -4 % of its launch; here -1 %, profiles/r6_ab_k9_ptrs.log).  Twelve more registers: the 12- and 17-ray kernels
    // have them (the target kernel: 238 -> 252 VGPRs, 57 -> 40 spilled SGPRs), the 33-ray kernels do not.
    constexpr bool PTRS = FAST && RPL <= 9 && MODE != 6 && MODE != 7;      // (the two-track form spends its registers on the second pass's header; mode 7 addresses by env)
    float* pa_act = act_buf + e_env;
    float* pa_lp = logprob_buf + e_env;
    float* pa_val = val_buf + e_env;
    float* pa_rew = rew_buf + e_env;
    float* pa_term = term_buf + N + e_env;      // flags that precede obs t + 1 (train.py:176-177,195)
    float* pa_trunc = trunc_buf + N + e_env;
#pragma unroll 1
    for (int t = 0; t < TT; ++t) {
        const bool tail = t == T;      // (uniform)
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
                        const float raw = sObs[(pbase + 16 * et + lc) * LDX + f];
                        x[et][ks] = (!FAST || f < D) ? raw : 0.0f;
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
                            // K padding (f >= D): the load is unconditional -- it stays inside the workgroup's LDS (a few floats
                            // into the next row / the action slots) -- and its value discarded by a select: a conditional
                            // load compiles into one exec-masked branch per feature, each with its own s_waitcnt lgkmcnt(0),
                            // i.e. 16 LDS round trips in series per step
                            const float raw = sObs[(pbase + 16 * et + lc) * LDX + f];
                            v[j] = f < D ? raw : 0.0f;
                            if constexpr (PREC == 2) v[j] = clamp_h(v[j] * PolScale<PREC>::sx);   // the observations' scaled domain
                        }
                        x[et][kb] = split8<PREC>(v);
                    }
                }
                float val[ET] = {};
                __builtin_amdgcn_s_setprio(0);
                PC_STAMP(1)
                if (!(dbg & 1)) policy_pass16<PREC, KB, ET>(sW1p, sW2p, sB1, sW2c, 0, NT / 2, x, out, val, lc, lk);
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
            for (int et = 0; et < ET; ++et) *reinterpret_cast<f32x4*>(myOut + (16 * et + lc) * LDO + 4 * lk) = out[et];
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // the tile is written and read by this wave only
            __builtin_amdgcn_wave_barrier();
            const uint64_t o = off0 + (uint64_t)t;
            if (FAST || A == 9) {     // Discrete(9): the env's two lanes draw together (policy_tail_pair); the action stays with them
                if (e_valid) {
                    float w[5];
                    // (four lanes per env: lanes 2, 3 repeat lanes 0, 1 -- the same values, the action in all four)
                    pair_outputs<LDO>(myOut, lane >> LGE, g & 1, PolScale<PREC>::so_inv, sB2, w);
                    float lp, val;
                    if (t == 0 || (o & 3) == 0) {   // uniform: ten rounds per 4 steps
                        uint64_t ctr = (uint64_t)e_env;             // (opaque: the first round's products of the lane's counter are formed here,
                        asm volatile("" : "+v"(ctr));               // not hoisted into four registers that live through all T steps)
                        rnd = philox_block(seed, o >> 2, ctr);
                    }
                    if constexpr (PC_ABLATE & 4) { act_reg = (int)(o & 7); lp = w[0]; val = w[4]; }
                    else policy_tail_pair(w, g & 1, philox_word_uniform(rnd, (unsigned)(o & 3)), act_reg, lp, val);
                    if constexpr (!FAST) { if (g == 0) sAct[el] = act_reg; }
                    if (g == 0) {
                        if (tail) {   // (once per launch: the address is formed here, not kept in two registers for T steps -- at 33 rays they were spilled)
                            int64_t ee = e_env;
                            asm volatile("" : "+v"(ee));
                            last_val[ee] = val;
                        } else if constexpr (PTRS) {
                            *pa_act = (float)act_reg;          // stored as float32 like the reference (buffer.py:13)
                            *pa_lp = lp;
                            *pa_val = val;
                        } else {
                            const int64_t row = (int64_t)t * N + e_env;
                            act_buf[row] = (float)act_reg;     // stored as float32 like the reference (buffer.py:13)
                            logprob_buf[row] = lp;
                            val_buf[row] = val;
                        }
                    }
                }
                if constexpr (PTRS) { pa_act += N; pa_lp += N; pa_val += N; }
            } else {
                const int64_t e = e_wave + lane;
                if (lane < 32 && e < N) {
                    float v[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(myOut[lane * LDO + i], PolScale<PREC>::so_inv, sB2[i]);   // outputs back from their scaled domain
                    int act;
                    float lp, val;
                    if (t == 0 || (o & 3) == 0) rnd = philox_block(seed, o >> 2, (uint64_t)e);  // uniform: ten rounds per 4 steps
                    policy_tail(v, A, philox_word_uniform(rnd, (unsigned)(o & 3)), act, lp, val, nullptr);
                    sAct[pbase + lane] = act;
                    if (tail) {
                        last_val[e] = val;
                    } else {
                        const int64_t row = (int64_t)t * N + e;
                        act_buf[row] = (float)act;     // stored as float32 like the reference (buffer.py:13)
                        logprob_buf[row] = lp;
                        val_buf[row] = val;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (tail) break;
        const bool last = t + 1 == T;
        PC_STAMP(3)
        if constexpr (FAST) {
            if (!(dbg & 2)) {
                __builtin_amdgcn_s_setprio(2);
                // ---------------- E(t)
                float rw, tf, cf;
                const int a = e_valid ? act_reg : 8;
                // one env step of the lanes active here, on the track (hh, ff): CarEnv.step, then gymnasium 0.29.1's same-step auto-reset
                const auto step_on = [&](const TrackHdr& hh, const FastTabs& ff, const FastLane& fll, const int (&gg)[2]) {
                    // (COOP: the careful slots as whole-wave jobs -- every lane is active here; not in the two-track form, whose passes are exec-masked)
                    const bool done = env_step_fast<RPL, MODE == 2 || MODE == 3 || MODE == 5, LGE, 1, ((MODE == 5 || MODE == 6 || MODE == 7) ? 5 : (MODE >= 3 ? 7 : 0)), RPL != 17, LIT, MODE != 6>(
                        p, hh, ff, fll, gg, g, st, k72, a, reward_scale, lrow, rw, tf, cf, t, lane, wave, 0, nullptr, true, nullptr, &hcar);
                    rsum += rw;
                    PC_STAMP(6)
                    if (__builtin_amdgcn_ballot_w64(done) != 0) {   // wave-uniform: ~1.5 % of env steps end an episode
                        if (done) {
                            // the reset observation's entries f = g, g + 2, ...: all reads issued, then the writes (rolled, every
                            // entry is an LDS round trip in series -- and some env of a wave finishes in a third of the steps of a
                            // young policy)
                            float ro[(DC + GE - 1) / GE];
#pragma unroll
                            for (int j = 0; j < (DC + GE - 1) / GE; ++j) ro[j] = ff.reset[g + GE * j];     // (the table has 40 slots: in bounds)
#pragma unroll
                            for (int j = 0; j < (DC + GE - 1) / GE; ++j)
                                if (g + GE * j < DC) lrow[g + GE * j] = ro[j];
                            env_reset_fast(hh, st, k72);
                            if constexpr (LIT) {
                                const double2 e0 = p.dirtab64[hh.rot_off];      // row 0 = start_rot
                                hcar = (f64x2){e0.x, e0.y};
                            }
                        }
                    }
                };
                if constexpr (MODE == 6) {
                    uint64_t todo = __builtin_amdgcn_ballot_w64(true);      // once per track present in the wave (K1's waterfall)
                    do {
                        const int cur = __builtin_amdgcn_readlane(trk, __builtin_ctzll(todo));
                        const bool match = trk == cur;
                        if (match) {
                            const TrackHdr hb = cload(p.hdr + cur);
                            const int by = 4 * TS6 * cur;                        // (two tracks: cur is 0 or 1)
                            FastLane flb = fl;
                            flb.rs0 += by;
                            flb.rs_last += by;
                            const int gqb[2] = {gq[0] + by, gq[1] + by};
                            step_on(hb, ft_shift(ft, TS6 * cur), flb, gqb);
                        }
                        todo &= ~__builtin_amdgcn_ballot_w64(match);
                    } while (todo);
                } else if constexpr (MODE == 7) {
                    step_on(hw, ft_shift(ft, TS6 * wtrk), fl, gq);
                } else {
                    step_on(h0, ft, fl, gq);
                }
                if constexpr (PTRS) {
                    if (last) {        // (uniform, once per launch)
                        pa_term = next_term + e_env;
                        pa_trunc = next_trunc + e_env;
                    }
                    if (g == 0 && e_valid) {
                        *pa_rew = rw;
                        *pa_term = tf;
                        *pa_trunc = cf;
                    }
                    pa_rew += N; pa_term += N; pa_trunc += N;
                } else if (g == 0 && e_valid) {
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
                const int n_rows = left >= EPWV ? EPWV : (int)left;
                const float* srcl = sObs + pbase * LDX;
                if constexpr (PC_ABLATE & 256) {
                } else if constexpr (MODE == 7) {
                    // the wave's rows -> rows e_block + lj(slot) of the buffer (the slots' envs: sAct, written at the kernel's start)
                    float* bg = (last ? next_obs : obs_buf + (int64_t)(t + 1) * N * D) + e_block * D;
#pragma unroll
                    for (int j = 0; j < (EPWV * DC + 63) / 64; ++j) {
                        const int i = lane + 64 * j;
                        if (64 * j + 63 < EPWV * DC || i < EPWV * DC) {
                            const int r = i / DC, c = i - r * DC;
                            const int lr = sAct[pbase + r];
                            if (e_block + lr < N && lr < 2 * EPWV) bg[lr * DC + c] = srcl[i];
                        }
                    }
                } else if (vec_ok && n_rows == EPWV) {
                    constexpr int NF4 = EPWV / 4 * DC;                  // the wave's rows as float4s: 8 D (32 envs: 3 or 5 stores per lane) or 4 D
#pragma unroll
                    for (int j = 0; j < (NF4 + 63) / 64; ++j) {
                        const int i = lane + 64 * j;
                        if (64 * j + 63 < NF4 || i < NF4) reinterpret_cast<f32x4*>(dstg)[i] = reinterpret_cast<const f32x4*>(srcl)[i];
                    }
                } else {
                    // (the unaligned caller's path: its addresses are formed here from opaque copies, not hoisted into registers that
                    // live through all T steps -- at 33 rays those were spilled to scratch)
                    int64_t ew = e_wave;
                    int pb = pbase, i0 = lane;
                    asm volatile("" : "+v"(ew), "+v"(pb), "+v"(i0));
                    float* dg = (last ? next_obs : obs_buf + (int64_t)(t + 1) * N * D) + ew * D;
                    const float* sl = sObs + pb * LDX;
                    for (int i = i0; i < n_rows * D; i += 64) dg[i] = sl[i];
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
                env_step_core<float, RPL, 1, true, true>(q, trk, g, 1, st, (int64_t)sAct[el], reward_scale, orow, nullptr, sObs + el * LDX, rw,
                                                   term, trunc, passed, 0, nullptr, rdl);
            else {
                // The step runs once per DISTINCT TRACK ID among the wave's envs (K1's waterfall): one pass for a single-track batch and
                // for mixed batches in blocks of 32 envs; with tracks INTERLEAVED inside the wave (car_env.py:621-628 lets every env sit
                // on its own track; configs[4]'s `track_id = i & 1` variant) one pass per track, so that a pass's track header and wall
                // chain stay wave-uniform (scalar loads).  The loop is driven by a ballot of the lanes still to do.
                uint64_t todo = __builtin_amdgcn_ballot_w64(true);
                do {
                    const int cur = __builtin_amdgcn_readlane(trk, __builtin_ctzll(todo));
                    const bool match = trk == cur;
                    if (match)
                        env_step_core<float, RPL, 1, false, true>(q, cur, g, 1, st, (int64_t)sAct[el], reward_scale, orow, nullptr, sObs + el * LDX, rw,
                                                                  term, trunc, passed);
                    todo &= ~__builtin_amdgcn_ballot_w64(match);
                } while (todo);
            }
            rsum += rw;
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
    if (e_valid && g == 0) {
        if constexpr (LIT) st.rot = p.dirtab64[rot_off_l + st.k * (p.R + 2) + p.R + 1].x;
        env_store<StateT>(ps, e_env, st);
        if (rew_sum) rew_sum[e_env] = rsum;
    }
}

// K9d: the persistent rollout of the BIT-EXACT dtype (PC_DTYPE_F64: float64 throughout in the reference's operation order, glibc's
// cos / sin by lookup -- env_step_core<double>, the very function env_step_kernel<double> runs).  K9's generic decomposition: a
// workgroup = 8 independent waves of 32 envs, 2 lanes per env, the policy pass in the split-operand arithmetic PREC on the matrix
// cores, the pair draw, then CarEnv.step for the wave's own envs with the observation row written straight into the rollout buffer and
// into the wave's LDS rows (the next policy pass reads it there); no barrier after staging.  Nothing of the track is staged: the
// float64 wall records come through scalar loads, the direction hash table through the vector cache (both L2-resident).
// Bit-identical to T x (policy_kernel<KS, false, PREC>; env_step_kernel<double>): same arithmetic, same Philox counters.
// (train.py:173-195 with the env of car_env.py:693-760 in its own float64)
// SEL (every track of the handle inside the selector's limits: host-checked): the env step is the per-step kernel's selector step (env_step_core<double, ..., SEL>: the
// float32 sweep over the chain in global memory, then the literal cast) -- what a track too long for the LDS tables of K9's literal
// kernels gets; without it every (ray, wall) pair is tested in float64 (the FILTER form).  Two instantiations: both paths in one
// kernel spilled 43 registers.
template <int KS, int RPL, int PREC, bool SEL = false>
__global__ __launch_bounds__(512) void rollout_f64_kernel(const EnvParams<double> p, const float* __restrict__ image, const int A,
                                                          const int T, const double reward_scale, const uint64_t seed,
                                                          const uint64_t offset, const uint64_t* __restrict__ offset_dev,
                                                          float* __restrict__ obs_buf, float* __restrict__ act_buf,
                                                          float* __restrict__ rew_buf, float* __restrict__ val_buf,
                                                          float* __restrict__ term_buf, float* __restrict__ trunc_buf,
                                                          float* __restrict__ logprob_buf, float* __restrict__ next_obs,
                                                          float* __restrict__ next_term, float* __restrict__ next_trunc,
                                                          const int epw, float* __restrict__ last_val, float* __restrict__ rew_sum) {
    static_assert(PREC != 0, "split-operand policy forms");
    constexpr int HID = 256, NT = 2 * HID / 16, ET = 2, LDO = 20;
    constexpr int NG = pol_ng(KS), KB = pol_kb(KS);
    constexpr int IMG = polx_image_dwords(PREC, NG);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* sB1 = lds + polx_w1_dwords(PREC, NG) + polx_w2_dwords(PREC);
    float* sB2 = sB1 + 512;
    const unsigned* sW1p = reinterpret_cast<const unsigned*>(lds);
    const unsigned* sW2p = sW1p + polx_w1_dwords(PREC, NG);
    const float* sW2c = sB2 + 16;
    const int64_t N = p.N;
    const int D = p.D;
    constexpr int LDX = 4 * KS + 1;                 // observation of the step in flight, [256 envs][LDX]
    float* sObs = lds + IMG;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lc = lane & 15, lk = lane >> 4;
    policy_stage_image<IMG>(image, lds, tid);
    const int pbase = wave * 32;
    const int el = pbase + (lane >> 1), g = lane & 1;
    const int64_t e_wave = (int64_t)blockIdx.x * epw + pbase;
    const int64_t e_env = e_wave + (lane >> 1);
    const bool e_valid = e_env < N;
    EnvRegs st = {};
    if (e_valid) st = env_load<double>(p, e_env);
    const int trk = p.track_id ? (int)p.track_id[e_valid ? e_env : N - 1] : 0;     // (per env: a wave may hold several tracks, see E(t))
    for (int f = g; f < 4 * KS; f += 2) sObs[el * LDX + f] = (e_valid && f < D) ? next_obs[e_env * D + f] : 0.0f;
    static_assert(4 * KS + 1 >= 20, "the output tile must fit the wave's observation rows");
    float* myOut = sObs + wave * 32 * LDX;          // the wave's output tile lives in its own observation rows (see rollout_kernel)
    const uint64_t off0 = offset + (offset_dev ? *offset_dev : 0);
    PhiloxBlock rnd = {};
    __syncthreads();  // the weight image is in place; from here on the waves never synchronise again
    if (pbase >= epw) return;
    asm volatile("" : "+v"(st.px), "+v"(st.py), "+v"(st.vx), "+v"(st.vy), "+v"(st.rot), "+v"(st.time), "+v"(st.next), "+v"(st.passed));
    const int TT = last_val ? T + 1 : T;   // tail iteration t == T: the final observation's value only (see rollout_kernel)
    float rsum = 0.0f;
    int act_reg = 8;
#pragma unroll 1
    for (int t = 0; t < TT; ++t) {
        const bool tail = t == T;      // (uniform)
        {
            // ---------------- P(t): Agent.get_action_and_value (model.py:34-41)
            {
                f32x4 out[ET];
#pragma unroll
                for (int et = 0; et < ET; ++et) out[et] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
                Pieces<PREC> x[ET][KB];
#pragma unroll
                for (int et = 0; et < ET; ++et) {
#pragma unroll
                    for (int kb = 0; kb < KB; ++kb) {
                        float v[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int f = 8 * (4 * kb + lk) + j;
                            const float raw = sObs[(pbase + 16 * et + lc) * LDX + (f < LDX ? f : 0)];
                            v[j] = f < D ? raw : 0.0f;
                            if constexpr (PREC == 2) v[j] = clamp_h(v[j] * PolScale<PREC>::sx);
                        }
                        x[et][kb] = split8<PREC>(v);
                    }
                }
                float val[ET] = {0.0f, 0.0f};
                policy_pass16<PREC, KB>(sW1p, sW2p, sB1, sW2c, 0, NT / 2, x, out, val, lc, lk);
#pragma unroll
                for (int et = 0; et < ET; ++et) {
                    float tv = val[et];
                    tv += __shfl_xor(tv, 16, 64);
                    tv += __shfl_xor(tv, 32, 64);
                    if (A >> 2 == lk) out[et][A & 3] += tv;
                }
#pragma unroll
                for (int et = 0; et < ET; ++et) *reinterpret_cast<f32x4*>(myOut + (16 * et + lc) * LDO + 4 * lk) = out[et];
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint64_t o = off0 + (uint64_t)t;
            if (e_valid) {      // Discrete(9): the env's two lanes draw together (policy_tail_pair, as policy_kernel does for A == 9)
                float w[5];
                pair_outputs<LDO>(myOut, lane >> 1, g, PolScale<PREC>::so_inv, sB2, w);
                float lp, vv;
                if (t == 0 || (o & 3) == 0) {
                    uint64_t ctr = (uint64_t)e_env;
                    asm volatile("" : "+v"(ctr));
                    rnd = philox_block(seed, o >> 2, ctr);
                }
                policy_tail_pair(w, g, philox_word_uniform(rnd, (unsigned)(o & 3)), act_reg, lp, vv);
                if (g == 0) {
                    if (tail) {
                        last_val[e_env] = vv;
                    } else {
                        const int64_t row = (int64_t)t * N + e_env;
                        act_buf[row] = (float)act_reg;     // stored as float32 like the reference (buffer.py:13)
                        logprob_buf[row] = lp;
                        val_buf[row] = vv;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (tail) break;
        const bool last = t + 1 == T;
        if (e_valid) {
            // ---------------- E(t): envs.step (train.py:185) in the reference's float64
            float* orow = last ? next_obs + e_env * D : obs_buf + ((int64_t)(t + 1) * N + e_env) * D;
            float rw;
            bool term, trunc;
            int passed;
            uint64_t todo = __builtin_amdgcn_ballot_w64(true);      // once per distinct track id among the wave's envs (see rollout_kernel's generic mode)
            do {
                const int cur = __builtin_amdgcn_readlane(trk, __builtin_ctzll(todo));
                const bool match = trk == cur;
                if (match)
                    env_step_core<double, RPL, 1, false, true, SEL ? 2 : 0>(p, cur, g, 1, st, (int64_t)act_reg, reward_scale, orow, nullptr, sObs + el * LDX, rw,
                                                                     term, trunc, passed);
                todo &= ~__builtin_amdgcn_ballot_w64(match);
            } while (todo);
            rsum += rw;
            if (g == 0) {
                rew_buf[(int64_t)t * N + e_env] = rw;
                float* tr = last ? next_term : term_buf + (int64_t)(t + 1) * N;    // flags that precede obs t+1 (train.py:176-177,195)
                float* tc = last ? next_trunc : trunc_buf + (int64_t)(t + 1) * N;
                tr[e_env] = term ? 1.0f : 0.0f;
                tc[e_env] = trunc ? 1.0f : 0.0f;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // obs rows in LDS are this wave's own
        __builtin_amdgcn_wave_barrier();
    }
    if (e_valid && g == 0) {
        env_store<double>(p, e_env, st);
        if (rew_sum) rew_sum[e_env] = rsum;
    }
}

// K9s: the same persistent rollout for SMALL batches (n_envs < ~32 k): a workgroup owns only 32 envs, so that
// n_envs / 32 workgroups fill the chip.  Per step: the 8 waves split the policy's hidden tiles exactly as
// policy_kernel<SPLIT> does (partial output tiles summed through LDS, same order: bit-identical), wave 0 draws the
// 32 actions, then all 512 lanes run the env step with 16 lanes per env.  Three workgroup barriers per step.
// MODE as in rollout_kernel: 0 = generic tables, env_step_core; 1 / 2 = single track, A = 9, every table in LDS behind LDS
// pointers, env_step_fast (2: with the 1/den table), dense observation rows copied out by three waves in 16-byte stores.
// LIT: the handle is PC_DTYPE_F64 -- env_step_wave's / env_step_fast's literal form, state with the rotation's row of the rotation table.
template <int KS, int RPL, int PREC, int MODE, int EPW, bool LIT = false>
__global__ __launch_bounds__(512) void rollout_small_kernel(const EnvParams<float> p, const float* __restrict__ image, const int A,
                                                            const int T, const double reward_scale, const uint64_t seed,
                                                            const uint64_t offset, const uint64_t* __restrict__ offset_dev,
                                                            float* __restrict__ obs_buf, float* __restrict__ act_buf,
                                                            float* __restrict__ rew_buf, float* __restrict__ val_buf,
                                                            float* __restrict__ term_buf, float* __restrict__ trunc_buf,
                                                            float* __restrict__ logprob_buf, float* __restrict__ next_obs,
                                                            float* __restrict__ next_term, float* __restrict__ next_trunc,
                                                            const int rden_lds, const int vec_ok,
                                                            float* __restrict__ last_val, float* __restrict__ rew_sum) {
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
    int* sHit = sAct + 32;                                 // [EPW envs][PARTS]: the sweep parts' collision / gate verdicts (128 words)
    // XPRE (16 envs per workgroup, fp16 x 2): the observation rows ALSO as pre-split policy operands, [16 envs][2 pieces][32 features] halves
    // (2 KB; the host sizes the launch for the 32-env form's partial tiles and rows, of which this form uses half: in bounds)
    constexpr bool XPRE = EPW == 16 && PREC == 2;
    static_assert(!XPRE || 8 * 16 * LDO + 16 * 40 + 512 <= 8 * 32 * LDO + 32 * 40, "the pre-split rows fit the slack of the 32-env launch size");
    float* sXp = reinterpret_cast<float*>(sHit + 128);
    float* sTab = sXp + (XPRE ? 512 : 0);                  // staged per-track tables
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lc = lane & 15, lk = lane >> 4;
    policy_stage_image<IMG>(image, lds, tid);
    // FAST with a mixed-track batch: the workgroup's envs lie on ONE track (every aligned block of 32 does), whose tables it stages
    const int trk_wg = (FAST && p.track_id)
                           ? __builtin_amdgcn_readfirstlane((int)p.track_id[min((int64_t)blockIdx.x * EPW, p.N - 1)]) : 0;
    const TrackHdr h0 = cload(p.hdr + trk_wg);
    EnvParams<float> q = p;
    FastTabs ft = {};
    static_assert(!LIT || FAST, "the literal form: the fast modes");
    if constexpr (FAST) ft = stage_fast_tables<true, true, LIT>(p, h0, trk_wg, sTab, tid, 512);
    else q = stage_tables(p, sTab, tid, 512);
    // the track's 1/den table, when the host found room for it (rden_lds != 0; sized for the batch's largest track)
    float* sRden = sTab + (FAST ? ft_floats(true, true) : TAB_FLOATS);
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(p.rden + h0.rden_off);
        const int n4 = rden_lds ? 361 * h0.nV / 4 : 0;     // nV is a multiple of 4
        for (int i = tid; i < n4; i += 512) reinterpret_cast<f32x4*>(sRden)[i] = src[i];
    }
    const lds_cfp rdl = (lds_cfp)sRden;

    // env step, 32 envs per workgroup: wave w sweeps part (w % PARTS) of the wall vertices for 16 envs, 4 lanes (ray groups) per env;
    // 16 envs per workgroup (WOWN): wave w owns envs 2 w and 2 w + 1 outright, 32 lanes per env = 8 sweep parts x 4 ray groups
    // (env_step_wave) -- the two envs it also draws the actions of
    constexpr int PARTS = 128 / EPW;
    constexpr bool WOWN = EPW == 16;
    const int part = WOWN ? (lane >> 2) & 7 : __builtin_amdgcn_readfirstlane(wave % PARTS);
    const int el = WOWN ? 2 * wave + (lane >> 5) : (wave / PARTS) * 16 + (lane >> 2);
    const int g = lane & 3;
    constexpr int EXS = EPW == 16 ? PARTS * (DC - 6) : PARTS * 34;   // floats per env: [rays][PARTS]
    static_assert(EPW * EXS <= 8 * EPW * LDO, "the exchange area aliases the partial output tiles");
    float* exch = sOut + el * EXS;                 // [rays][PARTS] of this env; aliases the partial output tiles (idle now)
    const int64_t e_wg = (int64_t)blockIdx.x * EPW;
    const int64_t e_env = e_wg + el;
    const bool e_valid = e_env < N;
    using StateT = std::conditional_t<LIT, double, float>;
    const EnvParams<StateT> ps = p.template as<StateT>();
    EnvRegs st = {};
    if (e_valid) st = env_load<StateT>(ps, e_env);
    // mixed-track batch: this wave's envs share one track (the host checked every aligned block of 32 envs)
    const int trk = p.track_id ? (int)p.track_id[e_valid ? e_env : N - 1] : 0;
    for (int f = g + 4 * part; f < (FAST ? D : 4 * KS); f += 4 * PARTS) sObs[el * LDX + f] = (e_valid && f < D) ? next_obs[e_env * D + f] : 0.0f;
    const lds_hp xrow = (lds_hp)sXp + el * 64;             // XPRE: this env's [2][32] halves
    if constexpr (XPRE) {                                  // every lane of an env: column g + 4 part of 32 (the K padding = 0, once and for all)
        const int f = g + 4 * part;
        write_pieces(xrow, f, (e_valid && f < D) ? next_obs[e_env * D + f] : 0.0f);
    }
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
    f64x2 hcar = {1.0, 0.0};      // LIT: (cos, sin) of the env's current rotation (see rollout_kernel)
    if constexpr (LIT) {
        k72 = Math<float>::mod72((int)__builtin_rint((st.rot - h0.start_rot) / 5.0));
        if (e_valid) {
            const double2 e0 = p.dirtab64[h0.rot_off + st.k * (p.R + 2)];
            hcar = (f64x2){e0.x, e0.y};
        }
        st.rot = 0.0;
    }
    const lds_fp lrow = (lds_fp)(sObs + el * LDX);
    __syncthreads();
    asm volatile("" : "+v"(st.px), "+v"(st.py), "+v"(st.vx), "+v"(st.vy), "+v"(st.k), "+v"(st.time), "+v"(st.next), "+v"(st.passed), "+v"(k72));
    if constexpr (LIT) asm volatile("" : "+v"(hcar));

    const int TT = last_val ? T + 1 : T;   // pc_rollout_ex: tail iteration t == T = the final observation's value only (see rollout_kernel)
    float rsum = 0.0f;
    // The rows a lane stores every step, as per-lane POINTERS that advance by one buffer row per step (a 64-bit add each) instead of
    // base + (t N + e) formed from kernel arguments: those live in SGPRs this kernel does not have (84 are spilled to vector lanes) and
    // every use reloaded a pair with v_readlane before the address arithmetic -- ~50 vector instructions per wave and step.
    constexpr bool PTRS = RPL <= 5;      // (the 33-ray small forms have no registers for them: they spilled)
    const int64_t e_draw = e_wg + wave * (EPW / 8) + lk;                              // the env this lane draws for (lk < EPW / 8, lc == 0)
    float* pa_act = act_buf + e_draw;
    float* pa_lp = logprob_buf + e_draw;
    float* pa_val = val_buf + e_draw;
    float* pa_rew = rew_buf + e_env;
    float* pa_term = term_buf + N + e_env;                                             // flags that precede obs t + 1 (train.py:176-177,195)
    float* pa_trunc = trunc_buf + N + e_env;
    float* pa_obs = obs_buf + (N + e_wg) * D + 4 * (lane + 64 * wave);                 // FAST: this lane's 16 bytes of the workgroup's row block
    const int64_t row_step = N, obs_step = N * D;
#pragma unroll 1
    for (int t = 0; t < TT; ++t) {
        const bool tail = t == T;          // (uniform)
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
                    const float raw = sObs[(16 * et + lc) * LDX + f];
                    x[et][ks] = (!FAST || f < D) ? raw : 0.0f;
                }
            if constexpr (ET == 2) {
                if (!(dbg & 1)) policy_pass<KS>(sW1, sB1, sW2, ht0, ht1, x, out, lc, lk, lane);  // dbg: timing ablations only
            }
        } else {
            Pieces<PREC> x[ET][KB];
            if constexpr (XPRE) {      // the operands were split where the entries were produced (env_step_wave / the reset fix-up): two 16-byte reads
                static_assert(ET == 1 && KB == 1, "16 envs, one K block");
                const u32x4* xp = reinterpret_cast<const u32x4*>(sXp) + lc * 8 + lk;      // env lc: 8 x 16 bytes, piece 0 then piece 1
                x[0][0].p[0] = xp[0];
                x[0][0].p[1] = xp[4];
            } else {
#pragma unroll
            for (int et = 0; et < ET; ++et) {
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int f = 8 * (4 * kb + lk) + j;
                        const float raw = sObs[(16 * et + lc) * LDX + f];   // unconditional (inside the workgroup's LDS), see K9
                        v[j] = f < D ? raw : 0.0f;
                        if constexpr (PREC == 2) v[j] = clamp_h(v[j] * PolScale<PREC>::sx);   // the observations' scaled domain
                    }
                    x[et][kb] = split8<PREC>(v);
                }
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
        int act = 8;
        if (lk < EPW / 8) {   // every wave draws for EPW / 8 of the envs, 16 lanes (= outputs) per env, exactly as policy_kernel<SPLIT>
            const int dl = wave * (EPW / 8) + lk, oi = lc;
            const int64_t e = e_wg + dl;
            float ps = 0.0f;
#pragma unroll
            for (int w = 0; w < 8; ++w) ps += sOut[(w * EPW + dl) * LDO + oi];  // fixed order
            const float tsum = __builtin_fmaf(ps, PolScale<PREC>::so_inv, sB2[oi]);   // outputs back from their scaled domain
            const uint64_t o = off0 + (uint64_t)t;
            if (t == 0 || (o & 3) == 0) rnd = philox_block(seed, o >> 2, (uint64_t)e);  // uniform: ten rounds per 4 steps
            float lp, val;
            if (!(dbg & 4)) policy_tail_row(tsum, oi, A, philox_word_uniform(rnd, (unsigned)(o & 3)), lane, act, lp, val);
            else { act = 0; lp = tsum; val = tsum; }
            if (oi == 0 && e < N) {
                if constexpr (!WOWN) sAct[dl] = act;
                if (tail) {
                    last_val[e] = val;
                } else if constexpr (PTRS) {
                    *pa_act = (float)act;
                    *pa_lp = lp;
                    *pa_val = val;
                } else {
                    const int64_t row = (int64_t)t * N + e;
                    act_buf[row] = (float)act;
                    logprob_buf[row] = lp;
                    val_buf[row] = val;
                }
            }
        }
        if constexpr (PTRS) {
            pa_act += row_step;
            pa_lp += row_step;
            pa_val += row_step;
        }
        if constexpr (!WOWN) lds_barrier();   // (WOWN: the wave steps the two envs it drew for -- lanes 0 and 16 hold their actions)
        PC_STAMP(3)
        if (tail) break;
        // ---------------- E(t): 4 waves x 4 lanes per env (one more barrier inside, where the sweep parts meet)
        const bool last = t + 1 == T;
        if constexpr (FAST) {
            if (!(dbg & 2)) {
                float rw, tf, cf;
                bool done;
                if constexpr (WOWN) {
                    const int a0 = __builtin_amdgcn_readlane(act, 0), a1 = __builtin_amdgcn_readlane(act, 16);
                    const int a = e_valid ? (lane < 32 ? a0 : a1) : 8;
                    if constexpr (MODE == 2)      // the host found room for the 1/den table: compiled for it (one env step in the kernel, not two)
                        done = env_step_wave<RPL, true, LIT>(p, h0, ft, fl, gq[0], g, part, st, k72, a, reward_scale, lrow, rw, tf, cf, lane, t, wave, &hcar, XPRE ? xrow : nullptr);
                    else
                        done = rden_lds   // (uniform)
                            ? env_step_wave<RPL, true, LIT>(p, h0, ft, fl, gq[0], g, part, st, k72, a, reward_scale, lrow, rw, tf, cf, lane, t, wave, &hcar, XPRE ? xrow : nullptr)
                            : env_step_wave<RPL, false, LIT>(p, h0, ft, fl, gq[0], g, part, st, k72, a, reward_scale, lrow, rw, tf, cf, lane, t, wave, &hcar, XPRE ? xrow : nullptr);
                } else {
                    const int a = e_valid ? sAct[el] : 8;
                    done = rden_lds   // (uniform)
                        ? env_step_fast<RPL, true, 2, PARTS, 0, true, LIT, true>(p, h0, ft, fl, gq, g, st, k72, a, reward_scale, lrow, rw, tf, cf, t, lane, wave, part, exch, part == 0, sHit + el * PARTS, &hcar)
                        : env_step_fast<RPL, false, 2, PARTS, 0, true, LIT, true>(p, h0, ft, fl, gq, g, st, k72, a, reward_scale, lrow, rw, tf, cf, t, lane, wave, part, exch, part == 0, sHit + el * PARTS, &hcar);
                }
                rsum += rw;
                if (__builtin_amdgcn_ballot_w64(done) != 0) {
                    if (done) {
                        if (part == 0)   // (uniform) the row-writing wave: reset observation of finished envs
                        {   // all reads, then the writes (see rollout_kernel)
                            float ro[(DC + 3) / 4];
#pragma unroll
                            for (int j = 0; j < (DC + 3) / 4; ++j) ro[j] = ft.reset[g + 4 * j];
#pragma unroll
                            for (int j = 0; j < (DC + 3) / 4; ++j)
                                if (g + 4 * j < DC) {
                                    lrow[g + 4 * j] = ro[j];
                                    if constexpr (XPRE) write_pieces(xrow, g + 4 * j, ro[j]);
                                }
                        }
                        env_reset_fast(h0, st, k72);
                        if constexpr (LIT) {
                            const double2 e0 = p.dirtab64[h0.rot_off];      // row 0 = start_rot
                            hcar = (f64x2){e0.x, e0.y};
                        }
                    }
                }
                if constexpr (PTRS) {
                    if (last) {        // (uniform, once per launch) the flags and the observation of the final step go to next_*
                        pa_term = next_term + e_env;
                        pa_trunc = next_trunc + e_env;
                        pa_obs = next_obs + e_wg * D + 4 * (lane + 64 * wave);
                    }
                    if (part == 0) {   // (uniform) the row-writing wave: per-env scalars
                        if (g == 0 && e_valid) {
                            *pa_rew = rw;
                            *pa_term = tf;
                            *pa_trunc = cf;
                        }
                    }
                    pa_rew += row_step;
                    pa_term += row_step;
                    pa_trunc += row_step;
                } else if (part == 0) {   // (uniform) the row-writing wave: per-env scalars
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
            lds_barrier();    // the workgroup's observation rows are complete
            PC_STAMP(7)
            // rows -> rollout buffer: the workgroup's 32 rows are contiguous there (32 * D floats): waves 0 .. 2 (.. 4) store 64 float4 each
            {
                float* dstg = (last ? next_obs : obs_buf + (int64_t)(t + 1) * N * D) + e_wg * D;
                const int64_t left = N - e_wg;
                const int n_rows = left >= EPW ? EPW : (int)left;
                if (vec_ok && n_rows == EPW) {
                    const int i = lane + 64 * wave;
                    if (i < EPW / 4 * DC) {
                        if constexpr (PTRS) *reinterpret_cast<f32x4*>(pa_obs) = reinterpret_cast<const f32x4*>(sObs)[i];
                        else reinterpret_cast<f32x4*>(dstg)[i] = reinterpret_cast<const f32x4*>(sObs)[i];
                    }
                } else {
                    for (int i = tid; i < n_rows * D; i += 512) dstg[i] = sObs[i];
                }
                if constexpr (PTRS) pa_obs += obs_step;
            }
        } else if constexpr (EPW == 32) {
            if (!(dbg & 2)) {
                float* orow = !e_valid ? nullptr : (last ? next_obs + e_env * D : obs_buf + ((int64_t)(t + 1) * N + e_env) * D);
                float rw;
                bool term, trunc;
                int passed;
                if (rden_lds)  // uniform
                    env_step_core<float, RPL, PARTS, true, true>(q, trk, g, 2, st, (int64_t)sAct[el], reward_scale, orow, nullptr,
                                                           e_valid && part == 0 ? sObs + el * LDX : nullptr, rw, term, trunc, passed, part,
                                                           exch, rdl);
                else
                    env_step_core<float, RPL, PARTS, false, true>(q, trk, g, 2, st, (int64_t)sAct[el], reward_scale, orow, nullptr,
                                                     e_valid && part == 0 ? sObs + el * LDX : nullptr, rw, term, trunc, passed, part, exch);
                rsum += rw;
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
    if (e_valid && g == 0 && part == 0) {
        if constexpr (LIT) st.rot = p.dirtab64[h0.rot_off + st.k * (p.R + 2) + p.R + 1].x;
        env_store<StateT>(ps, e_env, st);
        if (rew_sum) rew_sum[e_env] = rsum;
    }
}
