// env_math.hpp -- part of the single translation unit ppocar.hip (included there, in order; not a stand-alone header).
// device-side data (track header, env parameters) and the reference's ray / segment arithmetic in float64 and float32.
#pragma once

// Developer-only timing ablations (a SEPARATE library the product never loads: rollout.hpp, tools/ab_run.sh); 0 in every shipped build
#ifndef PC_ABLATE
#define PC_ABLATE 0
#endif

// ------------------------------------------------------------------------------------------
// device-side data
// ------------------------------------------------------------------------------------------
struct TrackHdr {        // one per track, read with scalar loads
    int wall_off, S;     // segs[wall_off .. wall_off+S): the walls
    int gate_off, G;     // segs[gate_off .. gate_off+G): the reward gates
    int head_off;        // F32: heading table [72] (cos, sin) of radians(start_rot + 5 j)
    int start_collides;  // Car.update at reset already hits a wall (car_env.py:686,468-469)
    int vtx_off, nV;     // F32: the walls again as vertex chains, vtx[vtx_off .. vtx_off+nV); nV is padded to a multiple of 4
    int dir_off;         // F32: ray direction table [361] of this track in dirtab / dirtab64 (entry 360 = (0, 0): no ray)
                         // F64: first slot of the track's angle -> (cos, sin) hash table in dirhash (Math<double>; head_off = its slot
                         //      mask), -1 = none
    int rden_off;        // F32: 1/den table [361][nV] of this track in rden (row 360 = +inf: never hits)
    int n_chain;         // F32: chain vertices before the padding to a multiple of 4 (vtx[n_chain .. nV) are sentinels)
    unsigned idx_mask;   // F32: (1 << b) - 1, b = max(5, ceil(log2(nV))): the low bits of a sweep candidate carry its vertex index
    double start_x, start_y, start_rot;
    double ax0, ay0;            // F32: the anchor of the sweep's float32 coordinates: the centre of the wall vertices' bounding box
    float bx0, bx1, by0, by1;   // F32: that bounding box (the sweep's flag threshold is priced from it)
    int n_scan;                 // F32: how many segments carry PC_SEG_SCAN (diagnostic)
    int brk2;                   // F32: index of the chain's SECOND chain-start vertex when the walls are exactly two chains (-1 otherwise)
    int vtxp_off;               // F32: those two chains have the same length (n_chain = 2 brk2): their vertices again, packed by
                                //      position in the chain, vtxp[vtxp_off .. vtxp_off + brk2) (-1 otherwise)
    int rot_off, n_rot;         // F64: the track's ROTATION TABLE in dirtab64 (Math<double>): n_rot rows of R + 1 entries; -1 = none
    int lat_off;                // the track's float32 direction lattice [361] in dirtab (F32: = dir_off; F64: the selector's directions, -1 = none)
    int sel_ok;                 // F64: the float32 selector may run on this track (chain tables built, <= 8192 vertices, fits 2000 px)
};

// One wall / gate segment as the reference holds it (Boundary.get_points, car_env.py:74): 32 bytes.
struct Seg { double x1, y1, x2, y2; };

// F32 wall sweep: the walls as chains of vertices, (xr, yr) = the vertex relative to the track's anchor (TrackHdr::ax0, ay0: the
// centre of its bounding box), rounded from float64 -- the sweep only SELECTS, so it can afford float32 coordinates as long as
// the flag threshold prices their rounding (flag_threshold).  Vertex k closes the segment (k-1, k) unless it
// starts a new chain: then its edge (ex, ey) is (0, 0) (a real wall has length).  (ex, ey) = the UNIT vector along p[k-1] - p[k]
// rounded from float64 -- the selector's u = cross(e, a) / cross(e, dir) does not depend on the edge's length, and with a unit
// edge cross(e, a) is the car's distance from the wall line in pixels, which makes its rounding threshold one number per car --,
// (exs, eys) the same scaled by 2^-40 (exact): the sweep's selector values live in that scaled domain (see wall_sweep_f32).
// 32 bytes = one s_load_dwordx8.
// Why chains: the reference's hit test 0 < t < 1 (car_env.py:178) is "the two endpoints lie strictly on
// opposite sides of the ray line".  Evaluated per VERTEX -- one cross product c_k = cross(p_k - pos, dir)
// shared by the two segments that meet there -- a float32 ray cannot slip between two adjacent walls
// through the rounding-wide crack that two independently rounded t's leave at their common corner.
struct Vtx { float xr, yr, ex, ey, exs, eys, pad0, pad1; };
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// Walls that are exactly two chains of the same length L (big_track.json: the outer and the inner loop, 13 vertices each):
// record i holds vertex i of BOTH chains -- component 0 = chain vertex i, component 1 = chain vertex L + i -- so that the
// sweep's packed-fp32 instructions advance both chains at once (wall_sweep_loops).  Same values as the two Vtx records.
struct VtxP { f32x2 xr, yr, ex, ey, exs, eys; };   // 48 bytes
__device__ __forceinline__ bool vtx_brk(const Vtx& v) {   // chain start / padding sentinel: a zero edge (integer test: stays on the SALU)
    return ((__float_as_uint(v.ex) | __float_as_uint(v.ey)) << 1) == 0u;
}
// The float64 refinement's view of the same chain, one 48-byte record per vertex k = the segment that vertex closes:
// (x1, y1) -> (x1 - ex, y1 - ey) with (ex, ey) = p[k-1] - p[k] formed in float64 as the reference forms (x1 - x2), (y1 - y2)
// (car_env.py:171); h = 0.5 - (0.05 px) / |e|: a refined hit whose parameter t satisfies |t - 0.5| < h lies at least 0.05 px
// inside the segment's ends (-1 for chain starts / padding, which have a zero edge: no segment); prev / next = the chain
// neighbours of the segment -- prev shares its first endpoint, next its second (0: none; a closed loop wraps).
struct SegD { double x1, y1, ex, ey, h; int prev_next, pad; };   // prev in bits 0..14, PC_SEG_SCAN, next in bits 16..30
// PC_SEG_SCAN: this wall comes closer to another one than float32 can order hits, without the two being plain chain neighbours
// (walls that cross or touch -- a T-junction, an X --, a spike sharper than ~13 degrees, a wall shorter than the end margin;
// found on the host at pc_env_create): h = -1, and a ray whose selection lands here is resolved by the float64 scan of the chain
constexpr int PC_SEG_SCAN = 0x8000;

template <typename T> struct EnvParams {
    int64_t N;
    int lg;              // log2(lanes per env)
    int n_nominal;       // Car num_rays (car_env.py:227)
    int q;               // n // 4: stride of the collision rays (car_env.py:389)
    int nc;              // how many there are: len(range(0, n, n // 4)) -- 4 when 4 divides n, up to 7 otherwise
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
    const VtxP* __restrict__ vtxp;          // F32 only: the chains of two-equal-loop tracks, packed (TrackHdr::vtxp_off)
    const double2* __restrict__ headtab;    // F32 only: (cos, sin) of radians(start_rot + 5 j), j < 72, per track
    // F32 only.  A ray's direction angle is start_rot + 5 k + step_deg * ray degrees (k = integer turn count): an integer
    // offset from start_rot, so all directions live on a 360-entry lattice per track.
    const float2* __restrict__ dirtab;      // [n_tracks][361] (cos, sin) of radians(start_rot + j), float64 libm, rounded
    const float* __restrict__ rden;         // [n_tracks][361][nV] 1 / (ey*dx - ex*dy) exactly as the sweep computes it (device-built)
    const double2* __restrict__ dirtab64;   // [n_tracks][361] the same lattice in float64 (libm): the refinement's ray directions
    const SegD* __restrict__ seg64;         // F32 only: the wall chains for the refinement, indexed like vtx
    const struct F64Dir* __restrict__ dirhash;   // F64 only: glibc's cos / sin of every angle an episode can reach (Math<double>)
    const float* __restrict__ reset_obs;    // [n_tracks][D]
    // The parameter only selects the arithmetic of the functions that take the struct (Math<T>, env_load<T>): the members do not depend
    // on it.  A kernel launched with EnvParams<float> that steps an F64 handle (the persistent kernels' literal form) reads the state
    // through the same members under the other name.
    template <typename U> __host__ __device__ EnvParams<U> as() const {
        static_assert(sizeof(EnvParams<U>) == sizeof(EnvParams<T>), "one layout");
        EnvParams<U> q;
        __builtin_memcpy(&q, this, sizeof(q));
        return q;
    }
};

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

// The same arithmetic, also returning t (car_env.py:175): where along the wall the hit lies (the selector kernels' end-margin test)
struct LitR { double d, t; bool hit; };
__device__ __forceinline__ LitR cast_ref_t(double x1, double y1, double x2, double y2, double x3, double y3, double dx, double dy) {
    // (straight-line: a zero denominator makes t infinite or NaN and the test below false -- :172's None --, and the distance is
    // formed whether or not the test holds and selected afterwards: nine of these in a row with a branch each cost the persistent
    // kernel 200 spilled registers)
    const double x4 = x3 + dx, y4 = y3 + dy;                                  // :169
    const double den = (x1 - x2) * (y3 - y4) - (y1 - y2) * (x3 - x4);         // :171
    const double t = ((x1 - x3) * (y3 - y4) - (y1 - y3) * (x3 - x4)) / den;   // :175
    // u = -un / den (:176) enters only through `u > 0` (:178): with IEEE division that is "the numerator -un and den are nonzero
    // and of one sign" (a quotient of these magnitudes cannot underflow to zero) -- no second division
    const double un = (x1 - x2) * (y1 - y3) - (y1 - y2) * (x1 - x3);
    const bool u_pos = ((un < 0) & (den > 0)) | ((un > 0) & (den < 0));
    const double ptx = x1 + t * (x2 - x1), pty = y1 + t * (y2 - y1);          // :180-181
    const double d0 = x3 - ptx, d1 = y3 - pty;
    const double d = sqrt(fma(d1, d1, d0 * d0));
    LitR r;
    r.hit = (0 < t) & (t < 1) & u_pos;                                        // :172 (den == 0: u_pos is false), :178
    r.t = t;
    r.d = r.hit ? d : 1000.0;
    return r;
}

// ---- F32 mode = float32 SELECTOR + float64 REFINEMENT ---------------------------------------------------------
// a_k = p_k - pos (formed in float64, then rounded), c_k = cross(a_k, dir) = ay_k*dx - ax_k*dy, e = p1 - p2.  With the
// reference's t, u (car_env.py:171-176; x3 - x4 = -dx, y3 - y4 = -dy):
//   den = ey*dx - ex*dy = c1 - c2,   t = c1/den,   u = (ey*ax1 - ex*ay1)/den = un/den
//   0 < t < 1  <=>  c1 and c2 have strictly opposite signs,   and the distance |pos - pt| equals u because |dir| = 1.
// The float32 wall sweep (env_step.hpp) only SELECTS, per ray, the wall segment with the smallest float32 u among the
// segments it cannot exclude; the distance that is reported -- observation, `< 10.0` tests (car_env.py:387-390) -- is then
// recomputed for that one segment in float64 from the float64 car position, the float64 direction lattice and the
// float64 wall coordinates, under the reference's strict test `0 < t < 1 and u > 0` (car_env.py:178).
__device__ __forceinline__ float cross_f(float ax, float ay, float dx, float dy) {
    return __builtin_fmaf(ay, dx, -(ax * dy));
}

// float64 numerators / denominator of Ray.cast (car_env.py:171-176) for the segment sg and the ray (px, py) + s (dx, dy)
struct CastD { double den, tn, un; };
__device__ __forceinline__ CastD cast_terms(const SegD& sg, const double px, const double py, const double dx, const double dy) {
    const double ax = sg.x1 - px, ay = sg.y1 - py;                 // (x1 - x3), (y1 - y3)
    CastD r;
    r.den = __builtin_fma(sg.ey, dx, -(sg.ex * dy));               // :171
    r.tn = __builtin_fma(ay, dx, -(ax * dy));                      // :175 numerator
    r.un = __builtin_fma(sg.ey, ax, -(sg.ex * ay));                // :176 numerator
    return r;
}
// 1 / den to ~2^-46 relative (v_rcp_f64 is good to ~2^-23; one Newton step squares that): the refined distance un * (1/den)
// is then within 2e-14 relative -- 2e-11 px at the 1000 px ray limit -- of the exactly rounded quotient.
__device__ __forceinline__ double rcp_d(const double den) {
    const double r0 = __builtin_amdgcn_rcp(den);
    return __builtin_fma(r0, __builtin_fma(-den, r0, 1.0), r0);
}
// The reference's `0 < t < 1 and u > 0` (car_env.py:178) is decidable on the numerators, without rounding a quotient (t > 0 iff tn
// and den have the same sign, t < 1 iff |tn| < |den|, u > 0 likewise by signs) -- but the reference does not evaluate that test
// in exact arithmetic: it forms (x4, y4) = pos + dir (car_env.py:169) and works
// with the ROUNDED differences (x3 - x4), (y3 - y4), i.e. with a direction perturbed by up to an ulp of the car's position (for
// dir = (cos 90 deg, 1) = (6e-17, 1) the x component vanishes altogether).  Away from a tie that changes nothing (3e-14 rad; the
// test's margins are relative 1e-13).  AT a tie -- a ray that passes exactly through a vertex, a car exactly on a wall's line,
// exactly parallel: what axis-aligned tracks with integer coordinates produce at will -- hit or miss is decided by those
// roundings, and the two walls that meet at the vertex can both say "miss" (the ray passes between them and reports whatever
// lies behind: 100s of pixels away).  So wherever a verdict is taken in float64 outside refine_fast's certified interior --
// the gate casts, the corner neighbours, the chain scan -- it is taken on the reference's OWN numerators and denominator, formed
// literally (car_env.py:166-176; cast_exact).  No quotient is needed for the verdict: with IEEE division, 0 < fl(n / d) <=> n and d
// have the same sign (nothing here underflows), fl(n / d) < 1 <=> |n| < |d| (the largest quotient of two doubles with |n| < |d| is
// 1 - 2^-53, which does not round up to 1), and u > 0 likewise by signs.  The DISTANCE of a hit is refine_fast's very expression
// (cast_terms' numerators, un / den to 2^-46: within a few float64 ulps of the reference's norm of the hit point): a ray reports the
// same bits whether its slot was certified by the sweep or flagged and resolved here -- which depends on how the rays are dealt
// to lanes (flags are kept per pair of slots) -- so the result does not depend on the launch geometry.
struct CastR { double d; bool hit; };
__device__ __forceinline__ CastR cast_exact(const SegD& sg, const double px, const double py, const double dx, const double dy) {
    const double x4 = px + dx, y4 = py + dy;                                   // :169
    const double mx = px - x4, my = py - y4;                                   // (x3 - x4), (y3 - y4)
    const double ax = sg.x1 - px, ay = sg.y1 - py;                             // (x1 - x3), (y1 - y3)
    const double den = sg.ex * my - sg.ey * mx;                                // :171 (the translation unit is compiled without contraction)
    const double tn = ax * my - ay * mx;                                       // :175 numerator
    const double un = -(sg.ex * ay - sg.ey * ax);                              // :176 numerator
    const bool dpos = den > 0.0, dneg = den < 0.0;                             // den == 0: None (:172)
    const bool t_ok = ((tn > 0.0) & dpos) | ((tn < 0.0) & dneg);               // 0 < t
    const bool u_ok = ((un > 0.0) & dpos) | ((un < 0.0) & dneg);               // u > 0
    CastR r;
    r.hit = t_ok & (__builtin_fabs(tn) < __builtin_fabs(den)) & u_ok;          // t < 1
    const CastD c = cast_terms(sg, px, py, dx, dy);
    r.d = r.hit ? c.un * rcp_d(c.den) : 1000.0;                                // (1000.0 = Ray.get_distance's `largest_distance` for a miss)
    return r;
}
// Ray.get_distance (car_env.py:186-213) of ONE ray against a whole wall chain in float64: the refinement's exhaustive form,
// taken only by the rare lanes whose float32 selection could not be certified (see refine_careful).  segs(1 .. n - 1).
template <typename LoadSeg>
__device__ __forceinline__ double scan_chain_d(const LoadSeg& segs, const int n, const double px, const double py, const double dx,
                                               const double dy) {
    double best = 1000.0;                                          // :198
    for (int k = 1; k < n; ++k) best = __builtin_fmin(best, cast_exact(segs(k), px, py, dx, dy).d);   // :203-207
    return best;
}
// ---- the float64 distance of one ray, given the float32 sweep's selection (candidate bits: vertex index k in the low bits;
// index 0 = "nothing certified": the sweep found no segment within 1000.5 px, or it flagged the ray as too close to a vertex
// or the car as too close to a wall line for float32 side tests to be trusted).
//   refine_fast     the selected segment's hit in float64; `ok` iff it lies at least 0.05 px inside both ends of the segment
//                   (|t - 0.5| < h; false for k == 0, whose zero edge makes t NaN) -- then u is the answer: for an unflagged ray the
//                   sweep's side tests and signs of u are the exact ones, so every wall the reference hits was a candidate and the
//                   selected one is a real hit in front of the car; float32 can order two hits wrongly only where two walls meet;
//   refine_careful  everything else (a few 1e-4 of the rays): a hit within 0.05 px of a corner is compared with the two chain
//                   neighbours under the strict test; with nothing certified the whole chain is scanned in float64.
// Every decision is per lane and depends on that lane's ray only: the result does not depend on the launch geometry.
// (segs(k): accessor of the chain table -- global memory in the per-step kernel, LDS in the persistent ones)
__device__ __forceinline__ double refine_fast(const SegD& sg, const double px, const double py, const double dx, const double dy, bool& ok) {
    const CastD c = cast_terms(sg, px, py, dx, dy);
    const double r = rcp_d(c.den);
    ok = __builtin_fabs(__builtin_fma(c.tn, r, -0.5)) < sg.h;
    return c.un * r;
}
template <typename LoadSeg>
__device__ __forceinline__ double refine_careful(const int k, const LoadSeg& segs, const int nV, const double px, const double py,
                                                 const double dx, const double dy) {
    double d = 1000.0;
    bool any = false;
    if (k != 0) {
        const SegD sg = segs(k);
        const CastR c = cast_exact(sg, px, py, dx, dy);
        if (c.hit && !(sg.prev_next & PC_SEG_SCAN)) {   // the selection is a hit: is a chain neighbour's hit (around the corner it lies next to) nearer?
            any = true;
            d = c.d;
            d = __builtin_fmin(d, cast_exact(segs(sg.prev_next & 0x7fff), px, py, dx, dy).d);   // (index 0 = no neighbour: zero edge, never a hit)
            d = __builtin_fmin(d, cast_exact(segs((int)(((unsigned)sg.prev_next >> 16) & 0x7fff)), px, py, dx, dy).d);
        }
    }
    if (!any) d = scan_chain_d(segs, nV, px, py, dx, dy);
    return d;
}
// ---- PC_DTYPE_F64 handles inside the persistent kernel (rollout_kernel<..., LIT>): the float32 sweep SELECTS exactly as above, and
// what is then evaluated for the selected wall is the reference's own LITERAL arithmetic (cast_ref) on the float64 car position,
// glibc's direction (the rotation table) and the wall's endpoints as the track file gives them: SegD records whose (ex, ey) fields
// carry (x2, y2) on such handles.  The argument is the one above, with "exact" replaced by "literal": the sweep flags every ray
// within tau ~ 2e-3 px of a vertex and every car within tau of a wall line, and the literal arithmetic differs from the exact one by
// ~1e-10 px at most (rounded x3 - x4: see cast_exact) -- so for an unflagged ray the literal hit set IS the candidate set, the
// selected wall is a literal hit, and a nearer literal hit can only lie within float32's resolution of it: around the corner next
// to the hit (lit_careful's neighbours) or on a wall the host marked PC_SEG_SCAN (the literal loop over all walls).  Ray.get_distance
// (car_env.py:186-213) is the minimum of the literal distances, and that is what comes out, bit for bit.
//   lit_fast     the selected wall's literal cast; ok iff it is a hit at least the end margin inside both ends (|t - 0.5| < h)
//   lit_careful  the rest: corner neighbours, or (nothing certified / the selection is no literal hit / PC_SEG_SCAN) every wall
__device__ __forceinline__ double lit_fast(const SegD& sg, const double px, const double py, const double dx, const double dy, bool& ok) {
    const LitR c = cast_ref_t(sg.x1, sg.y1, sg.ex, sg.ey, px, py, dx, dy);
    ok = c.hit & (__builtin_fabs(c.t - 0.5) < sg.h);
    return c.d;
}
template <typename LoadSeg>
__device__ __forceinline__ double lit_careful(const int k, const LoadSeg& segs, const Seg* walls, const int S, const double px, const double py,
                                              const double dx, const double dy) {
    double d = 1000.0;                                             // :198
    bool any = false;
    if (k != 0) {
        const SegD sg = segs(k);
        const LitR c = cast_ref_t(sg.x1, sg.y1, sg.ex, sg.ey, px, py, dx, dy);
        if (c.hit && !(sg.prev_next & PC_SEG_SCAN)) {
            any = true;
            const SegD a = segs(sg.prev_next & 0x7fff), b = segs((int)(((unsigned)sg.prev_next >> 16) & 0x7fff));   // (index 0: a chain start, x2 = x1: den == 0, never a hit)
            const double da = cast_ref(a.x1, a.y1, a.ex, a.ey, px, py, dx, dy), db = cast_ref(b.x1, b.y1, b.ex, b.ey, px, py, dx, dy);
            d = c.d;
            d = da < d ? da : d;                                   // :203-207
            d = db < d ? db : d;
        }
    }
    if (!any) {
        for (int w = 0; w < S; ++w) {
            const Seg sg = walls[w];
            const double dd = cast_ref(sg.x1, sg.y1, sg.x2, sg.y2, px, py, dx, dy);
            d = dd < d ? dd : d;
        }
    }
    return d;
}
// ---- the careful path of ONE ray as a job of the WHOLE WAVE (the table-driven env steps, all 64 lanes active) ---------------------
// refine_careful / lit_careful run on the lane that owns the ray: one or two lanes of a wave at work, the chain scan ~900 instructions
// deep -- and a third of all wave-steps of the benchmark meet one (4.5e-4 of the rays end within 0.05 px of a corner, 2.5e-4 are flagged:
// 0.3 careful iterations per wave-step, ~6 % of the persistent kernel's vector instructions).  Here lane j measures chain segment j
// (nV <= 64: FT_VTX_MAX) for the job's ray (position and direction wave-uniform), and the answer is the minimum over the SAME set of
// segments the per-lane functions take: {k, its two chain neighbours} if segment k is a hit and not marked PC_SEG_SCAN, else the
// whole chain (LIT: every wall of the track in the reference's own list, car_env.py:203-207).  Every distance is the same function of
// the same operands, the minimum of doubles is exact: the same bits (tests: the persistent kernels against env_step_kernel, which
// keeps the per-lane form).
__device__ __forceinline__ double wave_min_d(double v) {
    // rows of 16 lanes by DPP butterflies (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror), the four rows through readlane
    const auto step = [&](auto CTRL) {
        constexpr int ctrl = decltype(CTRL)::value;
        const int lo = __double2loint(v), hi = __double2hiint(v);
        const int lo2 = __builtin_amdgcn_update_dpp(lo, lo, ctrl, 0xf, 0xf, false), hi2 = __builtin_amdgcn_update_dpp(hi, hi, ctrl, 0xf, 0xf, false);
        v = __builtin_fmin(v, __hiloint2double(hi2, lo2));
    };
    step(std::integral_constant<int, 0xb1>{});
    step(std::integral_constant<int, 0x4e>{});
    step(std::integral_constant<int, 0x141>{});
    step(std::integral_constant<int, 0x140>{});
    const auto row = [&](const int l) {
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
    };
    return __builtin_fmin(__builtin_fmin(row(0), row(16)), __builtin_fmin(row(32), row(48)));
}
__device__ __forceinline__ double bcast_d(const double v, const int l) {     // lane l's value in every lane (l wave-uniform)
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
template <bool LIT, typename LoadSeg>
__device__ __forceinline__ double careful_wave(const int k, const LoadSeg& segs, const int nV, const Seg* walls, const int S, const double px,
                                               const double py, const double dx, const double dy, const int lane) {
    const SegD sg = segs(lane < nV ? lane : 0);      // (record 0 is a chain start: a zero edge, never a hit)
    double dj;
    bool hit;
    if constexpr (LIT) {
        const LitR c = cast_ref_t(sg.x1, sg.y1, sg.ex, sg.ey, px, py, dx, dy);
        dj = c.d;
        hit = c.hit;
    } else {
        const CastR c = cast_exact(sg, px, py, dx, dy);
        dj = c.d;
        hit = c.hit;
    }
    const uint64_t hitm = __builtin_amdgcn_ballot_w64(hit & (lane < nV)), scanm = __builtin_amdgcn_ballot_w64((sg.prev_next & PC_SEG_SCAN) != 0);
    const int pn = __builtin_amdgcn_readlane(sg.prev_next, k);       // (k < nV <= 64)
    const bool any = (k != 0) & (bool)((hitm >> k) & 1) & !(bool)((scanm >> k) & 1);      // wave-uniform
    double dm;
    if (any) {
        const bool in = (lane == k) | (lane == (pn & 0x7fff)) | (lane == (int)(((unsigned)pn >> 16) & 0x7fff));   // (neighbour index 0 = none: record 0 never hits)
        dm = (in & (lane < nV)) ? dj : 1000.0;
    } else if constexpr (LIT) {
        const Seg w = walls[lane < S ? lane : 0];
        const double dw = cast_ref_t(w.x1, w.y1, w.x2, w.y2, px, py, dx, dy).d;
        dm = lane < S ? dw : 1000.0;
    } else {
        dm = ((lane >= 1) & (lane < nV)) ? dj : 1000.0;
    }
    return wave_min_d(dm);
}
// min(1000, d) / 1000 as the observation holds it (Ray.get_distance :198,:210-211; car_env.py:593,:595)
__device__ __forceinline__ float obs_dist(const double d) { return (float)(__builtin_fmin(d, 1000.0) * 0.001); }
// one ray against one segment (the reward gates: Car.check_collision(gate), car_env.py:387-390), float64, the reference's verdict
__device__ __forceinline__ double cast_d(const Seg& s, const double px, const double py, const double dx, const double dy) {
    const SegD sg = {s.x1, s.y1, s.x1 - s.x2, s.y1 - s.y2, 0.0, 0, 0};
    return __builtin_fmin(cast_exact(sg, px, py, dx, dy).d, 1000.0);
}

template <typename T> struct Math;

// F64 mode's directions.  The reference forms np.cos / np.sin of np.radians(angle) (car_env.py:426-427, :463-466, :584) for
// angle = rotation [+ a], rotation = the start rotation after a sequence of +-5.0 (each sum rounded: :440-442), a = the ray's
// whole-degree offset (:269).  Inside an episode (at most 1000 turns) only a few thousand distinct float64 rotations can occur
// -- the roundings merge the paths -- and ~15 k distinct angles: the host enumerates them, evaluates cos / sin with glibc (the
// reference's own libm) and the device LOOKS THEM UP by the angle's bit pattern: the reference's bits by construction, where the
// device's own cos / sin (ocml) differ from glibc in the last place for some arguments.  An angle that is not in the table
// (set_state with a rotation no episode reaches; more than 16 tracks: dir_off < 0) is evaluated on the device as before.
struct F64Dir { unsigned long long key; double c, s; unsigned long long pad; };     // 32 bytes; key = the angle's bits (degrees)
constexpr unsigned long long F64DIR_EMPTY = 0x7ff8dead00000000ull;                  // (a NaN pattern no sum produces)
constexpr int F64DIR_MAX_PROBE = 8;                                                 // the host sizes the table so that this holds
__host__ __device__ inline unsigned f64dir_hash(unsigned long long k) {
    k ^= k >> 29;
    k *= 0xBF58476D1CE4E5B9ull;
    return (unsigned)(k >> 32);
}
template <> struct Math<double> {
    static __device__ __forceinline__ bool lookup(const EnvParams<double>& p, const TrackHdr& h, const double angle, double& c,
                                                  double& s) {
        if (h.dir_off < 0) return false;
        const unsigned long long key = (unsigned long long)__double_as_longlong(angle);
        const unsigned mask = (unsigned)h.head_off;       // F64 mode: the table's slot mask
        const unsigned slot = f64dir_hash(key) & mask;
        // straight-line probes (no loop with an early exit: inside the ray loops that would keep them from unrolling and send
        // their register arrays to scratch): all keys of the probe window first, then the one entry that matched
        int at = -1;
#pragma unroll
        for (int probe = 0; probe < F64DIR_MAX_PROBE; ++probe) {
            const unsigned long long k = p.dirhash[h.dir_off + ((slot + probe) & mask)].key;
            at = (k == key) ? probe : at;      // (a key occurs once)
        }
        const F64Dir* e = p.dirhash + h.dir_off + ((slot + (at < 0 ? 0 : at)) & mask);
        const double ec = e->c, es = e->s;
        if (at < 0) return false;
        c = ec;
        s = es;
        return true;
    }
    // Round 5: the ROTATION TABLE.  An env's heading is start_rot after a sequence of +-5.0 (car_env.py:440-442), one of the n_rot
    // float64 values the host enumerated (breadth first, rotation 0 = start_rot: what reset gives).  The env state carries that
    // rotation's INDEX k beside the value (iv.x; -1 = a rotation set_state gave that no episode reaches): row k of the table holds
    // the R rays' (cos, sin) -- glibc's, of the very angles rot + ray * step -- and, as entry R, the indices of rot - 5.0 and rot +
    // 5.0 ((double) left, (double) right; -1 beyond the 1000 turns an episode can make).  One 16-byte load per direction where the
    // hash lookup below takes eight probes and the entry; the hash stays for k = -1.  Same values either way.  Entry R + 1 = (the
    // rotation itself, -): the selector kernel keeps only the row in registers and stores the value from here.
    static __device__ __forceinline__ bool indexed(const TrackHdr& h, const int k) { return (k >= 0) & (h.rot_off >= 0); }
    static __device__ __forceinline__ double2 rot_entry(const EnvParams<double>& p, const TrackHdr& h, const int k, const int col) {
        return p.dirtab64[h.rot_off + k * (p.R + 2) + col];
    }
    // the index of rot -+ 5.0 (left: -5.0, car_env.py:440; right: +5.0, :442)
    static __device__ __forceinline__ int turn(const EnvParams<double>& p, const TrackHdr& h, const int k, const bool left) {
        if (!indexed(h, k)) return -1;
        const double2 t = rot_entry(p, h, k, p.R);
        return (int)(left ? t.x : t.y);
    }
    // heading (cos, sin) of the float64 heading, as the reference forms it (:426-427, :584)
    static __device__ __forceinline__ void heading(const EnvParams<double>& p, const TrackHdr& h, int k, double rot, double& c,
                                                   double& s) {
        if (indexed(h, k)) {
            const double2 e = rot_entry(p, h, k, 0);      // (rot + 0 * step is rot itself)
            c = e.x;
            s = e.y;
            return;
        }
        if (lookup(p, h, rot, c, s)) return;
        const double a = d_radians(rot);
        c = cos(a);
        s = sin(a);
    }
    static __device__ __forceinline__ void ray_dir(const EnvParams<double>& p, const TrackHdr& h, int ray, int k, double rot,
                                                   double& dx, double& dy) {
        if (indexed(h, k)) {
            const double2 e = rot_entry(p, h, k, ray);
            dx = e.x;
            dy = e.y;
            return;
        }
        const double deg = rot + (double)(ray * p.step_deg);   // Ray.update(x, y, rot + a) :463-466, :153
        if (lookup(p, h, deg, dx, dy)) return;
        const double a = d_radians(deg);
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
    // observation entry of a refined distance: min(1000, d) / 1000 (car_env.py:198,:593) as a float64 multiply, then the float32 cast (:595)
    static __device__ __forceinline__ float norm_dist(double d) { return obs_dist(d); }
    // float64 multiply by the reciprocal, then the float32 cast: equals (float)(v / d) unless v/d sits
    // within 1e-16 (relative) of a float32 rounding boundary
    static __device__ __forceinline__ float norm(double v, double d) { return (float)(v * (1.0 / d)); }
};
