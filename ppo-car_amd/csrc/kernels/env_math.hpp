// env_math.hpp -- part of the single translation unit ppocar.hip (included there, in order; not a stand-alone header).
// device-side data (track header, env parameters) and the reference's ray / segment arithmetic in float64 and float32.
#pragma once

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
