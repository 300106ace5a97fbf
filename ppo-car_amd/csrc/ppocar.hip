// ppocar.hip -- HIP kernels (gfx950 / CDNA4) and the C-ABI of libppocar.so.  ONE translation unit: the kernels live in
// kernels/*.hpp (env_math, env_step, gae_sample, policy, rollout, update), included below in dependency order; this file
// holds the host side (handles, launch configuration, the extern "C" entry points of include/ppocar.h).
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
//   K9  rollout_kernel<KS, RPL, PREC, MODE> / K9s rollout_small_kernel<..., EPW>   the whole rollout (train.py:173-195) as one
//                                       persistent launch (large / small batches)
//   K10-12 ppo_fwdbwd / grad_reduce / adam (+ clip_adam_mb, the multi-rank step)   one PPO minibatch step without any library GEMM
//   K13 xchg_allreduce_kernel          the per-minibatch gradient all-reduce as a one-shot exchange over peer-mapped buffers (pc_xchg_*)
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

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <climits>
#include <utility>
#include <vector>

#include "ppocar_internal.h"

// ---- the kernels, in dependency order (one translation unit: every kernel is compiled with this file's flags) ----
#include "kernels/env_math.hpp"
#include "kernels/env_step.hpp"
#include "kernels/gae_sample.hpp"
#include "kernels/policy.hpp"
#include "kernels/rollout.hpp"
#include "kernels/env_steps.hpp"
#include "kernels/update.hpp"
#include "kernels/exchange.hpp"

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
// Developer-only quick build (tools/ab_run.sh with -DPC_DEV_MIN=<mask>; never the product: the host layer refuses it like the
// ablation build): the dispatch tables keep only the benchmarked kernels -- bit 0: rollout_kernel<6, 9, 2, 3> (target), bit 1:
// rollout_small_kernel<6, 5, 2, 1, 16> (cfg1), bit 2: rollout_kernel<10, 17, 2, 1> (cfg2), bits 3 / 5: the literal form of bits 0 / 2
// for F64 handles, bit 4: rollout_f64_kernel<6, 9, 2> (the filter form), bit 6: the literal form of bit 1; the fp16x2 policy kernels, the
// float32 env-step kernels and the update kernels stay -- so that one kernel experiment compiles in seconds instead of 75.
#ifdef PC_DEV_MIN
#define PC_FULL(...) return PC_ERR_UNSUPPORTED
#define PC_DEV(bit, ...) do { if constexpr (((PC_DEV_MIN) >> (bit)) & 1) { __VA_ARGS__; } else return PC_ERR_UNSUPPORTED; } while (0)
#else
#define PC_FULL(...) __VA_ARGS__
#define PC_DEV(bit, ...) __VA_ARGS__
#endif
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

// pc_rollout's dispatch options: per env handle (pc_env_set_option).  There is no process-wide state: a new handle starts from
// the member initialisers below.
struct RolloutOpts {
    int form = -1;        // -1 auto, 0 = 256 envs per workgroup, 1 = 32 envs per workgroup
    int rden = 1;         // stage the 1/den table in LDS when it fits (0: never; test / tuning knob)
    int epw_override = 0; // 0 = automatic, 16 / 32 / 128 / 256 = force (test knob)
    int fast = 1;         // the fast modes (LDS tables behind LDS pointers) when the shape allows them (0: never; A/B knob)
    int nv28 = 1;         // kernels compiled for a padded wall chain of 28 vertices (big_track.json) when every track of the batch has one
    int deinterleave = 1; // two tracks interleaved in evenly split blocks: the block's two waves de-interleave it (0: the per-track passes; A/B and test knob)
    int step_form = 0;    // pc_env_step / pc_env_step_many: 0 = automatic (pc_env_step: the table-driven form K1f from PC_STEP_FAST_MIN_ENVS envs
                          // on, where the shape has one; pc_env_step_many: wherever the shape has one), 1 = always the generic per-step kernel
                          // K1, 2 = K1f wherever the shape has one (any batch size)
};
constexpr int64_t PC_STEP_FAST_MIN_ENVS = 8192;   // below: K1's 8+ lanes per env fill the device as well as K1f's workgroups of 128 envs (call to call,
                                                  // tools/step_forms_probe.py: 11.2 against 12.5 us at 4096 envs, 13.6 = 13.6 at 8192, 21 against 15 at 16384, 34 against 21 at 65536)
constexpr int kDefaultPolicyPrecision = 2;   // pc_policy_create(precision = -1): 0 = fp32-input MFMA; split forms on the 16-bit matrix cores (need D <= 40, A <= 9): 1 = bf16 x 3, 2 = fp16 x 2

struct pc_env {
    int device = 0;
    RolloutOpts opt;
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
    bool mixed = false;            // a track_id array was given (known before the geometry is chosen)
    bool track_blocks32 = false;   // mixed tracks: every aligned block of 32 envs holds ONE track (what pc_rollout needs)
    int track_block = 0;           // ... the largest of 256 / 128 / 64 / 32 for which that holds (0: none)
    bool track_bal64 = false, track_bal32 = false;   // two tracks, interleaved, every aligned block of 64 / 32 envs split evenly between them (and N a
                                                     // multiple of the block): the block's two waves de-interleave it (rollout_kernel's mode 7)
    TrackHdr* hdr = nullptr;
    Seg* segs = nullptr;
    Vtx* vtx = nullptr;
    VtxP* vtxp = nullptr;
    double2* headtab = nullptr;
    float2* dirtab = nullptr;
    float* rden = nullptr;
    double2* dirtab64 = nullptr;
    SegD* seg64 = nullptr;
    F64Dir* dirhash = nullptr;
    std::vector<std::unordered_map<uint64_t, int>> rot_ids;   // F64, host only: per track, rotation bits -> row of the rotation table (pc_env_set_state)
    std::vector<std::vector<int>> rot_depth;                  // F64, host only: per track and row, how many turns from start_rot reach it
    int last_kernel = 0;           // PC_KERNEL_*: what the last successful pc_rollout launched (pc_env_last_rollout_kernel)
    int last_step_kernel = 0;      // PC_STEP_*: what the last pc_env_step / pc_env_step_many launched (pc_env_last_step_kernel)
    bool f64_offgrid = false;      // F64: pc_env_set_state left an env whose episode can leave the rotation table (a rotation that is not a
                                   // row, or a row more turns from start_rot than the env's time step): the selector kernel needs rows
    float* reset_obs = nullptr;

    template <typename T> EnvParams<T> params() const {
        EnvParams<T> p;
        p.N = N;
        p.lg = lg;
        p.n_nominal = n_nominal;
        p.q = n_nominal / 4;
        p.nc = (n_nominal + p.q - 1) / p.q;
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
        p.vtxp = vtxp;
        p.headtab = headtab;
        p.dirtab = dirtab;
        p.rden = rden;
        p.dirtab64 = dirtab64;
        p.seg64 = seg64;
        p.dirhash = dirhash;
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
        // F64 keeps 6 VGPRs per ray slot; a mixed-track batch reads every table through per-env pointers: 33 slots on one lane
        // spilled there (564 B of scratch) -- two lanes per env are the better geometry anyway
        const int max_rpl = (dtype == PC_DTYPE_F64 || mixed) ? 17 : 33;
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
    if (e->track_id) {
        if constexpr (RPL <= 17)     // (choose_geometry never gives a mixed-track batch more slots per lane)
            hipLaunchKernelGGL((env_step_kernel<T, RPL, true>), dim3(e->blocks), dim3(256), 0, st, e->params<T>(), actions,
                               reward_scale, obs, reward, term, trunc, gates_passed, final_obs);
    } else
        hipLaunchKernelGGL((env_step_kernel<T, RPL, false>), dim3(e->blocks), dim3(256), 0, st, e->params<T>(), actions,
                           reward_scale, obs, reward, term, trunc, gates_passed, final_obs);
}

// K1f (env_steps_fast_kernel): T successive steps as one launch, where the shape has the table-driven form -- 12 / 16 / 32 nominal rays,
// every track's gather tables inside the LDS limits, a mixed batch in blocks of one track per workgroup; F64 handles: every track inside
// the selector's limits with its rotation table, every env's rotation on it (the conditions of pc_rollout's literal kernels).
// PC_ERR_UNSUPPORTED: no such form for this handle (the caller launches K1).  `table`: stage the 1/den table too (worth it for T > 1).
static int steps_fast_launch(pc_env* e, const int64_t* actions, int64_t T, double reward_scale, float* obs, float* reward, float* term,
                             float* trunc, bool table, hipStream_t st, int32_t* gates_passed = nullptr, float* final_obs = nullptr) {
    const bool f64 = e->dtype == PC_DTYPE_F64;
    const bool rays12 = e->n_nominal == 12 && e->R == 12, rays16 = e->n_nominal == 16 && e->R == 17, rays32 = e->n_nominal == 32 && e->R == 33;
    if (!(rays12 || rays16 || rays32) || !e->opt.fast || T < 1 || T > INT_MAX) return PC_ERR_UNSUPPORTED;
    if ((gates_passed || final_obs) && T != 1) return PC_ERR_INVALID_ARG;      // (the optional outputs are pc_env_step's)
    int max_G = 0, max_nV = 0;
    bool all_nv28 = e->opt.nv28 != 0, all_loops = e->opt.nv28 != 0, tabs = true, all_rden = true;
    int sum_nV = 0;
    for (const TrackHdr& h : e->hdr_host) {
        max_G = std::max(max_G, h.G);
        max_nV = std::max(max_nV, h.nV);
        sum_nV += h.nV;
        tabs = tabs && h.lat_off >= 0 && (!f64 || (h.sel_ok && h.rot_off >= 0));
        all_rden = all_rden && h.rden_off >= 0;
        all_nv28 = all_nv28 && h.nV == 28 && h.n_chain == 26 && h.brk2 == 13 && h.vtxp_off >= 0;   // big_track's layout: two loops of 12 walls
        all_loops = all_loops && h.vtxp_off >= 0 && (h.brk2 == 13 || h.brk2 == 9) && h.n_chain == 2 * h.brk2 && h.nV == 4 * ((h.brk2 + 1) / 2);
    }
    const int epw = e->N <= 32768 ? 128 : 256;      // (one wave per SIMD on twice the workgroups up to 32768 envs, as pc_rollout's big form)
    // two tracks interleaved in evenly split blocks of 64 envs: de-interleaved by wave (the kernel's TWO form: 16 rays, the reference's track layouts)
    const bool two = e->track_id && e->track_block < epw && e->n_tracks == 2 && e->track_bal64 && all_loops && rays16 && e->opt.deinterleave;
    if (!tabs || max_G > TAB_MAX_GATES || max_nV > FT_VTX_MAX || (f64 && e->f64_offgrid) || (e->track_id && e->track_block < epw && !two)) return PC_ERR_UNSUPPORTED;
    size_t lds = (size_t)(256 * e->D + (two ? 256 + 2 * ft_floats(false, true) : ft_floats(false, true))) * sizeof(float);
    const int rden_all = 361 * (two ? sum_nV : max_nV);
    const bool tab = table && all_rden && e->opt.rden != 0 && lds + (size_t)rden_all * sizeof(float) <= 160 * 1024;
    if (tab) lds += (size_t)rden_all * sizeof(float);
    const int ts_floats = two ? ((ft_floats(false, true) + (tab ? 361 * e->hdr_host[0].nV : 0) + 3) & ~3) : 0;
    const int blocks = (int)((e->N + epw - 1) / epw);
    const int vec_ok = ((e->N * e->D) % 4 == 0 && ((uintptr_t)obs & 15) == 0) ? 1 : 0;
    EnvParams<float> prm = e->params<float>();
    prm.lg = 1;
#define PC_STEPS(RPLV, SWPV, TABV, LITV)                                                                                 \
    do {                                                                                                                 \
        static bool attr_set[64] = {false};                                                                              \
        if (e->device >= 64 || !attr_set[e->device]) {                                                                    \
            HIPCHK(hipFuncSetAttribute((const void*)env_steps_fast_kernel<RPLV, SWPV, TABV, LITV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
            if (e->device < 64) attr_set[e->device] = true;                                                                \
        }                                                                                                                \
        hipLaunchKernelGGL((env_steps_fast_kernel<RPLV, SWPV, TABV, LITV>), dim3(blocks), dim3(512), lds, st, prm, actions, (int)T, reward_scale, obs, \
                           reward, term, trunc, epw, vec_ok, gates_passed, final_obs, ts_floats);                        \
    } while (0)
#define PC_STEPS_T(RPLV, SWPV, LITV) do { if (tab) PC_STEPS(RPLV, SWPV, true, LITV); else PC_STEPS(RPLV, SWPV, false, LITV); } while (0)
#define PC_STEPS2(TABV, LITV)                                                                                            \
    do {                                                                                                                 \
        static bool attr_set[64] = {false};                                                                              \
        if (e->device >= 64 || !attr_set[e->device]) {                                                                    \
            HIPCHK(hipFuncSetAttribute((const void*)env_steps_fast_kernel<9, 5, TABV, LITV, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
            if (e->device < 64) attr_set[e->device] = true;                                                                \
        }                                                                                                                \
        hipLaunchKernelGGL((env_steps_fast_kernel<9, 5, TABV, LITV, true>), dim3(blocks), dim3(512), lds, st, prm, actions, (int)T, reward_scale, obs, \
                           reward, term, trunc, epw, vec_ok, gates_passed, final_obs, ts_floats);                        \
    } while (0)
    if (two) { PC_FULL(if (f64) { if (tab) PC_STEPS2(true, true); else PC_STEPS2(false, true); } else { if (tab) PC_STEPS2(true, false); else PC_STEPS2(false, false); }); }
    else if (rays16 && all_nv28) { if (f64) PC_DEV(3, PC_STEPS_T(9, 7, true)); else PC_DEV(0, PC_STEPS_T(9, 7, false)); }
    else if (rays16) { PC_FULL(if (f64) PC_STEPS_T(9, 0, true); else PC_STEPS_T(9, 0, false)); }
    else if (rays12) { PC_FULL(if (f64) PC_STEPS_T(6, 0, true); else PC_STEPS_T(6, 0, false)); }
    else { PC_FULL(if (f64) PC_STEPS_T(17, 0, true); else PC_STEPS_T(17, 0, false)); }
#undef PC_STEPS2
#undef PC_STEPS_T
#undef PC_STEPS
    HIPCHK(hipGetLastError());
    e->last_step_kernel = tab ? PC_STEP_K1F_TABLE : PC_STEP_K1F;
    return PC_OK;
}

// K1 (env_step_kernel): one step of every env, the generic kernel -- any ray count on the menu, any track, per-env track ids
static int step_generic_launch(pc_env* e, const int64_t* actions, double reward_scale, float* obs, float* reward, float* terminated,
                               float* truncated, int32_t* gates_passed, float* final_obs, hipStream_t st) {
    e->last_step_kernel = PC_STEP_K1;
#define PC_CASE(T, M)                                                                                        \
    case M:                                                                                                  \
        launch_step<T, M>(e, actions, reward_scale, obs, reward, terminated, truncated, gates_passed, final_obs, st); \
        break;
    if (e->dtype == PC_DTYPE_F64) {
        switch (e->rpl) {
#ifndef PC_DEV_MIN
            PC_CASE(double, 1) PC_CASE(double, 2) PC_CASE(double, 3) PC_CASE(double, 5) PC_CASE(double, 6)
            PC_CASE(double, 9) PC_CASE(double, 12) PC_CASE(double, 17)
#endif
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
        case PC_ERR_TIMEOUT: return "a peer did not arrive at the gradient exchange (pc_xchg)";
        default: return "unknown error";
    }
}

const char* pc_last_hip_error(void) { return g_hip_err.c_str(); }

int pc_ray_count(int n) {
    g_hip_err.clear();   // (pc_last_hip_error speaks of THIS call)
    if (n < 4 || n > 360) return PC_ERR_INVALID_ARG;
    const int step = 360 / n;
    return (360 + step - 1) / step;  // len(range(0, 360, 360 // n)), car_env.py:269
}

int pc_track_load_json(const char* path, pc_track** out) {
    g_hip_err.clear();   // (pc_last_hip_error speaks of THIS call)
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
    g_hip_err.clear();   // (pc_last_hip_error speaks of THIS call)
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
    g_hip_err.clear();   // (pc_last_hip_error speaks of THIS call)
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
    g_hip_err.clear();   // (pc_last_hip_error speaks of THIS call)
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
    (void)hipFree(e->vtxp);
    (void)hipFree(e->headtab);
    (void)hipFree(e->dirtab);
    (void)hipFree(e->rden);
    (void)hipFree(e->dirtab64);
    (void)hipFree(e->seg64);
    (void)hipFree(e->dirhash);
    (void)hipFree(e->reset_obs);
    delete e;
}

// glibc's cos and sin, each through its own call: an optimiser that sees both of one argument may merge them into sincos(), whose
// cosine differs from cos() in the last place for some arguments -- and the tables below stand for the reference's separate
// np.cos / np.sin calls (car_env.py:426-427, :584)
__attribute__((noinline)) static double libm_cos(double a) { return std::cos(a); }
__attribute__((noinline)) static double libm_sin(double a) { return std::sin(a); }

static int env_create_impl(pc_env* e, const pc_track* const* tracks, const uint8_t* track_id) {
    const bool f64 = e->dtype == PC_DTYPE_F64;
    // ---- host images of the track table
    std::vector<Seg> segs;
    std::vector<Vtx> vtx;
    std::vector<VtxP> vtxp;
    std::vector<F64Dir> dirhash;
    std::vector<double2> headtab;
    std::vector<float2> dirtab;
    std::vector<double2> dirtab64;
    std::vector<SegD> seg64;
    std::vector<double2> vpos;      // host only: the chain vertices' exact positions (indexed like vtx)
    size_t rden_floats = 0;
    e->hdr_host.resize(e->n_tracks);
    e->rot_ids.assign(e->n_tracks, {});
    e->rot_depth.assign(e->n_tracks, {});
    for (int k = 0; k < e->n_tracks; ++k) {
        const pc_track* t = tracks[k];
        TrackHdr& h = e->hdr_host[k];
        h.S = t->n_walls();
        h.G = t->n_gates();
        h.n_scan = 0;
        h.rot_off = -1;     // (F64 handles: set below)
        h.n_rot = 0;
        h.lat_off = -1;
        h.sel_ok = 1;
        h.wall_off = (int)segs.size();
        for (size_t i = 0; i < t->walls.size(); i += 4) segs.push_back(Seg{t->walls[i], t->walls[i + 1], t->walls[i + 2], t->walls[i + 3]});
        h.gate_off = (int)segs.size();
        for (size_t i = 0; i < t->gates.size(); i += 4) segs.push_back(Seg{t->gates[i], t->gates[i + 1], t->gates[i + 2], t->gates[i + 3]});
        // walls as vertex chains: a segment continues the chain iff it starts exactly where the previous ended.  The sweep's
        // float32 coordinates are relative to the ANCHOR = the centre of the vertices' bounding box.
        h.vtx_off = (int)vtx.size();
        {
            double bx0 = 1e300, bx1 = -1e300, by0 = 1e300, by1 = -1e300;
            for (int w = 0; w < h.S; ++w) {
                const Seg& sg = segs[h.wall_off + w];
                bx0 = std::min({bx0, sg.x1, sg.x2}); bx1 = std::max({bx1, sg.x1, sg.x2});
                by0 = std::min({by0, sg.y1, sg.y2}); by1 = std::max({by1, sg.y1, sg.y2});
            }
            h.ax0 = 0.5 * (bx0 + bx1);
            h.ay0 = 0.5 * (by0 + by1);
            h.bx0 = (float)bx0; h.bx1 = (float)bx1; h.by0 = (float)by0; h.by1 = (float)by1;
        }
        // the sweep's view of a wall's closing vertex: its anchor-relative position, the UNIT vector along (x1 - x2, y1 - y2)
        // (car_env.py:171) in float32, and that vector's copy scaled by 2^-40
        const auto edge = [&h](const Seg& sg) {
            const double ex = sg.x1 - sg.x2, ey = sg.y1 - sg.y2, len = std::hypot(ex, ey);
            const float xr = (float)(sg.x2 - h.ax0), yr = (float)(sg.y2 - h.ay0);
            if (len == 0.0) return Vtx{xr, yr, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f};     // a wall without length is never hit: a chain start
            const float fx = (float)(ex / len), fy = (float)(ey / len);
            return Vtx{xr, yr, fx, fy, fx * 0x1p-40f, fy * 0x1p-40f, 0.f, 0.f};
        };
        std::vector<int> wall_k(h.S);      // host only: wall w = the segment closed by chain vertex wall_k[w]
        for (int w = 0; w < h.S; ++w) {
            const Seg& sg = segs[h.wall_off + w];
            const bool cont = w > 0 && segs[h.wall_off + w - 1].x2 == sg.x1 && segs[h.wall_off + w - 1].y2 == sg.y1;
            if (!cont) {
                vtx.push_back(Vtx{(float)(sg.x1 - h.ax0), (float)(sg.y1 - h.ay0), 0.f, 0.f, 1.f, 0.f, 0.f, 0.f});   // chain start: zero edge (scaled copy (1, 0): see Sweep::cand)
                seg64.push_back(SegD{sg.x1, sg.y1, 0.0, 0.0, -1.0, 0, 0});
                vpos.push_back(make_double2(sg.x1, sg.y1));
            }
            vtx.push_back(edge(sg));
            seg64.push_back(SegD{sg.x1, sg.y1, sg.x1 - sg.x2, sg.y1 - sg.y2, -1.0, 0, 0});
            vpos.push_back(make_double2(sg.x2, sg.y2));
            wall_k[w] = (int)vtx.size() - 1 - h.vtx_off;
        }
        h.n_chain = (int)vtx.size() - h.vtx_off;
        while ((vtx.size() - h.vtx_off) % 4) {  // the sweep walks vertex groups of four: pad with chain-start sentinels
            vtx.push_back(Vtx{vtx.back().xr, vtx.back().yr, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f});
            seg64.push_back(SegD{seg64.back().x1, seg64.back().y1, 0.0, 0.0, -1.0, 0, 0});
            vpos.push_back(vpos.back());
        }
        h.nV = (int)vtx.size() - h.vtx_off;
        // F32 mode: the selector's "nothing selected" pattern (SEL_INIT) carries vertex index 0 only while the index takes at
        // most 13 of the candidate's mantissa bits; the float32 coordinates (relative to the track's anchor) and the flag
        // thresholds are priced for a track that fits 2000 px.  F64 mode has neither limit.
        if (h.nV > 65535 || (!f64 && h.nV > 8192)) {
            g_hip_err = "track " + std::to_string(k) + ": " + std::to_string(h.nV) + " chain vertices; dtype f32 takes at most 8192 (use dtype f64)";
            return PC_ERR_UNSUPPORTED;
        }
        if (!f64 && (h.bx1 - h.bx0 > 2000.0f || h.by1 - h.by0 > 2000.0f)) {
            g_hip_err = "track " + std::to_string(k) + ": the walls' bounding box exceeds 2000 px; dtype f32 is priced for tracks that fit (use dtype f64)";
            return PC_ERR_UNSUPPORTED;
        }
        // F64 handles: the float32 SELECTOR (the persistent kernel's literal form, env_step_fast<..., LIT>) runs on a track within
        // those same limits; any other track takes the filter form (env_step_core<double>), which has none.
        const bool sel = !f64 || (h.nV <= 8192 && h.bx1 - h.bx0 <= 2000.0f && h.by1 - h.by0 <= 2000.0f);
        h.sel_ok = sel ? 1 : 0;
        {   // low bits of a sweep candidate that carry the vertex index (at least 5: the unrolled 28-vertex sweep's constant)
            int b = 5;
            while ((1 << b) < h.nV) ++b;
            h.idx_mask = (1u << b) - 1u;
        }
        // chain neighbours and end margins of every segment (SegD::h, SegD::prev_next), bounding box of the vertices
        {
            const int n = h.nV, o = h.vtx_off;
            const auto is_start = [&](int k) { return seg64[o + k].ex == 0.0 && seg64[o + k].ey == 0.0; };
            {   // exactly two chains?  (what the kernels compiled for big_track's layout rely on: TrackHdr::brk2)
                int n_starts = 0, second = -1;
                for (int k = 0; k < h.n_chain; ++k)
                    if (is_start(k) && ++n_starts == 2) second = k;
                h.brk2 = n_starts == 2 ? second : -1;
                h.vtxp_off = -1;
                if (h.brk2 > 0 && h.n_chain == 2 * h.brk2) {     // ... of the same length: the packed copy (VtxP)
                    h.vtxp_off = (int)vtxp.size();
                    for (int i = 0; i < h.brk2; ++i) {
                        const Vtx &a = vtx[o + i], &b = vtx[o + h.brk2 + i];
                        vtxp.push_back(VtxP{{a.xr, b.xr}, {a.yr, b.yr}, {a.ex, b.ex}, {a.ey, b.ey}, {a.exs, b.exs}, {a.eys, b.eys}});
                    }
                }
            }
            // What float32 can get wrong is the ORDER of two hits that lie within its resolution of each other.  The selector
            // keeps 23 - b mantissa bits of a candidate (b index bits): two hits closer than sel_res = 1001 px * 2^-(23 - b) along
            // a ray (beyond 1000 px the reported distance is 1000 either way) may be taken in the wrong order.
            //   * Two walls that share a vertex V at an angle of at least ~13 degrees: such hits lie within sel_res / sin(13 deg) of
            //     V, so a refined hit within `margin` = max(0.05 px, 4.6 sel_res) of a segment's end is compared with the chain
            //     neighbours under the strict test (SegD::h; refine_careful).
            //   * Anything else that brings two walls within `near` = max(0.05 px, 1.5 sel_res) of each other -- walls that cross
            //     or touch without being chain neighbours (a T-junction, an X), a spike sharper than 13 degrees, a wall shorter than
            //     2 margin (its neighbours' neighbours are that close) -- cannot be settled by looking at two neighbours: those
            //     segments carry PC_SEG_SCAN and every ray whose selection lands on one of them is resolved by the float64 scan of
            //     the whole chain under the reference's strict test (car_env.py:178), i.e. exactly.
            int bits = 5;
            while ((1 << bits) < h.nV) ++bits;
            const double sel_res = 1001.0 * std::ldexp(1.0, -(23 - bits));
            const double margin = std::max(0.05, 4.6 * sel_res), near = std::max(0.05, 1.5 * sel_res);
            // (where the selector runs: nV <= 8192 there, so prev / next fit their 15 bits beside PC_SEG_SCAN; the F64 filter form never reads seg64)
            for (int k = 0; k < n && sel; ++k) {
                if (is_start(k)) continue;     // chain starts / padding: no segment (h = -1: |t - 0.5| < h never holds)
                int c0 = k;     // first vertex of this chain, and its last
                while (!is_start(c0)) --c0;
                int c1 = k;
                while (c1 + 1 < h.n_chain && !is_start(c1 + 1)) ++c1;
                const bool closed = c1 > c0 && vpos[o + c0].x == vpos[o + c1].x && vpos[o + c0].y == vpos[o + c1].y;
                const int prev = k - 1 > c0 ? k - 1 : (closed && c1 != k ? c1 : 0);       // shares this segment's first endpoint
                const int next = k + 1 <= c1 ? k + 1 : (closed && c0 + 1 != k ? c0 + 1 : 0);   // shares its second endpoint
                static_assert(PC_SEG_SCAN == 0x8000, "prev in bits 0..14, PC_SEG_SCAN in bit 15, next in bits 16..30");
                seg64[o + k].prev_next = prev | (next << 16);      // (prev, next < nV <= 8192)
                const double len = std::hypot(seg64[o + k].ex, seg64[o + k].ey);
                seg64[o + k].h = 0.5 - margin / len;
                if (len < 2.0 * margin) seg64[o + k].prev_next |= PC_SEG_SCAN;
            }
            if (sel) {
                const auto seg_of = [&](int w) { return segs[h.wall_off + w]; };
                const auto pt_seg = [](double px, double py, const Seg& s) {     // distance of a point from a segment
                    const double ex = s.x2 - s.x1, ey = s.y2 - s.y1, l2 = ex * ex + ey * ey;
                    double t = l2 > 0.0 ? ((px - s.x1) * ex + (py - s.y1) * ey) / l2 : 0.0;
                    t = std::min(1.0, std::max(0.0, t));
                    return std::hypot(px - (s.x1 + t * ex), py - (s.y1 + t * ey));
                };
                const auto orient = [](const Seg& s, double px, double py) { return (s.x2 - s.x1) * (py - s.y1) - (s.y2 - s.y1) * (px - s.x1); };
                for (int a = 0; a < h.S; ++a) {
                    const Seg sa = seg_of(a);
                    const int ka = wall_k[a];
                    if (is_start(ka)) continue;     // (a wall without length is a chain start: never hit)
                    const int pa = seg64[o + ka].prev_next & 0x7fff, na = (int)(((unsigned)seg64[o + ka].prev_next >> 16) & 0x7fff);
                    for (int b = a + 1; b < h.S; ++b) {
                        const Seg sb = seg_of(b);
                        const int kb = wall_k[b];
                        if (is_start(kb)) continue;
                        bool bad;
                        if (pa == kb || na == kb) {
                            // chain neighbours: a spike sharper than ~13 degrees (|sin| < 0.22 with the walls folding back on each other)
                            const double ax = sa.x2 - sa.x1, ay = sa.y2 - sa.y1, bx = sb.x2 - sb.x1, by = sb.y2 - sb.y1;
                            const double la = std::hypot(ax, ay), lb = std::hypot(bx, by);
                            const double sn = std::fabs(ax * by - ay * bx) / (la * lb), cs = (ax * bx + ay * by) / (la * lb);
                            // consecutive walls run head to tail: folding back = their directions nearly opposite
                            bad = sn < 0.22 && cs < 0.0;
                        } else {
                            const double o1 = orient(sa, sb.x1, sb.y1), o2 = orient(sa, sb.x2, sb.y2), o3 = orient(sb, sa.x1, sa.y1), o4 = orient(sb, sa.x2, sa.y2);
                            const bool cross = ((o1 > 0) != (o2 > 0)) && ((o3 > 0) != (o4 > 0));
                            const double d = cross ? 0.0 : std::min({pt_seg(sa.x1, sa.y1, sb), pt_seg(sa.x2, sa.y2, sb), pt_seg(sb.x1, sb.y1, sa), pt_seg(sb.x2, sb.y2, sa)});
                            bad = d < near;
                        }
                        if (bad) {
                            seg64[o + ka].prev_next |= PC_SEG_SCAN;
                            seg64[o + kb].prev_next |= PC_SEG_SCAN;
                        }
                    }
                }
                for (int k = 0; k < n; ++k)
                    if (seg64[o + k].prev_next & PC_SEG_SCAN) { seg64[o + k].h = -1.0; ++h.n_scan; }
            }
            // F64 handles: the literal arithmetic wants the wall's SECOND ENDPOINT as the track file gives it (x1 - ex need not be
            // x2 to the last bit): the records' (ex, ey) fields carry (x2, y2) from here on (lit_fast); a chain start or padding
            // record gets x2 = x1: den == 0, never a hit
            if (f64 && sel) {
                for (int k = 0; k < n; ++k) {
                    SegD& r = seg64[o + k];
                    const bool start = is_start(k);
                    r.ex = start ? r.x1 : vpos[o + k].x;
                    r.ey = start ? r.y1 : vpos[o + k].y;
                }
            }
        }
        if (f64) {
            // F64 mode: every angle an episode can reach (see Math<double>), glibc's cos / sin of it, hashed by the angle's bits
            h.dir_off = -1;
            h.head_off = 0;
            h.rot_off = -1;
            h.n_rot = 0;
            if (e->n_tracks <= 16) {
                std::unordered_set<uint64_t> rots, frontier, keys;
                const auto bits = [](double v) { uint64_t b; std::memcpy(&b, &v, 8); return b; };
                const auto val = [](uint64_t b) { double v; std::memcpy(&v, &b, 8); return v; };
                std::vector<uint64_t> rot_list;                       // index -> rotation (breadth first; index 0 = start_rot: what reset gives)
                std::unordered_map<uint64_t, int>& rid = e->rot_ids[k];
                rid.clear();
                std::vector<int>& depth = e->rot_depth[k];
                int cur_depth = 0;
                const auto add_rot = [&](uint64_t b) {
                    if (!rots.insert(b).second) return false;
                    rid[b] = (int)rot_list.size();
                    rot_list.push_back(b);
                    depth.push_back(cur_depth);
                    return true;
                };
                add_rot(bits(t->start_rot));
                frontier = rots;
                for (int turn = 0; turn < 1000 && !frontier.empty(); ++turn) {      // CarEnv truncates at 1000 steps (car_env.py:749)
                    cur_depth = turn + 1;
                    std::unordered_set<uint64_t> next;
                    for (const uint64_t b : frontier)
                        for (const double w : {val(b) + 5.0, val(b) - 5.0})            // :440-442
                            if (add_rot(bits(w))) next.insert(bits(w));
                    frontier.swap(next);
                }
                {   // the rotation table (Math<double>): row i = the R rays' (cos, sin) at rotation i, then (index of rot - 5.0, index of rot + 5.0), then (rot, -)
                    const int step_deg_ = 360 / e->n_nominal;
                    h.rot_off = (int)dirtab64.size();
                    h.n_rot = (int)rot_list.size();
                    for (const uint64_t b : rot_list) {
                        for (int ray = 0; ray < e->R; ++ray) {
                            const double a = (val(b) + (double)(ray * step_deg_)) * (PC_PI / 180.0);   // np.radians(rot + a), :269, :465
                            dirtab64.push_back(make_double2(libm_cos(a), libm_sin(a)));
                        }
                        const auto lk = rid.find(bits(val(b) - 5.0)), rk = rid.find(bits(val(b) + 5.0));
                        dirtab64.push_back(make_double2(lk == rid.end() ? -1.0 : (double)lk->second, rk == rid.end() ? -1.0 : (double)rk->second));
                        dirtab64.push_back(make_double2(val(b), 0.0));
                    }
                }
                const int step_deg = 360 / e->n_nominal;
                for (const uint64_t b : rots)
                    for (int ray = 0; ray < e->R; ++ray) keys.insert(bits(val(b) + (double)(ray * step_deg)));   // :269, :465
                size_t cap = 1024;
                while (cap < 4 * keys.size()) cap <<= 1;
                std::vector<F64Dir> tab;
                for (;; cap <<= 1) {            // (grown until no probe sequence is longer than the device follows)
                    tab.assign(cap, F64Dir{F64DIR_EMPTY, 0.0, 0.0, 0});
                    bool ok = true;
                    for (const uint64_t k : keys) {
                        size_t slot = f64dir_hash(k) & (cap - 1);
                        int probe = 0;
                        while (tab[slot].key != F64DIR_EMPTY && probe < F64DIR_MAX_PROBE) { slot = (slot + 1) & (cap - 1); ++probe; }
                        if (probe == F64DIR_MAX_PROBE) { ok = false; break; }
                        const double a = val(k) * (PC_PI / 180.0);   // np.radians
                        tab[slot] = F64Dir{k, libm_cos(a), libm_sin(a), 0};
                    }
                    if (ok) break;
                }
                h.dir_off = (int)dirhash.size();
                h.head_off = (int)(cap - 1);
                dirhash.insert(dirhash.end(), tab.begin(), tab.end());
            }
            if (dirtab64.empty()) dirtab64.push_back(make_double2(0.0, 0.0));
            // the selector's float32 direction lattice (start_rot + j degrees: every angle rot + a is one of them mod 360; the float32
            // direction only selects, so that cos of the unreduced angle differs in float64's last places does not matter)
            h.lat_off = (int)dirtab.size();
            for (int j = 0; j < 360; ++j) {
                const double a = (t->start_rot + (double)j) * (PC_PI / 180.0);
                dirtab.push_back(make_float2((float)libm_cos(a), (float)libm_sin(a)));
            }
            dirtab.push_back(make_float2(0.f, 0.f));
        } else {
            h.dir_off = (int)dirtab.size();
            h.lat_off = h.dir_off;
            for (int j = 0; j < 360; ++j) {  // direction lattice: start_rot + j degrees, np.radians then libm cos/sin
                const double a = (t->start_rot + (double)j) * (PC_PI / 180.0);
                dirtab.push_back(make_float2((float)libm_cos(a), (float)libm_sin(a)));
                dirtab64.push_back(make_double2(libm_cos(a), libm_sin(a)));
            }
            dirtab.push_back(make_float2(0.f, 0.f));
            dirtab64.push_back(make_double2(0.0, 0.0));
        }
        // the float32 1/den table [361][nV] exists only where a kernel can stage it in LDS: chains of at most FT_VTX_MAX vertices, and on F64
        // handles only for tracks the selector may run on (a 65535-vertex float64 track would cost 95 MB it can never read)
        const bool want_rden = h.nV <= FT_VTX_MAX && (!f64 || h.sel_ok);
        if (rden_floats + (size_t)361 * h.nV > (size_t)INT_MAX) {
            g_hip_err = "pc_env_create: the tracks' 1/den tables exceed 2^31 floats";
            return PC_ERR_UNSUPPORTED;
        }
        h.rden_off = want_rden ? (int)rden_floats : -1;
        if (want_rden) rden_floats += (size_t)361 * h.nV;
        if (!f64) h.head_off = (int)headtab.size();
        h.start_collides = 0;
        h.start_x = t->start_x;
        h.start_y = t->start_y;
        h.start_rot = t->start_rot;
        for (int j = 0; j < 72; ++j) {  // heading grid: start_rot + 5 j degrees, np.radians then libm cos/sin
            const double a = (t->start_rot + 5.0 * j) * (PC_PI / 180.0);
            headtab.push_back(make_double2(libm_cos(a), libm_sin(a)));
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
        for (int blk = 64; blk >= 32; blk >>= 1) {
            bool ok = e->n_tracks == 2 && N % blk == 0;
            for (size_t b = 0; b < N && ok; b += blk) {
                int ones = 0;
                for (int i = 0; i < blk; ++i) ones += track_id[b + i] == 1, ok = ok && track_id[b + i] < 2;
                ok = ok && ones == blk / 2;
            }
            (blk == 64 ? e->track_bal64 : e->track_bal32) = ok;
        }
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
    static_assert(sizeof(VtxP) == 48, "packed vertex records: 48 bytes");
    if (!vtxp.empty()) {
        HIPCHK(hipMalloc((void**)&e->vtxp, vtxp.size() * sizeof(VtxP)));
        HIPCHK(hipMemcpy(e->vtxp, vtxp.data(), vtxp.size() * sizeof(VtxP), hipMemcpyHostToDevice));
    }
    HIPCHK(hipMalloc((void**)&e->headtab, headtab.size() * sizeof(double2)));
    HIPCHK(hipMemcpy(e->headtab, headtab.data(), headtab.size() * sizeof(double2), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void**)&e->dirtab, dirtab.size() * sizeof(float2)));
    HIPCHK(hipMemcpy(e->dirtab, dirtab.data(), dirtab.size() * sizeof(float2), hipMemcpyHostToDevice));
    static_assert(sizeof(SegD) == 48, "refinement table: 48 bytes per chain vertex");
    HIPCHK(hipMalloc((void**)&e->dirtab64, dirtab64.size() * sizeof(double2)));
    HIPCHK(hipMemcpy(e->dirtab64, dirtab64.data(), dirtab64.size() * sizeof(double2), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void**)&e->seg64, seg64.size() * sizeof(SegD)));
    HIPCHK(hipMemcpy(e->seg64, seg64.data(), seg64.size() * sizeof(SegD), hipMemcpyHostToDevice));
    if (!dirhash.empty()) {
        HIPCHK(hipMalloc((void**)&e->dirhash, dirhash.size() * sizeof(F64Dir)));
        HIPCHK(hipMemcpy(e->dirhash, dirhash.data(), dirhash.size() * sizeof(F64Dir), hipMemcpyHostToDevice));
    }
    {
        HIPCHK(hipMalloc((void**)&e->rden, (rden_floats ? rden_floats : 1) * sizeof(float)));
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
    g_hip_err.clear();
    pc_env* e = new (std::nothrow) pc_env;
    if (!e) return PC_ERR_INVALID_ARG;
    e->device = device;
    e->dtype = dtype;
    e->N = n_envs;
    e->n_nominal = num_rays_nominal;
    e->R = pc_ray_count(num_rays_nominal);
    e->D = 6 + e->R;
    e->n_tracks = n_tracks;
    e->mixed = track_id != nullptr;
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
    g_hip_err.clear();   // (pc_last_hip_error speaks of THIS call)
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

int pc_env_track_info(const pc_env* e, int track, int* n_walls, int* n_chain_vertices, int* n_scan_segments) {
    g_hip_err.clear();
    if (!e || track < 0 || track >= e->n_tracks) return PC_ERR_INVALID_ARG;
    const TrackHdr& h = e->hdr_host[track];
    if (n_walls) *n_walls = h.S;
    if (n_chain_vertices) *n_chain_vertices = h.n_chain;
    if (n_scan_segments) *n_scan_segments = (e->dtype == PC_DTYPE_F32 || h.sel_ok) ? h.n_scan : 0;
    return PC_OK;
}

int pc_env_last_rollout_kernel(const pc_env* e) { return e ? e->last_kernel : PC_ERR_INVALID_ARG; }

int pc_env_launch_info(const pc_env* e, int* lanes_per_env, int* rays_per_lane, int* blocks, int* threads) {
    g_hip_err.clear();   // (pc_last_hip_error speaks of THIS call)
    if (!e) return PC_ERR_INVALID_ARG;
    if (lanes_per_env) *lanes_per_env = 1 << e->lg;
    if (rays_per_lane) *rays_per_lane = e->rpl;
    if (blocks) *blocks = e->blocks;
    if (threads) *threads = 256;
    return PC_OK;
}

int pc_env_reset(pc_env* e, float* obs, void* stream) {
    g_hip_err.clear();   // (pc_last_hip_error speaks of THIS call)
    if (!e) return PC_ERR_INVALID_ARG;
    DeviceGuard guard(e->device);
    if (!guard.ok) return PC_ERR_NO_DEVICE;
    const int blocks = (int)((e->N + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    e->f64_offgrid = false;
    if (e->dtype == PC_DTYPE_F64)
        hipLaunchKernelGGL(env_reset_kernel<double>, dim3(blocks), dim3(256), 0, st, e->params<double>(), obs);
    else
        hipLaunchKernelGGL(env_reset_kernel<float>, dim3(blocks), dim3(256), 0, st, e->params<float>(), obs);
    HIPCHK(hipGetLastError());
    return PC_OK;
}

int pc_env_step(pc_env* e, const int64_t* actions, double reward_scale, float* obs, float* reward, float* terminated,
                float* truncated, int32_t* gates_passed, float* final_obs, void* stream) {
    g_hip_err.clear();   // (pc_last_hip_error speaks of THIS call)
    if (!e || !actions || !obs || !reward || !terminated || !truncated) return PC_ERR_INVALID_ARG;
    DeviceGuard guard(e->device);
    if (!guard.ok) return PC_ERR_NO_DEVICE;
    hipStream_t st = (hipStream_t)stream;
    // the table-driven form (K1f) where the handle has one
    if (e->opt.step_form == 2 || (e->opt.step_form == 0 && e->N >= PC_STEP_FAST_MIN_ENVS)) {
        const int rc = steps_fast_launch(e, actions, 1, reward_scale, obs, reward, terminated, truncated, false, st, gates_passed, final_obs);
        if (rc != PC_ERR_UNSUPPORTED) return rc;
    }
    return step_generic_launch(e, actions, reward_scale, obs, reward, terminated, truncated, gates_passed, final_obs, st);
}

int pc_env_step_many(pc_env* e, const int64_t* actions, int64_t T, double reward_scale, float* obs, float* reward, float* terminated,
                     float* truncated, void* stream) {
    g_hip_err.clear();   // (pc_last_hip_error speaks of THIS call)
    if (!e || !actions || !obs || !reward || !terminated || !truncated || T < 1) return PC_ERR_INVALID_ARG;
    DeviceGuard guard(e->device);
    if (!guard.ok) return PC_ERR_NO_DEVICE;
    if (e->opt.step_form != 1) {      // (one launch instead of T: worth it at any batch size)
        const int rc = steps_fast_launch(e, actions, T, reward_scale, obs, reward, terminated, truncated, T > 1, (hipStream_t)stream);
        if (rc != PC_ERR_UNSUPPORTED) return rc;
    }
    // no table-driven form for this handle: the same T steps as T launches of K1, row by row
    int rc = PC_OK;
    for (int64_t t = 0; t < T && rc == PC_OK; ++t)
        rc = step_generic_launch(e, actions + t * e->N, reward_scale, obs + t * e->N * e->D, reward + t * e->N, terminated + t * e->N, truncated + t * e->N,
                                 nullptr, nullptr, (hipStream_t)stream);
    return rc;
}

int pc_env_last_step_kernel(const pc_env* e) { return e ? e->last_step_kernel : PC_ERR_INVALID_ARG; }

int pc_env_info(pc_env* e, int32_t* gates_passed, int32_t* time_passed, void* stream) {
    g_hip_err.clear();   // (pc_last_hip_error speaks of THIS call)
    if (!e) return PC_ERR_INVALID_ARG;
    DeviceGuard guard(e->device);
    if (!guard.ok) return PC_ERR_NO_DEVICE;
    hipLaunchKernelGGL(env_info_kernel, dim3((unsigned)((e->N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, e->iv, e->N, gates_passed,
                       time_passed);
    HIPCHK(hipGetLastError());
    return PC_OK;
}

#ifdef PC_DEV_MIN
int pc_build_ablate(void) { return 0x100 | PC_ABLATE; }     // a developer quick build is refused by the host layer like an ablation build
#else
int pc_build_ablate(void) { return PC_ABLATE; }
#endif

#ifdef PC_STAMPS
// developer build only (not declared in ppocar.h): copy the phase stamps of the last pc_rollout launch to the host
int pc_debug_read_stamps(unsigned long long* out, int n) {
    g_hip_err.clear();   // (pc_last_hip_error speaks of THIS call)
    const int total = 8 * STAMP_NT * STAMP_NPH;
    if (n < total) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), total * sizeof(unsigned long long)) != hipSuccess) return -3;
    return total;
}
int pc_debug_read_stamps_u(unsigned long long* out) {
    g_hip_err.clear();   // (pc_last_hip_error speaks of THIS call)
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps_u), 16 * sizeof(unsigned long long)) != hipSuccess) return -3;
    return 16;
}
#endif

int pc_env_get_state(pc_env* e, double* px, double* py, double* vx, double* vy, double* rot, int64_t* time_step,
                     int64_t* next_gate, int64_t* passed) {
    g_hip_err.clear();   // (pc_last_hip_error speaks of THIS call)
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
    g_hip_err.clear();   // (pc_last_hip_error speaks of THIS call)
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
        if (rot && f64) {      // the rotation's row of the track's rotation table, or -1: a value no episode reaches (the kernels then hash / evaluate it)
            uint64_t b;
            std::memcpy(&b, &rot[i], 8);
            const auto& ids = e->rot_ids[tid[i]];
            const auto it = ids.find(b);
            iv[i].x = it == ids.end() ? -1 : it->second;
        }
        if (time_step) iv[i].y = (int)time_step[i];
        if (next_gate) {
            if (next_gate[i] < 0 || next_gate[i] >= e->hdr_host[tid[i]].G) return PC_ERR_INVALID_ARG;
            iv[i].z = (int)next_gate[i];
        }
        if (passed) iv[i].w = (int)passed[i];
    }
    bool offgrid = false;
    if (f64) {      // can every env's episode stay inside its track's rotation table?  (what the selector kernel needs: rollout_f64_impl)
        for (size_t i = 0; i < N && !offgrid; ++i) {
            const std::vector<int>& depth = e->rot_depth[tid[i]];
            const int id = iv[i].x;
            offgrid = id < 0 || id >= (int)depth.size() || depth[id] > iv[i].y;
        }
        e->f64_offgrid = true;      // (until the copies below have succeeded: a half-written state must not reach the literal kernels)
    }
    HIPCHK(hipMemcpy(e->iv, iv.data(), N * sizeof(int4), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->pv, pv.data(), 4 * N * sizeof(double), hipMemcpyHostToDevice));
    if (f64 && rot) HIPCHK(hipMemcpy(e->rot, rot, N * sizeof(double), hipMemcpyHostToDevice));
    if (f64) e->f64_offgrid = offgrid;
    return PC_OK;
}

int pc_gae(int device, const float* rew, const float* val, const float* term, const float* trunc, const float* last_val,
           const float* last_term, const float* last_trunc, double gamma, double lam, int64_t T, int64_t N, float* adv,
           float* ret, void* stream) {
    g_hip_err.clear();   // (pc_last_hip_error speaks of THIS call)
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
    g_hip_err.clear();   // (pc_last_hip_error speaks of THIS call)
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
// Batches up to this size take the forms that cut the work of 32 envs over a whole workgroup (policy_kernel<SPLIT>,
// rollout_small_kernel): n_envs / 32 workgroups, so 16384 envs are two rounds of 256 -- about what the 128-env big form
// needs for anything up to 32768 envs.  The same bound for both kernels keeps the default per-step and persistent paths
// bit-identical.
// Up to this many envs the small form (32 or 16 envs per workgroup, the policy's hidden tiles split over the waves) beats 128-env
// workgroups of independent waves: its 256 workgroups of 32 envs fill the chip once; one env more starts a second round of them and the
// step takes 11.8 us where the big form takes 9.8 (tools/form_sweep.py, profiles/r5_form_sweep.txt: 16384 until round 5)
#define PC_SPLIT_MAX_ENVS 8192
// ... and up to this many the 16-envs-per-wave form of the big kernels (rollout_kernel<..., LGE = 2>: two waves per SIMD where 32-env
// waves leave one) beats them; above, 32-env waves come in pairs themselves
#define PC_MEDIUM_MAX_ENVS 32768
static const int64_t g_rollout_epw128_max = 32768;  // big form at or below this many envs: 128 envs (4 waves) per workgroup

// the arithmetic form a (D, A) shape gets when `requested` is asked for: the split forms cover D <= 40, A <= 9
static int policy_prec(int requested, int D, int A) { return (requested >= 1 && D <= 40 && A <= 9) ? requested : 0; }

// One policy step's configuration: shape, arithmetic form, work decomposition.  Immutable after creation; any number of
// handles with different forms can live in one process (each weight image belongs to the handle that packed it).
struct pc_policy {
    int device = 0, D = 0, H = 0, A = 0;
    int precision = 0;   // the form the shape actually gets (policy_prec)
    int split = -1;      // -1 automatic (by batch size), 0 never, 1 always
};

extern "C" {

int pc_policy_create(int device, int D, int H, int A, int precision, int split, pc_policy** out) {
    g_hip_err.clear();   // (pc_last_hip_error speaks of THIS call)
    if (!out || precision < -1 || precision > 2 || split < -1 || split > 1) return PC_ERR_INVALID_ARG;
    if (H != 256 || A < 1 || A > 15 || D < 1 || D > 40) return PC_ERR_UNSUPPORTED;  // the caller falls back to its own GEMMs
    pc_policy* p = new (std::nothrow) pc_policy;
    if (!p) return PC_ERR_INVALID_ARG;
    p->device = device;
    p->D = D;
    p->H = H;
    p->A = A;
    p->precision = policy_prec(precision < 0 ? kDefaultPolicyPrecision : precision, D, A);
    p->split = split;
    *out = p;
    return PC_OK;
}

void pc_policy_destroy(pc_policy* p) { delete p; }

int pc_policy_get(const pc_policy* p, int* precision, int* split, int64_t* image_floats) {
    g_hip_err.clear();   // (pc_last_hip_error speaks of THIS call)
    if (!p) return PC_ERR_INVALID_ARG;
    if (precision) *precision = p->precision;
    if (split) *split = p->split;
    if (image_floats) *image_floats = p->precision ? polx_image_dwords(p->precision, pol_ng(policy_ks(p->D))) : pol_image_padded(policy_ks(p->D));
    return PC_OK;
}

int pc_env_set_option(pc_env* e, int option, int value) {
    g_hip_err.clear();   // (pc_last_hip_error speaks of THIS call)
    if (!e) return PC_ERR_INVALID_ARG;
    switch (option) {
        case PC_OPT_ROLLOUT_FORM:
            if (value < -1 || value > 4) return PC_ERR_INVALID_ARG;
            e->opt.rden = (value == 2 || value == 3) ? 0 : 1;
            e->opt.form = (value == 2 || value == 3) ? value - 2 : value;      // (4: the 16-envs-per-wave form, where the shape has one)
            return PC_OK;
        case PC_OPT_ROLLOUT_EPW:
            if (value != 0 && value != 16 && value != 32 && value != 128 && value != 256) return PC_ERR_INVALID_ARG;
            e->opt.epw_override = value;
            return PC_OK;
        case PC_OPT_ROLLOUT_FAST:
            if (value < 0 || value > 3) return PC_ERR_INVALID_ARG;
            e->opt.fast = value != 0;
            e->opt.nv28 = value == 1 || value == 3;
            e->opt.deinterleave = value != 3;
            return PC_OK;
        case PC_OPT_STEP_FORM:
            if (value < 0 || value > 2) return PC_ERR_INVALID_ARG;
            e->opt.step_form = value;
            return PC_OK;
        default: return PC_ERR_INVALID_ARG;
    }
}

int pc_env_get_option(const pc_env* e, int option, int* value) {
    g_hip_err.clear();   // (pc_last_hip_error speaks of THIS call)
    if (!e || !value) return PC_ERR_INVALID_ARG;
    switch (option) {
        case PC_OPT_ROLLOUT_FORM: *value = e->opt.form < 0 ? -1 : e->opt.form + (e->opt.rden ? 0 : 2); return PC_OK;
        case PC_OPT_ROLLOUT_EPW: *value = e->opt.epw_override; return PC_OK;
        case PC_OPT_ROLLOUT_FAST: *value = !e->opt.fast ? 0 : (!e->opt.nv28 ? 2 : (e->opt.deinterleave ? 1 : 3)); return PC_OK;
        case PC_OPT_STEP_FORM: *value = e->opt.step_form; return PC_OK;
        default: return PC_ERR_INVALID_ARG;
    }
}

}  // extern "C"

static int64_t policy_image_floats_impl(int prec, int D) { return prec ? polx_image_dwords(prec, pol_ng(policy_ks(D))) : pol_image_padded(policy_ks(D)); }

static int policy_pack_impl(int device, int prec, int D, int H, int A, const float* aW1, const float* ab1, const float* aW2, const float* ab2,
                            const float* cW1, const float* cb1, const float* cW2, const float* cb2, float* image, int* status, void* stream) {
    if (!aW1 || !ab1 || !aW2 || !ab2 || !cW1 || !cb1 || !cW2 || !cb2 || !image) return PC_ERR_INVALID_ARG;
    if (H != 256 || A < 1 || A > 15 || D < 1 || D > 40) return PC_ERR_UNSUPPORTED;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1 || device < 0 || device >= count) return PC_ERR_NO_DEVICE;
    DeviceGuard guard(device);
    if (!guard.ok) return PC_ERR_NO_DEVICE;
    if (status) HIPCHK(hipMemsetAsync(status, 0, sizeof(int), (hipStream_t)stream));     // (forms without scaled domains leave it 0: no operand of theirs saturates)
#define PC_PACK(PRC, NGV)                                                                                                \
    hipLaunchKernelGGL((policy_pack16_kernel<PRC, NGV>), dim3(64), dim3(256), 0, (hipStream_t)stream, D, A, aW1, ab1, aW2, ab2, cW1, \
                       cb1, cW2, cb2, reinterpret_cast<unsigned*>(image), status)
    const int ng = pol_ng(policy_ks(D));
    if (prec == 1) { PC_FULL(if (ng == 5) PC_PACK(1, 5); else PC_PACK(1, 3)); }
    else if (prec == 2) { if (ng == 5) PC_PACK(2, 5); else PC_PACK(2, 3); }
#undef PC_PACK
    else
        PC_FULL(hipLaunchKernelGGL(policy_pack_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, policy_ks(D), D, A, aW1, ab1, aW2, ab2,
                                   cW1, cb1, cW2, cb2, image));
    HIPCHK(hipGetLastError());
    return PC_OK;
}

static int policy_act_impl(int device, int prec, int split_mode, const float* obs, int64_t N, int D, int H, int A, const float* image, uint64_t seed,
                           uint64_t offset, const uint64_t* offset_dev, int64_t* action, float* action_f32, float* logprob, float* value,
                           float* logits_out, void* stream) {
    if (!obs || !image || !action || !logprob || !value || N < 1) return PC_ERR_INVALID_ARG;
    if (H != 256 || A < 1 || A > 15 || D < 1 || D > 40) return PC_ERR_UNSUPPORTED;  // the caller falls back to its own GEMMs
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1 || device < 0 || device >= count) return PC_ERR_NO_DEVICE;
    DeviceGuard guard(device);
    if (!guard.ok) return PC_ERR_NO_DEVICE;
    const int KS = policy_ks(D);
    const size_t lds = (size_t)((prec ? polx_image_dwords(prec, pol_ng(KS)) : pol_image_padded(KS)) + 8 * 32 * 20) * sizeof(float);
    static int n_cu[64] = {0};
    if (device < 64 && n_cu[device] == 0) {
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, device));
        n_cu[device] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int cus = device < 64 ? n_cu[device] : 256;
    // too few 256-env workgroups to fill the chip: split the hidden tiles over the waves instead
    const bool split = split_mode < 0 ? N <= PC_SPLIT_MAX_ENVS : split_mode == 1;
    const int64_t chunks = split ? (N + 31) / 32 : (N + 255) / 256;
    const int blocks = (int)(chunks < cus ? chunks : cus);  // one ~100-KB-LDS workgroup per CU, persistent over env chunks
    hipStream_t st = (hipStream_t)stream;
#define PC_POL(KSV, SPL, PRC)                                                                                            \
    do {                                                                                                                 \
        static bool attr_set[64] = {false};                                                                              \
        if (device >= 64 || !attr_set[device]) {                                                                         \
            HIPCHK(hipFuncSetAttribute((const void*)policy_kernel<KSV, SPL, PRC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
            if (device < 64) attr_set[device] = true;                                                                    \
        }                                                                                                                \
        hipLaunchKernelGGL((policy_kernel<KSV, SPL, PRC>), dim3(blocks), dim3(512), lds, st, obs, N, D, A, image, seed, offset, offset_dev, \
                           action, action_f32, logprob, value, logits_out);                                              \
    } while (0)
    if (prec == 1) {
        PC_FULL(if (split) { if (KS == 5) PC_POL(5, true, 1); else if (KS == 6) PC_POL(6, true, 1); else PC_POL(10, true, 1); }
                else { if (KS == 5) PC_POL(5, false, 1); else if (KS == 6) PC_POL(6, false, 1); else PC_POL(10, false, 1); });
    } else if (prec == 2) {
        if (split) { if (KS == 5) PC_FULL(PC_POL(5, true, 2)); else if (KS == 6) PC_POL(6, true, 2); else PC_POL(10, true, 2); }
        else { if (KS == 5) PC_FULL(PC_POL(5, false, 2)); else if (KS == 6) PC_POL(6, false, 2); else PC_POL(10, false, 2); }
    } else if (split) {
        PC_FULL(if (KS == 5) PC_POL(5, true, 0);
                else if (KS == 6) PC_POL(6, true, 0);
                else PC_POL(10, true, 0));
    } else {
        PC_FULL(if (KS == 5) PC_POL(5, false, 0);
                else if (KS == 6) PC_POL(6, false, 0);
                else PC_POL(10, false, 0));
    }
#undef PC_POL
    HIPCHK(hipGetLastError());
    return PC_OK;
}

int pc_policy_pack(const pc_policy* p, const float* aW1, const float* ab1, const float* aW2, const float* ab2, const float* cW1,
                   const float* cb1, const float* cW2, const float* cb2, float* image, void* stream) {
    g_hip_err.clear();   // (pc_last_hip_error speaks of THIS call)
    if (!p) return PC_ERR_INVALID_ARG;
    return policy_pack_impl(p->device, p->precision, p->D, p->H, p->A, aW1, ab1, aW2, ab2, cW1, cb1, cW2, cb2, image, nullptr, stream);
}

int pc_policy_pack_checked(const pc_policy* p, const float* aW1, const float* ab1, const float* aW2, const float* ab2, const float* cW1,
                           const float* cb1, const float* cW2, const float* cb2, float* image, int32_t* range_status, void* stream) {
    g_hip_err.clear();   // (pc_last_hip_error speaks of THIS call)
    if (!p || !range_status) return PC_ERR_INVALID_ARG;
    return policy_pack_impl(p->device, p->precision, p->D, p->H, p->A, aW1, ab1, aW2, ab2, cW1, cb1, cW2, cb2, image, range_status, stream);
}

int pc_policy_act(const pc_policy* p, const float* obs, int64_t N, const float* image, uint64_t seed, uint64_t offset,
                  const uint64_t* offset_dev, int64_t* action, float* action_f32, float* logprob, float* value, float* logits_out,
                  void* stream) {
    g_hip_err.clear();   // (pc_last_hip_error speaks of THIS call)
    if (!p) return PC_ERR_INVALID_ARG;
    return policy_act_impl(p->device, p->precision, p->split, obs, N, p->D, p->H, p->A, image, seed, offset, offset_dev, action, action_f32,
                           logprob, value, logits_out, stream);
}

int pc_ppo_gather(int device, const int64_t* idx, int B, int D, const float* obs, const float* act, const float* logprob,
                  const float* adv, const float* ret, float* o_obs, float* o_act, float* o_logprob, float* o_adv, float* o_ret,
                  void* stream) {
    g_hip_err.clear();   // (pc_last_hip_error speaks of THIS call)
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
    g_hip_err.clear();   // (pc_last_hip_error speaks of THIS call)
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
    g_hip_err.clear();   // (pc_last_hip_error speaks of THIS call)
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
    g_hip_err.clear();   // (pc_last_hip_error speaks of THIS call)
    if (!param || !grad || !exp_avg || !exp_avg_sq || !step_count || !lr_dev || n < 1 || n > (1 << 26)) return PC_ERR_INVALID_ARG;
    DeviceGuard guard(device);
    if (!guard.ok) return PC_ERR_NO_DEVICE;
    hipLaunchKernelGGL(clip_adam_mb_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, param, const_cast<float*>(grad),
                       exp_avg, exp_avg_sq, step_count, lr_dev, (int)n, (float)max_norm, (float)grad_scale, (float)beta1, (float)beta2, (float)eps);
    HIPCHK(hipGetLastError());
    return PC_OK;
}

// pc_rollout for PC_DTYPE_F64 handles: K9d, rollout_f64_kernel (the bit-exact env inside the persistent launch).  Discrete(9), the
// split-operand policy forms, 12 or 16 nominal rays (6 / 9 ray slots per lane); anything else is
// PC_ERR_UNSUPPORTED and runs through the per-step kernels, which fill the same buffers bit for bit.
static int rollout_f64_impl(pc_env* e, int prec_request, const float* image, int A, int64_t T, double reward_scale, uint64_t seed, uint64_t offset,
                            const uint64_t* offset_dev, float* obs_buf, float* act_buf, float* rew_buf, float* val_buf, float* term_buf,
                            float* trunc_buf, float* logprob_buf, float* next_obs, float* next_term, float* next_trunc, float* last_value,
                            float* reward_sum, void* stream) {
    if (A != 9) return PC_ERR_UNSUPPORTED;
    // mixed tracks interleaved inside a wave (no aligned block of 32 envs on one track): the generic kernel K9d, whose env step runs once
    // per distinct track id of a wave; the literal forms stage ONE track's tables per workgroup
    const bool interleaved = e->track_id && !e->track_blocks32;
    const int KS = policy_ks(e->D);
    const int prec = policy_prec(prec_request, e->D, A);
    DeviceGuard guard(e->device);
    if (!guard.ok) return PC_ERR_NO_DEVICE;
    const int rpl = (e->R + 1) / 2;
    const int epw = e->opt.epw_override >= 128 ? e->opt.epw_override : (e->N <= g_rollout_epw128_max ? 128 : 256);
    const int blocks = (int)((e->N + epw - 1) / epw);
    hipStream_t st = (hipStream_t)stream;
    // ---- the SELECTOR form: K9 itself (rollout_kernel<..., LIT>) -- the float32 sweep picks each ray's wall, the reference's literal
    // float64 arithmetic measures it (env_step_fast's literal form, lit_fast / lit_careful).  What it needs: every track inside the
    // selector's limits with its rotation table built, every env's rotation a row of that table for the rest of its episode
    // (f64_offgrid), the fast modes' shape (12, 16 or 32 nominal rays, fp16 x 2 policy arithmetic, PC_OPT_ROLLOUT_FAST not 0) and LDS
    // for the tables and (12 / 16 rays) the 1/den table.  The sweep is chosen as for F32 handles: chain-packed for two equal loops of 13 vertices
    // (big_track.json) or, mixed, of 13 or 9; the generic sweeps for any other track.  Anything else: the filter form below.
    {
        int max_G = 0, max_nV = 0;
        bool all_nv28 = e->opt.nv28 != 0, all_loops = e->opt.nv28 != 0, tabs = true;
        for (const TrackHdr& h : e->hdr_host) {
            max_G = std::max(max_G, h.G);
            max_nV = std::max(max_nV, h.nV);
            tabs = tabs && h.sel_ok && h.rot_off >= 0 && h.lat_off >= 0;
            all_nv28 = all_nv28 && h.nV == 28 && h.n_chain == 26 && h.brk2 == 13 && h.vtxp_off >= 0;
            all_loops = all_loops && h.vtxp_off >= 0 && (h.brk2 == 13 || h.brk2 == 9) && h.n_chain == 2 * h.brk2 && h.nV == 4 * ((h.brk2 + 1) / 2);
        }
        const int img = polx_image_dwords(prec, pol_ng(KS));
        const bool rays16 = KS == 6 && rpl == 9 && e->n_nominal == 16, rays12 = KS == 5 && rpl == 6 && e->n_nominal == 12;
        const bool rays33 = KS == 10 && rpl == 17 && e->n_nominal == 32;
        int rden_lds = rays33 ? 0 : 361 * max_nV;      // (33 rays: no room for the table -- the sweep forms 1/den itself, as for F32 handles)
        size_t lds_sel = (size_t)k9_fast_lds_floats(img, 32, e->D, !rays33, rden_lds) * sizeof(float);
        if (rden_lds && lds_sel > 160 * 1024) {        // ... and so does a 12 / 16-ray track whose table does not fit (more than ~36 chain vertices)
            rden_lds = 0;
            lds_sel = (size_t)k9_fast_lds_floats(img, 32, e->D, true, 0) * sizeof(float);
        }
        if (prec == 0) {
            // THE STRICTEST CELL: float64 env (the literal form) AND the policy GEMMs as the exact fp32 chain (v_mfma_f32_16x16x4_f32) -- every
            // number of the rollout in the reference's own arithmetic -- as one persistent launch: K9's literal form with the fp32 weight
            // image (16 -> 17 rays, the big form, the generic sweeps; the fp32 image leaves no room for the 1/den table, as for F32
            // handles).  Other shapes: the per-step kernels.
            const int img0 = pol_image_padded(KS);
            const size_t lds0 = (size_t)k9_fast_lds_floats(img0, 32, e->D, true, 0) * sizeof(float);
            if (!(tabs && rays16 && !e->f64_offgrid && !interleaved && e->D >= 17 && max_G <= TAB_MAX_GATES && max_nV <= FT_VTX_MAX && e->opt.fast &&
                  (!e->track_id || e->track_block >= epw) && lds0 <= 160 * 1024))
                return PC_ERR_UNSUPPORTED;
            const int vec_ok = ((e->N * e->D) % 4 == 0 && (((uintptr_t)obs_buf | (uintptr_t)next_obs) & 15) == 0) ? 1 : 0;
            EnvParams<float> prm = e->params<float>();
            prm.lg = 1;
#ifndef PC_DEV_MIN
            static bool attr_set[64] = {false};
            if (e->device >= 64 || !attr_set[e->device]) {
                HIPCHK(hipFuncSetAttribute((const void*)rollout_kernel<6, 9, 0, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                if (e->device < 64) attr_set[e->device] = true;
            }
#endif
            PC_FULL(hipLaunchKernelGGL((rollout_kernel<6, 9, 0, 1, true>), dim3(blocks), dim3(512), lds0, st, prm, image, A, (int)T, reward_scale, seed, offset,
                                       offset_dev, obs_buf, act_buf, rew_buf, val_buf, term_buf, trunc_buf, logprob_buf, next_obs, next_term, next_trunc,
                                       0, epw, vec_ok, last_value, reward_sum));
            HIPCHK(hipGetLastError());
            e->last_kernel = PC_KERNEL_K9_LITERAL;
            return PC_OK;
        }
        // tracks interleaved inside the waves: the two-track fast form of F32 handles (rollout_impl) with the literal env step
        if (interleaved && prec == 2 && tabs && rays16 && !e->f64_offgrid && e->opt.fast && e->opt.nv28 != 0 && e->n_tracks == 2 && all_loops &&
            max_G <= TAB_MAX_GATES && max_nV <= FT_VTX_MAX && e->D >= 17) {
            const int ts6 = (ft_floats(false, true) + 3) & ~3;
            const size_t lds6 = (size_t)k9_fast_lds_floats(img, 32, e->D, true, ts6) * sizeof(float);
            if (lds6 <= 160 * 1024 && (e->opt.form == 4 || (e->opt.form < 0 && e->opt.epw_override == 0 && e->N > PC_SPLIT_MAX_ENVS && e->N <= PC_MEDIUM_MAX_ENVS))) {
                const int vec6 = ((e->N * e->D) % 4 == 0 && (((uintptr_t)obs_buf | (uintptr_t)next_obs) & 15) == 0) ? 1 : 0;      // 16 envs per wave
                const size_t lds6m = (size_t)k9_fast_lds_floats(img, 16, e->D, true, ts6) * sizeof(float);
                EnvParams<float> prm6 = e->params<float>();
                prm6.lg = 2;
#ifndef PC_DEV_MIN
                static bool attr6m[64] = {false};
                if (e->device >= 64 || !attr6m[e->device]) {
                    HIPCHK(hipFuncSetAttribute((const void*)rollout_kernel<6, 5, 2, 6, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                    if (e->device < 64) attr6m[e->device] = true;
                }
#endif
                if (e->track_bal32 && e->opt.deinterleave) {
#ifndef PC_DEV_MIN
                    static bool attr7m[64] = {false};
                    if (e->device >= 64 || !attr7m[e->device]) {
                        HIPCHK(hipFuncSetAttribute((const void*)rollout_kernel<6, 5, 2, 7, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                        if (e->device < 64) attr7m[e->device] = true;
                    }
#endif
                    PC_FULL(hipLaunchKernelGGL((rollout_kernel<6, 5, 2, 7, true, 2>), dim3((int)((e->N + 127) / 128)), dim3(512), lds6m, st, prm6, image, A, (int)T, reward_scale, seed,
                                               offset, offset_dev, obs_buf, act_buf, rew_buf, val_buf, term_buf, trunc_buf, logprob_buf, next_obs, next_term, next_trunc, 0, 128,
                                               vec6, last_value, reward_sum));
                } else
                PC_FULL(hipLaunchKernelGGL((rollout_kernel<6, 5, 2, 6, true, 2>), dim3((int)((e->N + 127) / 128)), dim3(512), lds6m, st, prm6, image, A, (int)T, reward_scale, seed,
                                           offset, offset_dev, obs_buf, act_buf, rew_buf, val_buf, term_buf, trunc_buf, logprob_buf, next_obs, next_term, next_trunc, 0, 128,
                                           vec6, last_value, reward_sum));
                HIPCHK(hipGetLastError());
                e->last_kernel = PC_KERNEL_K9M_LITERAL;
                return PC_OK;
            }
            if (lds6 <= 160 * 1024) {
                const int vec6 = ((e->N * e->D) % 4 == 0 && (((uintptr_t)obs_buf | (uintptr_t)next_obs) & 15) == 0) ? 1 : 0;
                EnvParams<float> prm6 = e->params<float>();
                prm6.lg = 1;
#ifndef PC_DEV_MIN
                static bool attr6[64] = {false};
                if (e->device >= 64 || !attr6[e->device]) {
                    HIPCHK(hipFuncSetAttribute((const void*)rollout_kernel<6, 9, 2, 6, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                    if (e->device < 64) attr6[e->device] = true;
                }
#endif
                if (e->track_bal64 && e->opt.deinterleave) {
#ifndef PC_DEV_MIN
                    static bool attr7[64] = {false};
                    if (e->device >= 64 || !attr7[e->device]) {
                        HIPCHK(hipFuncSetAttribute((const void*)rollout_kernel<6, 9, 2, 7, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                        if (e->device < 64) attr7[e->device] = true;
                    }
#endif
                    PC_FULL(hipLaunchKernelGGL((rollout_kernel<6, 9, 2, 7, true>), dim3(blocks), dim3(512), lds6, st, prm6, image, A, (int)T, reward_scale, seed, offset,
                                               offset_dev, obs_buf, act_buf, rew_buf, val_buf, term_buf, trunc_buf, logprob_buf, next_obs, next_term, next_trunc, 0, epw,
                                               vec6, last_value, reward_sum));
                } else
                PC_FULL(hipLaunchKernelGGL((rollout_kernel<6, 9, 2, 6, true>), dim3(blocks), dim3(512), lds6, st, prm6, image, A, (int)T, reward_scale, seed, offset,
                                           offset_dev, obs_buf, act_buf, rew_buf, val_buf, term_buf, trunc_buf, logprob_buf, next_obs, next_term, next_trunc, 0, epw,
                                           vec6, last_value, reward_sum));
                HIPCHK(hipGetLastError());
                e->last_kernel = PC_KERNEL_K9_LITERAL;
                return PC_OK;
            }
        }
        const bool shape = (rays16 || rays12 || rays33) && prec == 2 && e->D >= 17 && e->D <= 40 && max_G <= TAB_MAX_GATES &&
                           max_nV <= FT_VTX_MAX && e->opt.fast && e->opt.rden != 0 && (!e->track_id || e->track_block >= epw) &&
                           lds_sel <= 160 * 1024;
        // ... and up to PC_SPLIT_MAX_ENVS envs at 12 / 16 rays the SMALL form (K9s: 16 envs per workgroup up to 4096 envs, wave-owned envs -- env_step_wave's
        // literal form --, else 32 with the sweep parts on four waves), chosen as for F32 handles (PC_OPT_ROLLOUT_FORM / _EPW)
        const RolloutOpts& o = e->opt;
        const bool small = o.form == 1 || (o.form < 0 && e->N <= PC_SPLIT_MAX_ENVS);
        const bool epw16 = o.epw_override == 16 || (o.epw_override != 32 && e->N <= 4096);
        const int rden_small = 361 * max_nV;
        size_t lds_small = (size_t)(img + 8 * 32 * 17 + 32 * 40 + 32 + 128 + ft_floats(true, true)) * sizeof(float);
        const int rden_small_lds = (o.rden != 0 && lds_small + (size_t)rden_small * sizeof(float) <= 160 * 1024) ? rden_small : 0;
        lds_small += (size_t)rden_small_lds * sizeof(float);
        const bool shape_small = (rays16 || rays12) && prec == 2 && small && max_G <= TAB_MAX_GATES && max_nV <= FT_VTX_MAX && o.fast &&
                                 lds_small <= 160 * 1024;
        if (tabs && shape_small && !e->f64_offgrid && !interleaved) {
            const int vec_ok = ((e->N * e->D) % 4 == 0 && (((uintptr_t)obs_buf | (uintptr_t)next_obs) & 15) == 0) ? 1 : 0;
            EnvParams<float> prm = e->params<float>();
            prm.lg = 2;
#define PC_ROLLS_LIT(KSV, RPLV, EPWV)                                                                                    \
    do {                                                                                                                 \
        const int blocks_small = (int)((e->N + EPWV - 1) / EPWV);                                                        \
        static bool attr_set[64] = {false};                                                                              \
        if (e->device >= 64 || !attr_set[e->device]) {                                                                    \
            HIPCHK(hipFuncSetAttribute((const void*)rollout_small_kernel<KSV, RPLV, 2, 1, EPWV, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
            if (e->device < 64) attr_set[e->device] = true;                                                                \
        }                                                                                                                \
        hipLaunchKernelGGL((rollout_small_kernel<KSV, RPLV, 2, 1, EPWV, true>), dim3(blocks_small), dim3(512), lds_small, st, prm, image, A, (int)T, \
                           reward_scale, seed, offset, offset_dev, obs_buf, act_buf, rew_buf, val_buf, term_buf, trunc_buf, logprob_buf, \
                           next_obs, next_term, next_trunc, rden_small_lds, vec_ok, last_value, reward_sum);             \
    } while (0)
            if (rays16) {
                if (epw16) PC_DEV(6, PC_ROLLS_LIT(6, 5, 16));      // wave-owned envs (env_step_wave)
                else PC_FULL(PC_ROLLS_LIT(6, 5, 32));              // four sweep parts = waves (env_step_fast<..., 2, 4>)
            } else {                                               // 12 rays: three ray slots per lane
                if (epw16) PC_FULL(PC_ROLLS_LIT(5, 3, 16));
                else PC_FULL(PC_ROLLS_LIT(5, 3, 32));
            }
#undef PC_ROLLS_LIT
            HIPCHK(hipGetLastError());
            e->last_kernel = PC_KERNEL_K9S_LITERAL;
            return PC_OK;
        }
        // ... between PC_SPLIT_MAX_ENVS and PC_MEDIUM_MAX_ENVS envs at 16 rays the 16-envs-per-wave form (as for F32 handles)
        {
            const bool want = o.form == 4 || (o.form < 0 && o.epw_override == 0 && e->N > PC_SPLIT_MAX_ENVS && e->N <= PC_MEDIUM_MAX_ENVS);
            const int rden_m = 361 * max_nV;
            const size_t lds_m = (size_t)k9_fast_lds_floats(img, 16, e->D, true, rden_m) * sizeof(float);
            if (want && tabs && shape && rays16 && !e->f64_offgrid && !interleaved && (all_nv28 || all_loops) && (!e->track_id || e->track_block >= 128) &&
                lds_m <= 160 * 1024) {
                const int vec_ok = ((e->N * e->D) % 4 == 0 && (((uintptr_t)obs_buf | (uintptr_t)next_obs) & 15) == 0) ? 1 : 0;
                EnvParams<float> prm = e->params<float>();
                prm.lg = 2;
                const int blocks_m = (int)((e->N + 127) / 128);
#define PC_ROLL_MEDL(MD)                                                                                                 \
    do {                                                                                                                 \
        static bool attr_set[64] = {false};                                                                              \
        if (e->device >= 64 || !attr_set[e->device]) {                                                                    \
            HIPCHK(hipFuncSetAttribute((const void*)rollout_kernel<6, 5, 2, MD, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
            if (e->device < 64) attr_set[e->device] = true;                                                                \
        }                                                                                                                \
        hipLaunchKernelGGL((rollout_kernel<6, 5, 2, MD, true, 2>), dim3(blocks_m), dim3(512), lds_m, st, prm, image, A, (int)T, reward_scale, seed, \
                           offset, offset_dev, obs_buf, act_buf, rew_buf, val_buf, term_buf, trunc_buf, logprob_buf, next_obs,  \
                           next_term, next_trunc, rden_m, 128, vec_ok, last_value, reward_sum);                          \
    } while (0)
                if (all_nv28) PC_DEV(8, PC_ROLL_MEDL(3)); else PC_FULL(PC_ROLL_MEDL(5));
#undef PC_ROLL_MEDL
                HIPCHK(hipGetLastError());
                e->last_kernel = PC_KERNEL_K9M_LITERAL;
                return PC_OK;
            }
        }
        if (tabs && shape && !e->f64_offgrid && !interleaved) {
            const int vec_ok = ((e->N * e->D) % 4 == 0 && (((uintptr_t)obs_buf | (uintptr_t)next_obs) & 15) == 0) ? 1 : 0;
            EnvParams<float> prm = e->params<float>();
            prm.lg = 1;
#define PC_ROLL_LIT(KSV, RPLV, MD)                                                                                       \
    do {                                                                                                                 \
        static bool attr_set[64] = {false};                                                                              \
        if (e->device >= 64 || !attr_set[e->device]) {                                                                    \
            HIPCHK(hipFuncSetAttribute((const void*)rollout_kernel<KSV, RPLV, 2, MD, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
            if (e->device < 64) attr_set[e->device] = true;                                                                \
        }                                                                                                                \
        hipLaunchKernelGGL((rollout_kernel<KSV, RPLV, 2, MD, true>), dim3(blocks), dim3(512), lds_sel, st, prm, image, A, (int)T, reward_scale, seed, \
                           offset, offset_dev, obs_buf, act_buf, rew_buf, val_buf, term_buf, trunc_buf, logprob_buf, next_obs,  \
                           next_term, next_trunc, rden_lds, epw, vec_ok, last_value, reward_sum);                        \
    } while (0)
            if (rays16 && all_nv28) PC_DEV(3, PC_ROLL_LIT(6, 9, 3));            // big_track.json's layout
            else if (rays16 && all_loops) PC_FULL(PC_ROLL_LIT(6, 9, 5));        // ... mixed with track.json's
            else if (rays16) { PC_FULL(if (rden_lds) PC_ROLL_LIT(6, 9, 2); else PC_ROLL_LIT(6, 9, 1)); }     // any other track: the generic sweeps
            else if (rays33) PC_DEV(5, PC_ROLL_LIT(10, 17, 1));                 // 32 -> 33 rays: no room for the 1/den table
            else { PC_FULL(if (rden_lds) PC_ROLL_LIT(5, 6, 2); else PC_ROLL_LIT(5, 6, 1)); }                // 12 rays
#undef PC_ROLL_LIT
            HIPCHK(hipGetLastError());
            e->last_kernel = PC_KERNEL_K9_LITERAL;
            return PC_OK;
        }
    }
    const size_t lds = (size_t)(polx_image_dwords(prec, pol_ng(KS)) + 256 * (4 * KS + 1)) * sizeof(float);
    if (lds > 160 * 1024) return PC_ERR_UNSUPPORTED;
    // K9d: the generic kernel.  With PC_OPT_ROLLOUT_FAST = 0 every (ray, wall) pair is tested in float64 (the FILTER form: no float32
    // anywhere); otherwise tracks inside the selector's limits step as the per-step kernel does (sweep over the chain in global
    // memory, literal cast: rollout_f64_kernel<..., SEL>) -- what is left for this kernel by default are tracks too long for the
    // LDS tables and rotations off the table; bf16 x 3 keeps the filter.
    bool all_sel = true;
    for (const TrackHdr& h : e->hdr_host) all_sel = all_sel && h.sel_ok;
    const int sel_on = (e->opt.fast && all_sel) ? 1 : 0;
    const EnvParams<double> prm = [&] { EnvParams<double> q = e->params<double>(); q.lg = 1; return q; }();
#define PC_ROLLD(KSV, RPLV, PRC, SELV)                                                                                   \
    do {                                                                                                                 \
        static bool attr_set[64] = {false};                                                                              \
        if (e->device >= 64 || !attr_set[e->device]) {                                                                    \
            HIPCHK(hipFuncSetAttribute((const void*)rollout_f64_kernel<KSV, RPLV, PRC, SELV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
            if (e->device < 64) attr_set[e->device] = true;                                                                \
        }                                                                                                                \
        hipLaunchKernelGGL((rollout_f64_kernel<KSV, RPLV, PRC, SELV>), dim3(blocks), dim3(512), lds, st, prm, image, A, (int)T, reward_scale, seed, \
                           offset, offset_dev, obs_buf, act_buf, rew_buf, val_buf, term_buf, trunc_buf, logprob_buf, next_obs,  \
                           next_term, next_trunc, epw, last_value, reward_sum);                                          \
    } while (0)
    // (the selector variant for the fp16 x 2 forms only: bf16 x 3 -- 12 rays -- keeps the filter)
    const bool sel_k = sel_on && prec == 2;
    if (KS == 5 && rpl == 6) { PC_FULL(if (sel_k) PC_ROLLD(5, 6, 2, true); else if (prec == 2) PC_ROLLD(5, 6, 2, false); else PC_ROLLD(5, 6, 1, false)); }              // 12 rays, D = 18
    else if (KS == 6 && rpl == 9) {                                                                                    // 16 -> 17 rays, D = 23 (bf16 x 3 there spills 3 registers: not built)
        if (prec != 2) return PC_ERR_UNSUPPORTED;
        if (sel_k) PC_FULL(PC_ROLLD(6, 9, 2, true)); else PC_DEV(4, PC_ROLLD(6, 9, 2, false));
    }
    else return PC_ERR_UNSUPPORTED;     // (32 -> 33 rays: 17 float64 ray slots per lane beside the policy state spill 96 registers -- not built;
                                        //  the per-step kernels run that shape)
#undef PC_ROLLD
    HIPCHK(hipGetLastError());
    e->last_kernel = sel_k ? PC_KERNEL_K9D_SELECTOR : PC_KERNEL_K9D_FILTER;
    return PC_OK;
}

static int rollout_impl(pc_env* e, int prec_request, const float* image, int A, int64_t T, double reward_scale, uint64_t seed, uint64_t offset,
                        const uint64_t* offset_dev, float* obs_buf, float* act_buf, float* rew_buf, float* val_buf, float* term_buf,
                        float* trunc_buf, float* logprob_buf, float* next_obs, float* next_term, float* next_trunc, float* last_value,
                        float* reward_sum, void* stream) {
    if (!e || !image || !obs_buf || !act_buf || !rew_buf || !val_buf || !term_buf || !trunc_buf || !logprob_buf || !next_obs ||
        !next_term || !next_trunc || T < 1 || T > (1 << 24))
        return PC_ERR_INVALID_ARG;
    // mixed tracks: a wave (big form) / a workgroup (small form) steps 32 consecutive envs, which must share one track
    if (e->dtype == PC_DTYPE_F64)
        return rollout_f64_impl(e, prec_request, image, A, T, reward_scale, seed, offset, offset_dev, obs_buf, act_buf, rew_buf, val_buf, term_buf,
                                trunc_buf, logprob_buf, next_obs, next_term, next_trunc, last_value, reward_sum, stream);
    if (e->dtype != PC_DTYPE_F32 || A < 1 || A > 15) return PC_ERR_UNSUPPORTED;
    // mixed tracks interleaved inside a wave (no aligned block of 32 envs on one track; car_env.py:621-628 makes that legal): the BIG
    // form's generic mode, whose env step runs once per distinct track id of a wave (K1's waterfall) -- the fast modes stage ONE track's
    // tables per workgroup, and the small form's sweep parts meet across workgroup barriers that a per-wave loop cannot contain
    const bool interleaved = e->track_id && !e->track_blocks32;
    const int KS = policy_ks(e->D);
    DeviceGuard guard(e->device);
    if (!guard.ok) return PC_ERR_NO_DEVICE;
    const int prec = policy_prec(prec_request, e->D, A);
    const RolloutOpts& o = e->opt;
    const int img = prec ? polx_image_dwords(prec, pol_ng(KS)) : pol_image_padded(KS);
    // large batches: 256 envs per workgroup, every wave independent; small batches: 32 envs per workgroup, hidden tiles and
    // wall-sweep parts split over the waves
    if (interleaved && o.form == 1) return PC_ERR_UNSUPPORTED;
    // ... and the FAST form of that layout for the reference's own pair of tracks (two tracks, each two equal loops of 13 or 9 chain
    // vertices, 16 -> 17 rays, fp16 x 2): both tracks' tables in LDS, the table-driven env step once per track of a wave (rollout_kernel<6, 9, 2, 6>)
    if (interleaved && o.fast && o.nv28 != 0 && e->n_tracks == 2 && policy_ks(e->D) == 6 && e->n_nominal == 16 && A == 9 && policy_prec(prec_request, e->D, A) == 2 &&
        e->D >= 17 && e->D <= 40) {
        bool loops2 = true;
        int mg = 0;
        for (const TrackHdr& h : e->hdr_host) {
            loops2 = loops2 && h.vtxp_off >= 0 && (h.brk2 == 13 || h.brk2 == 9) && h.n_chain == 2 * h.brk2 && h.nV == 4 * ((h.brk2 + 1) / 2) && h.nV <= FT_VTX_MAX;
            mg = std::max(mg, h.G);
        }
        const int img6 = polx_image_dwords(2, pol_ng(6));
        const int ts6 = (ft_floats(false, true) + 3) & ~3;
        const size_t lds6 = (size_t)k9_fast_lds_floats(img6, 32, e->D, true, ts6) * sizeof(float);
        if (loops2 && mg <= TAB_MAX_GATES && lds6 <= 160 * 1024) {
            DeviceGuard guard6(e->device);
            if (!guard6.ok) return PC_ERR_NO_DEVICE;
            if (o.form == 4 || (o.form < 0 && o.epw_override == 0 && e->N > PC_SPLIT_MAX_ENVS && e->N <= PC_MEDIUM_MAX_ENVS)) {
                // 16 envs per wave (4 lanes per env), as for single-track and block-mixed batches of this size: two waves on every SIMD
                const int vec6 = ((e->N * e->D) % 4 == 0 && (((uintptr_t)obs_buf | (uintptr_t)next_obs) & 15) == 0) ? 1 : 0;
                const size_t lds6m = (size_t)k9_fast_lds_floats(img6, 16, e->D, true, ts6) * sizeof(float);
                EnvParams<float> prm6 = e->params<float>();
                prm6.lg = 2;
#ifndef PC_DEV_MIN
                static bool attr6m[64] = {false};
                if (e->device >= 64 || !attr6m[e->device]) {
                    HIPCHK(hipFuncSetAttribute((const void*)rollout_kernel<6, 5, 2, 6, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                    if (e->device < 64) attr6m[e->device] = true;
                }
#endif
                if (e->track_bal32 && o.deinterleave) {      // every block of 32 envs split evenly: the block's two waves de-interleave it (mode 7)
#ifndef PC_DEV_MIN
                    static bool attr7m[64] = {false};
                    if (e->device >= 64 || !attr7m[e->device]) {
                        HIPCHK(hipFuncSetAttribute((const void*)rollout_kernel<6, 5, 2, 7, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                        if (e->device < 64) attr7m[e->device] = true;
                    }
#endif
                    PC_FULL(hipLaunchKernelGGL((rollout_kernel<6, 5, 2, 7, false, 2>), dim3((int)((e->N + 127) / 128)), dim3(512), lds6m, (hipStream_t)stream, prm6, image, A, (int)T,
                                               reward_scale, seed, offset, offset_dev, obs_buf, act_buf, rew_buf, val_buf, term_buf, trunc_buf, logprob_buf, next_obs, next_term,
                                               next_trunc, 0, 128, vec6, last_value, reward_sum));
                } else
                PC_FULL(hipLaunchKernelGGL((rollout_kernel<6, 5, 2, 6, false, 2>), dim3((int)((e->N + 127) / 128)), dim3(512), lds6m, (hipStream_t)stream, prm6, image, A, (int)T,
                                           reward_scale, seed, offset, offset_dev, obs_buf, act_buf, rew_buf, val_buf, term_buf, trunc_buf, logprob_buf, next_obs, next_term,
                                           next_trunc, 0, 128, vec6, last_value, reward_sum));
                HIPCHK(hipGetLastError());
                e->last_kernel = PC_KERNEL_K9M;
                return PC_OK;
            }
            const int epw6 = o.epw_override >= 128 ? o.epw_override : (e->N <= g_rollout_epw128_max ? 128 : 256);
            const int blocks6 = (int)((e->N + epw6 - 1) / epw6);
            const int vec6 = ((e->N * e->D) % 4 == 0 && (((uintptr_t)obs_buf | (uintptr_t)next_obs) & 15) == 0) ? 1 : 0;
            EnvParams<float> prm6 = e->params<float>();
            prm6.lg = 1;
#ifndef PC_DEV_MIN
            static bool attr6[64] = {false};
            if (e->device >= 64 || !attr6[e->device]) {
                HIPCHK(hipFuncSetAttribute((const void*)rollout_kernel<6, 9, 2, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                if (e->device < 64) attr6[e->device] = true;
            }
#endif
            if (e->track_bal64 && o.deinterleave) {
#ifndef PC_DEV_MIN
                static bool attr7[64] = {false};
                if (e->device >= 64 || !attr7[e->device]) {
                    HIPCHK(hipFuncSetAttribute((const void*)rollout_kernel<6, 9, 2, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                    if (e->device < 64) attr7[e->device] = true;
                }
#endif
                PC_FULL(hipLaunchKernelGGL((rollout_kernel<6, 9, 2, 7>), dim3(blocks6), dim3(512), lds6, (hipStream_t)stream, prm6, image, A, (int)T, reward_scale, seed, offset,
                                           offset_dev, obs_buf, act_buf, rew_buf, val_buf, term_buf, trunc_buf, logprob_buf, next_obs, next_term, next_trunc, 0, epw6,
                                           vec6, last_value, reward_sum));
            } else
            PC_FULL(hipLaunchKernelGGL((rollout_kernel<6, 9, 2, 6>), dim3(blocks6), dim3(512), lds6, (hipStream_t)stream, prm6, image, A, (int)T, reward_scale, seed, offset,
                                       offset_dev, obs_buf, act_buf, rew_buf, val_buf, term_buf, trunc_buf, logprob_buf, next_obs, next_term, next_trunc, 0, epw6,
                                       vec6, last_value, reward_sum));
            HIPCHK(hipGetLastError());
            e->last_kernel = PC_KERNEL_K9;
            return PC_OK;
        }
    }
    const bool small = !interleaved && (o.form == 1 || (o.form < 0 && e->N <= PC_SPLIT_MAX_ENVS));
    const int epw = o.epw_override >= 128 ? o.epw_override
                                                  : ((!small && e->N <= g_rollout_epw128_max) ? 128 : 256);   // big form: envs per workgroup
    int max_G = 0, max_nV = 0;
    for (const TrackHdr& h : e->hdr_host) { max_G = std::max(max_G, h.G); max_nV = std::max(max_nV, h.nV); }
    // fast mode: Discrete(9), every gather table in LDS behind LDS pointers, dense observation rows (big form: each wave's output
    // tile aliases its own 32 observation rows -- dead between the policy pass's operand load and the env step's store of the
    // next observation: needs D >= 17).  A workgroup stages ONE track's tables: single-track batches, or mixed ones in which
    // every workgroup's block of envs lies on one track.
    // The fast kernels carry their observation width as a compile-time constant derived from the ray slots per lane (12 / 17 /
    // 33 rays) and cast exactly four collision rays at the reward gate: num_rays 12 / 16 / 32 (what BASELINE's configs name).  Any
    // other count that maps onto the same slots (17 or 18 nominal rays -> 18 actual: the slots of 17; 31 -> 33 with five
    // collision rays) takes the generic mode, which reads all of that from the handle.
    const bool fast_rays = e->n_nominal == 12 || e->n_nominal == 16 || e->n_nominal == 32;
    const bool fast_shape = A == 9 && fast_rays && e->D >= 17 && e->D <= 40 && max_G <= TAB_MAX_GATES && o.fast;
    // (the float64 refinement gathers the chain from LDS: at most FT_VTX_MAX vertices; a shape whose fast-mode tables do not fit
    // beside the weight image -- the fp32 image at 33 rays -- takes the generic mode)
    const size_t lds_fast_big = (size_t)k9_fast_lds_floats(img, 32, e->D, KS != 10, 0) * sizeof(float);   // (33 rays: one turn of the float64 lattice)
    const bool fast = !small && fast_shape && max_nV <= FT_VTX_MAX && (!e->track_id || e->track_block >= epw) && lds_fast_big <= 160 * 1024;
    const size_t lds_big = fast ? lds_fast_big : (size_t)(img + 256 * (4 * KS + 1) + 256 + TAB_FLOATS) * sizeof(float);
    const size_t lds_fast_small = (size_t)(img + 8 * 32 * 17 + 32 * 40 + 32 + 128 + ft_floats(true, true)) * sizeof(float);
    const bool fast_small = small && fast_shape && max_nV <= FT_VTX_MAX && lds_fast_small <= 160 * 1024;     // (a small-form workgroup is 16 or 32 envs)
    const size_t lds_small = fast_small ? lds_fast_small : (size_t)(img + 8 * 32 * 17 + 32 * (4 * KS + 1) + 32 + 128 + TAB_FLOATS) * sizeof(float);
    size_t lds = small ? lds_small : lds_big;
    if (lds > 160 * 1024) return PC_ERR_UNSUPPORTED;
    // the track's 1/den table rides along in LDS when it fits (big_track: 361 x 28 floats = 40 KB); else the sweep forms
    // den and its reciprocal itself -- same bits either way.  Mixed batches: in the fast modes only (room for the largest track).
    // (the big form at 33 rays has 4 KB left: no closed track's table fits, so that shape is built without the table mode)
    int rden_lds = 361 * max_nV;
    bool all_rden = true;
    for (const TrackHdr& h : e->hdr_host) all_rden = all_rden && h.rden_off >= 0;
    if (!all_rden || o.rden == 0 || (e->track_id && !(small ? fast_small : fast)) || lds + (size_t)rden_lds * sizeof(float) > 160 * 1024 ||
        (!small && KS == 10))
        rden_lds = 0;
    lds += (size_t)rden_lds * sizeof(float);
    const int rpl = small ? (e->R + 3) / 4 : (e->R + 1) / 2;  // 4 (x 4 sweep parts) or 2 lanes per env
    // small form: 16 envs per workgroup up to 4096 envs (<= 256 workgroups: one per CU), else 32
    const int epw_small = (o.epw_override == 16 || o.epw_override == 32) ? o.epw_override
                          : ((fast_small && prec != 0 && e->R <= 17 && e->N <= 4096) ? 16 : 32);
    if (small && epw_small == 16 && !(fast_small && prec != 0 && e->R <= 17)) return PC_ERR_UNSUPPORTED;
    const int blocks = (int)(small ? (e->N + epw_small - 1) / epw_small : (e->N + epw - 1) / epw);
    // 16-byte stores of the waves' 32-row blocks: the rows' offsets inside the buffers AND the buffers themselves are aligned
    const int vec_ok = ((e->N * e->D) % 4 == 0 && (((uintptr_t)obs_buf | (uintptr_t)next_obs) & 15) == 0) ? 1 : 0;
    const int mode = (fast || fast_small) ? (rden_lds ? 2 : 1) : 0;
    bool all_nv28 = o.nv28 != 0;
    bool all_loops = o.nv28 != 0;       // ... or every track is two equal chains of 13 or of 9 vertices (track.json: 8 walls per loop): the mixed form of those kernels
    for (const TrackHdr& h : e->hdr_host) {
        all_nv28 = all_nv28 && h.nV == 28 && h.n_chain == 26 && h.brk2 == 13 && h.vtxp_off >= 0;   // big_track's layout: two loops of 12 walls
        all_loops = all_loops && h.vtxp_off >= 0 && (h.brk2 == 13 || h.brk2 == 9) && h.n_chain == 2 * h.brk2 && h.nV == 4 * ((h.brk2 + 1) / 2);
    }
    hipStream_t st = (hipStream_t)stream;
    EnvParams<float> prm = e->params<float>();
    prm.lg = small ? 2 : 1;
    // ---- the 16-envs-per-wave form (rollout_kernel<6, 5, 2, MD, LIT, 2>): 17 rays, fp16 x 2, the chain-packed sweeps' track layouts, the
    // 1/den table in LDS; automatic between PC_SPLIT_MAX_ENVS and PC_MEDIUM_MAX_ENVS envs, PC_OPT_ROLLOUT_FORM = 4 at any size
    {
        const bool want = o.form == 4 || (o.form < 0 && o.epw_override == 0 && e->N > PC_SPLIT_MAX_ENVS && e->N <= PC_MEDIUM_MAX_ENVS);
        const int rden_m = 361 * max_nV;
        const size_t lds_m = (size_t)k9_fast_lds_floats(img, 16, e->D, true, rden_m) * sizeof(float);
        if (want && KS == 6 && prec == 2 && e->n_nominal == 16 && fast_shape && o.rden != 0 && max_nV <= FT_VTX_MAX && (all_nv28 || all_loops) &&
            (!e->track_id || e->track_block >= 128) && lds_m <= 160 * 1024) {
            const int blocks_m = (int)((e->N + 127) / 128);
            prm.lg = 2;
#define PC_ROLL_MED(MD)                                                                                                  \
    do {                                                                                                                 \
        static bool attr_set[64] = {false};                                                                              \
        if (e->device >= 64 || !attr_set[e->device]) {                                                                    \
            HIPCHK(hipFuncSetAttribute((const void*)rollout_kernel<6, 5, 2, MD, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
            if (e->device < 64) attr_set[e->device] = true;                                                                \
        }                                                                                                                \
        hipLaunchKernelGGL((rollout_kernel<6, 5, 2, MD, false, 2>), dim3(blocks_m), dim3(512), lds_m, st, prm, image, A, (int)T, reward_scale, seed, \
                           offset, offset_dev, obs_buf, act_buf, rew_buf, val_buf, term_buf, trunc_buf, logprob_buf, next_obs,  \
                           next_term, next_trunc, rden_m, 128, vec_ok, last_value, reward_sum);                          \
    } while (0)
            if (all_nv28) PC_DEV(7, PC_ROLL_MED(3)); else PC_FULL(PC_ROLL_MED(5));
#undef PC_ROLL_MED
            HIPCHK(hipGetLastError());
            e->last_kernel = PC_KERNEL_K9M;
            return PC_OK;
        }
    }
#define PC_ROLL_M(KSV, RPLV, PRC, MD)                                                                                    \
    do {                                                                                                                 \
        static bool attr_set[64] = {false};                                                                              \
        if (e->device >= 64 || !attr_set[e->device]) {                                                                    \
            HIPCHK(hipFuncSetAttribute((const void*)rollout_kernel<KSV, RPLV, PRC, MD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
            if (e->device < 64) attr_set[e->device] = true;                                                                                \
        }                                                                                                                \
        hipLaunchKernelGGL((rollout_kernel<KSV, RPLV, PRC, MD>), dim3(blocks), dim3(512), lds, st, prm, image, A, (int)T, reward_scale, seed, \
                           offset, offset_dev, obs_buf, act_buf, rew_buf, val_buf, term_buf, trunc_buf, logprob_buf, next_obs,  \
                           next_term, next_trunc, rden_lds, epw, vec_ok, last_value, reward_sum);                        \
    } while (0)
#define PC_ROLL(KSV, RPLV, PRC)                                                                                          \
    do {                                                                                                                 \
        if constexpr (PRC == 2 && KSV == 6) {   /* (17 rays, default arithmetic only: the chain-of-28 kernels) */        \
            if (mode == 2 && all_nv28) { PC_DEV(0, PC_ROLL_M(KSV, RPLV, PRC, 3)); break; }                               \
            if (mode == 2 && all_loops) { PC_FULL(PC_ROLL_M(KSV, RPLV, PRC, 5)); break; }                                \
            if (mode == 1 && all_nv28) { PC_FULL(PC_ROLL_M(KSV, RPLV, PRC, 4)); break; }                                 \
        }                                                                                                                \
        if (mode == 2) PC_FULL(PC_ROLL_M(KSV, RPLV, PRC, 2));                                                            \
        else if (mode == 1) PC_FULL(PC_ROLL_M(KSV, RPLV, PRC, 1));                                                       \
        else PC_FULL(PC_ROLL_M(KSV, RPLV, PRC, 0));                                                                      \
    } while (0)
#define PC_ROLLS_M(KSV, RPLV, PRC, MD, EPWV)                                                                             \
    do {                                                                                                                 \
        static bool attr_set[64] = {false};                                                                              \
        if (e->device >= 64 || !attr_set[e->device]) {                                                                    \
            HIPCHK(hipFuncSetAttribute((const void*)rollout_small_kernel<KSV, RPLV, PRC, MD, EPWV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
            if (e->device < 64) attr_set[e->device] = true;                                                                                \
        }                                                                                                                \
        hipLaunchKernelGGL((rollout_small_kernel<KSV, RPLV, PRC, MD, EPWV>), dim3(blocks), dim3(512), lds, st, prm, image, A, (int)T, reward_scale, \
                           seed, offset, offset_dev, obs_buf, act_buf, rew_buf, val_buf, term_buf, trunc_buf, logprob_buf, next_obs, \
                           next_term, next_trunc, rden_lds, vec_ok, last_value, reward_sum);                                     \
    } while (0)
#define PC_ROLLS(KSV, RPLV, PRC)                                                                                         \
    do {                                                                                                                 \
        if constexpr (PRC != 0 && RPLV <= 5) {                                                                           \
            if (mode && epw_small == 16) {                                                                               \
                if constexpr (PRC == 2 && KSV == 6) {   /* (configs[1]'s kernel: with the 1/den table in LDS the env step is compiled for it) */ \
                    if (rden_lds) PC_DEV(1, PC_ROLLS_M(KSV, RPLV, PRC, 2, 16)); else PC_FULL(PC_ROLLS_M(KSV, RPLV, PRC, 1, 16));    \
                } else PC_FULL(PC_ROLLS_M(KSV, RPLV, PRC, 1, 16));                                                               \
                break;                                                                                                           \
            }                                                                                                                    \
        }                                                                                                                \
        if (mode) PC_FULL(PC_ROLLS_M(KSV, RPLV, PRC, 1, 32));    /* (the small form takes the 1/den table as a run-time branch) */   \
        else PC_FULL(PC_ROLLS_M(KSV, RPLV, PRC, 0, 32));                                                                 \
    } while (0)
    if (small) {
        if (KS == 5 && rpl == 3) { PC_FULL(if (prec == 2) PC_ROLLS(5, 3, 2); else if (prec) PC_ROLLS(5, 3, 1); else PC_ROLLS(5, 3, 0)); }        // 12 rays
        else if (KS == 6 && rpl == 5) { if (prec == 2) PC_ROLLS(6, 5, 2); else PC_FULL(if (prec) PC_ROLLS(6, 5, 1); else PC_ROLLS(6, 5, 0)); }   // 16 -> 17 rays
        else if (KS == 10 && rpl == 9) {                                                                                                      // 32 -> 33 rays
            PC_FULL(if (prec == 2) PC_ROLLS(10, 9, 2);
                    else if (prec) { if (mode) PC_ROLLS_M(10, 9, 1, 1, 32); else return PC_ERR_UNSUPPORTED; }   // (bf16 x 3 in the generic mode spilled: the caller's per-step kernels take that shape)
                    else PC_ROLLS(10, 9, 0));
        }
        else return PC_ERR_UNSUPPORTED;
    } else if (KS == 5 && rpl == 6) { PC_FULL(if (prec == 2) PC_ROLL(5, 6, 2); else if (prec) PC_ROLL(5, 6, 1); else PC_ROLL(5, 6, 0)); }       // 12 rays, D = 18
    else if (KS == 6 && rpl == 9) { if (prec == 2) PC_ROLL(6, 9, 2); else PC_FULL(if (prec) PC_ROLL(6, 9, 1); else PC_ROLL(6, 9, 0)); }          // 16 -> 17 rays, D = 23
    else if (KS == 10 && rpl == 17 && prec) {                                                                                             // 32 -> 33 rays, D = 39
        // (the chain-of-28 variant spills: not built.  The GENERIC mode at 33 rays -- a mixed-track batch whose workgroups straddle
        // tracks, fast mode switched off -- spilled 100+ registers beside the split operands' policy state: not built either; that
        // shape is PC_ERR_UNSUPPORTED here and runs through the per-step kernels, bit-identical by construction)
        if (!mode) return PC_ERR_UNSUPPORTED;
        if (prec == 2) PC_DEV(2, PC_ROLL_M(10, 17, 2, 1));
        else PC_FULL(PC_ROLL_M(10, 17, 1, 1));
    }
    else return PC_ERR_UNSUPPORTED;
#undef PC_ROLL_M
#undef PC_ROLLS_M
#undef PC_ROLLS
#undef PC_ROLL
    HIPCHK(hipGetLastError());
    e->last_kernel = small ? PC_KERNEL_K9S : PC_KERNEL_K9;
    return PC_OK;
}

int pc_rollout(pc_env* e, const pc_policy* p, const float* image, int64_t T, double reward_scale, uint64_t seed, uint64_t offset,
               const uint64_t* offset_dev, float* obs_buf, float* act_buf, float* rew_buf, float* val_buf, float* term_buf,
               float* trunc_buf, float* logprob_buf, float* next_obs, float* next_term, float* next_trunc, float* last_value,
               float* reward_sum, void* stream) {
    g_hip_err.clear();   // (pc_last_hip_error speaks of THIS call)
    if (!e || !p) return PC_ERR_INVALID_ARG;
    if (p->D != e->D || p->device != e->device) return PC_ERR_INVALID_ARG;     // the policy was built for another observation width / device
    return rollout_impl(e, p->precision, image, p->A, T, reward_scale, seed, offset, offset_dev, obs_buf, act_buf, rew_buf, val_buf, term_buf,
                        trunc_buf, logprob_buf, next_obs, next_term, next_trunc, last_value, reward_sum, stream);
}

int64_t pc_ppo_workspace_floats(int B, int D, int H, int A) {
    g_hip_err.clear();   // (pc_last_hip_error speaks of THIS call)
    if (H != 256 || A < 1 || A > 15 || D < 1 || D > 40 || B < 2 || B > 1024) return PC_ERR_UNSUPPORTED;
    const int64_t n_param = 2 * ((int64_t)H * D + H) + (int64_t)A * H + A + H + 1;
    const int64_t n_part = (B + FB_S - 1) / FB_S;
    return n_part * ((n_param + 3) & ~(int64_t)3) + n_part * 4 + (n_param + 255) / 256;
}

// the layout of pc_ppo_minibatch's workspace and the launches of one minibatch step
struct MbPlan {
    int n_param, n_part, n_blk, n_pad, HD, mid_end;
    float *partial, *metric_partial, *norm_partial;
    MbPlan(int B, int D, int H, int A, float* workspace) {
        n_param = 2 * (H * D + H) + A * H + A + H + 1;
        n_part = (B + FB_S - 1) / FB_S;
        n_blk = (n_param + 255) / 256;
        n_pad = (n_param + 3) & ~3;          // a partial's row stride: 16-byte aligned rows
        HD = H * D;
        mid_end = HD + H + A * H + A;        // natural offset of critic.0.weight (ppo_fwdbwd_body's o_cW1)
        partial = workspace;
        metric_partial = partial + (size_t)n_part * n_pad;
        norm_partial = metric_partial + n_part * 4;
    }
};

static void launch_fwdbwd(const MbPlan& pl, const int64_t* idx, const float* prep, int B, int D, int A, const float* obs, const float* act,
                          const float* old_logprob, const float* adv, const float* ret, const float* param, double clip_ratio, double vf_coef,
                          double ent_coef, const AdamDefer& df, hipStream_t st) {
#define PC_FB(DM, ACV, DCV, DF)                                                                                          \
    hipLaunchKernelGGL((ppo_fwdbwd_kernel<DM, ACV, DCV, DF>), dim3(pl.n_part), dim3(256), 0, st, idx, B, D, A, obs, act, old_logprob, adv, ret, param, \
                       (float)clip_ratio, (float)vf_coef, (float)ent_coef, pl.partial, pl.metric_partial, prep, df)
    // CarEnv's shapes (Discrete(9); 6 + 12 / 17 / 33 rays) have their action count and observation width compiled in -- and the
    // deferred clip + Adam prologue (mb_defer_shape)
    if (A == 9 && D == 23) { if (df.grad) PC_FB(24, 9, 23, true); else PC_FB(24, 9, 23, false); }
    else if (A == 9 && D == 18) { if (df.grad) PC_FB(24, 9, 18, true); else PC_FB(24, 9, 18, false); }
    else if (A == 9 && D == 39) { if (df.grad) PC_FB(40, 9, 39, true); else PC_FB(40, 9, 39, false); }
    else if (D <= 24) PC_FB(24, 0, 0, false);
    else PC_FB(40, 0, 0, false);
#undef PC_FB
}
static bool mb_defer_shape(int D, int A) { return A == 9 && (D == 18 || D == 23 || D == 39); }

static void launch_reduce(const MbPlan& pl, int B, double vf_coef, double ent_coef, float* grad, float* metrics, float* step_count, hipStream_t st) {
    hipLaunchKernelGGL(grad_reduce_kernel, dim3(pl.n_blk), dim3(256), 0, st, pl.partial, pl.n_part, pl.n_param, pl.HD, pl.mid_end, pl.n_pad, grad,
                       pl.norm_partial, pl.metric_partial, B, (float)vf_coef, (float)ent_coef, metrics, step_count);
}

static void launch_adam(const MbPlan& pl, const float* p_in, const float* m_in, const float* v_in, float* grad, float* p_out, float* m_out,
                        float* v_out, const float* step_count, const float* lr_dev, double max_norm, double beta1, double beta2, double eps,
                        hipStream_t st) {
    hipLaunchKernelGGL(adam_kernel, dim3(pl.n_blk), dim3(256), 0, st, p_in, m_in, v_in, grad, p_out, m_out, v_out, step_count, lr_dev,
                       pl.norm_partial, pl.n_blk, pl.n_param, (float)max_norm, (float)beta1, (float)beta2, (float)eps);
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
    const MbPlan pl(B, D, H, A, workspace);
    hipStream_t st = (hipStream_t)stream;
    launch_fwdbwd(pl, idx, prep, B, D, A, obs, act, old_logprob, adv, ret, param, clip_ratio, vf_coef, ent_coef, AdamDefer{}, st);
    launch_reduce(pl, B, vf_coef, ent_coef, grad, metrics, apply ? step_count : nullptr, st);
    if (apply == 1) launch_adam(pl, param, exp_avg, exp_avg_sq, grad, param, exp_avg, exp_avg_sq, step_count, lr_dev, max_norm, beta1, beta2, eps, st);
    HIPCHK(hipGetLastError());
    return PC_OK;
}

int pc_ppo_minibatch(int device, const int64_t* idx, int B, int D, int H, int A, const float* obs, const float* act,
                     const float* old_logprob, const float* adv, const float* ret, float* param, float* grad, float* exp_avg,
                     float* exp_avg_sq, float* step_count, const float* lr_dev, double clip_ratio, double vf_coef, double ent_coef,
                     double max_norm, double beta1, double beta2, double eps, float* metrics, float* workspace, int apply,
                     void* stream) {
    g_hip_err.clear();   // (pc_last_hip_error speaks of THIS call)
    return ppo_minibatch_impl(device, idx, nullptr, B, D, H, A, obs, act, old_logprob, adv, ret, param, grad, exp_avg, exp_avg_sq,
                              step_count, lr_dev, clip_ratio, vf_coef, ent_coef, max_norm, beta1, beta2, eps, metrics, workspace, apply,
                              stream);
}

int64_t pc_ppo_prepared_floats(int B, int D) {
    g_hip_err.clear();   // (pc_last_hip_error speaks of THIS call)
    if (D < 1 || D > 40 || B < 2 || B > 1024) return PC_ERR_UNSUPPORTED;
    return (int64_t)B * (D + 4) + 4;
}

int pc_ppo_prepare(int device, const int64_t* idx, int64_t idx_ld, int n_mb, int B, int D, const float* obs, const float* act,
                   const float* old_logprob, const float* adv, const float* ret, float* prepared, void* stream) {
    g_hip_err.clear();   // (pc_last_hip_error speaks of THIS call)
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
    g_hip_err.clear();   // (pc_last_hip_error speaks of THIS call)
    if (!prepared_mb) return PC_ERR_INVALID_ARG;
    return ppo_minibatch_impl(device, nullptr, prepared_mb, B, D, H, A, nullptr, nullptr, nullptr, nullptr, nullptr, param, grad, exp_avg,
                              exp_avg_sq, step_count, lr_dev, clip_ratio, vf_coef, ent_coef, max_norm, beta1, beta2, eps, metrics,
                              workspace, apply, stream);
}


int64_t pc_ppo_epoch_state_floats(int D, int H, int A) {
    g_hip_err.clear();   // (pc_last_hip_error speaks of THIS call)
    if (H != 256 || A < 1 || A > 15 || D < 1 || D > 40) return PC_ERR_UNSUPPORTED;
    const int64_t n_param = 2 * ((int64_t)H * D + H) + (int64_t)A * H + A + H + 1;
    return 3 * ((n_param + 3) & ~(int64_t)3);
}

int pc_ppo_epoch_prepared(int device, const float* prepared, int n_mb, int B, int D, int H, int A, float* param, float* grad, float* exp_avg,
                          float* exp_avg_sq, float* step_count, const float* lr_dev, double clip_ratio, double vf_coef, double ent_coef,
                          double max_norm, double beta1, double beta2, double eps, float* metrics, float* workspace, float* state2,
                          void* stream) {
    g_hip_err.clear();   // (pc_last_hip_error speaks of THIS call)
    if (!prepared || !param || !grad || !exp_avg || !exp_avg_sq || !step_count || !lr_dev || !metrics || !workspace || !state2 || n_mb < 1)
        return PC_ERR_INVALID_ARG;
    if (H != 256 || A < 1 || A > 15 || D < 1 || D > 40 || B < 2 || B > 1024) return PC_ERR_UNSUPPORTED;
    if ((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq | (uintptr_t)state2) & 15) != 0) return PC_ERR_INVALID_ARG;   // 16-byte loads / stores
    DeviceGuard guard(device);
    if (!guard.ok) return PC_ERR_NO_DEVICE;
    const MbPlan pl(B, D, H, A, workspace);
    hipStream_t st = (hipStream_t)stream;
    // two generations of (param, exp_avg, exp_avg_sq): the caller's tensors and `state2`
    float* P[2] = {param, state2};
    float* M[2] = {exp_avg, state2 + pl.n_pad};
    float* V[2] = {exp_avg_sq, state2 + 2 * (size_t)pl.n_pad};
    const int64_t pf = (int64_t)B * (D + 4) + 4;
    int cur = 0;
    const bool defer = mb_defer_shape(D, A);     // (other shapes: the generic kernels, three launches per minibatch -- the same bits)
    for (int m = 0; m < n_mb; ++m) {
        AdamDefer df{};
        if (!defer && m > 0) launch_adam(pl, param, exp_avg, exp_avg_sq, grad, param, exp_avg, exp_avg_sq, step_count, lr_dev, max_norm, beta1, beta2, eps, st);
        if (defer && m > 0) {     // the previous minibatch's gradient is applied by this launch as it loads the parameters
            df.grad = grad;
            df.m_in = M[cur];
            df.v_in = V[cur];
            df.p_out = P[cur ^ 1];
            df.m_out = M[cur ^ 1];
            df.v_out = V[cur ^ 1];
            df.norm_partial = pl.norm_partial;
            df.n_norm = pl.n_blk;
            df.step_count = step_count;
            df.lr_dev = lr_dev;
            df.max_norm = (float)max_norm;
            df.beta1 = (float)beta1;
            df.beta2 = (float)beta2;
            df.eps = (float)eps;
        }
        launch_fwdbwd(pl, nullptr, prepared + (size_t)m * pf, B, D, A, nullptr, nullptr, nullptr, nullptr, nullptr, P[cur], clip_ratio, vf_coef, ent_coef,
                      df, st);
        if (defer && m > 0) cur ^= 1;
        launch_reduce(pl, B, vf_coef, ent_coef, grad, metrics, step_count, st);
    }
    // the last gradient, and the state home to the caller's tensors
    launch_adam(pl, P[cur], M[cur], V[cur], grad, param, exp_avg, exp_avg_sq, step_count, lr_dev, max_norm, beta1, beta2, eps, st);
    HIPCHK(hipGetLastError());
    return PC_OK;
}

// ---- pc_xchg: one-shot all-reduce over peer-mapped staging buffers (kernels/exchange.hpp) --------------------------------
}  // extern "C"

struct pc_xchg {
    int device = 0, rank = 0, world = 1;
    int64_t n = 0;
    int n_pad = 0, n_chunks = 0;
    char* local = nullptr;                       // this rank's staging allocation (uncached device memory)
    char* peer[XCHG_MAX_RANKS] = {nullptr};      // every rank's allocation as mapped here (peer[rank] == local)
    bool ipc[XCHG_MAX_RANKS] = {false};          // peer[r] came from hipIpcOpenMemHandle (closed at destroy); false: a pointer of this process
    bool connected = false;
    double timeout_s = 20.0;
    size_t data_bytes() const { return (size_t)2 * world * n_pad * sizeof(float); }
    size_t flag_bytes() const { return (size_t)2 * world * n_chunks * sizeof(unsigned); }
    size_t total_bytes() const { return data_bytes() + flag_bytes() + (size_t)n_chunks * sizeof(unsigned) + 64; }
    XchgView view() const {
        XchgView v;
        for (int r = 0; r < XCHG_MAX_RANKS; ++r) {
            char* b = r < world ? peer[r] : nullptr;
            v.data[r] = reinterpret_cast<float*>(b);
            v.flags[r] = reinterpret_cast<unsigned*>(b ? b + data_bytes() : nullptr);
        }
        v.epoch = reinterpret_cast<unsigned*>(local + data_bytes() + flag_bytes());
        v.error = reinterpret_cast<int*>(local + data_bytes() + flag_bytes() + (size_t)n_chunks * sizeof(unsigned));
        v.rank = rank;
        v.world = world;
        v.n = (int)n;
        v.n_pad = n_pad;
        v.n_chunks = n_chunks;
        v.timeout_ticks = (unsigned long long)(timeout_s * XCHG_TICKS_PER_SECOND);
        return v;
    }
};

extern "C" {

int pc_xchg_create(int device, int rank, int world, int64_t n_floats, pc_xchg** out) {
    g_hip_err.clear();   // (pc_last_hip_error speaks of THIS call)
    if (!out || world < 1 || world > XCHG_MAX_RANKS || rank < 0 || rank >= world || n_floats < 1 || n_floats > (1 << 24)) return PC_ERR_INVALID_ARG;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1 || device < 0 || device >= count) return PC_ERR_NO_DEVICE;
    DeviceGuard guard(device);
    if (!guard.ok) return PC_ERR_NO_DEVICE;
    pc_xchg* x = new (std::nothrow) pc_xchg;
    if (!x) return PC_ERR_INVALID_ARG;
    x->device = device;
    x->rank = rank;
    x->world = world;
    x->n = n_floats;
    x->n_chunks = (int)((n_floats + XCHG_CHUNK - 1) / XCHG_CHUNK);
    x->n_pad = x->n_chunks * XCHG_CHUNK;
    void* p = nullptr;
    // uncached, fine-grained device memory: a peer's stores must be visible to a kernel that is already running here
    hipError_t e = hipExtMallocWithFlags(&p, x->total_bytes(), hipDeviceMallocUncached);
    if (e != hipSuccess) e = hipExtMallocWithFlags(&p, x->total_bytes(), hipDeviceMallocFinegrained);
    if (e != hipSuccess) {
        g_hip_err = std::string("hipExtMallocWithFlags: ") + hipGetErrorString(e);
        delete x;
        return PC_ERR_HIP;
    }
    x->local = static_cast<char*>(p);
    x->peer[rank] = x->local;
    if (hipMemset(p, 0, x->total_bytes()) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        (void)hipFree(p);
        delete x;
        return PC_ERR_HIP;
    }
    x->connected = world == 1;
    *out = x;
    return PC_OK;
}

// The handle a rank publishes = the hipIpcMemHandle_t of its staging buffer (64 bytes) followed by the PCI bus id of the device the
// buffer lives on (NUL-terminated text, hipDeviceGetPCIBusId): a peer resolves THAT to its own ordinal of the device -- the
// pointer a peer gets from hipIpcOpenMemHandle says nothing reliable about the owning device (hipPointerGetAttributes on an IPC
// mapping reports the opener's device or fails).
int pc_xchg_local_handle(pc_xchg* x, void* handle_out) {
    if (!x || !handle_out) return PC_ERR_INVALID_ARG;
    static_assert(sizeof(hipIpcMemHandle_t) == 64 && PC_XCHG_HANDLE_BYTES == 128, "handle layout");
    g_hip_err.clear();
    DeviceGuard guard(x->device);
    if (!guard.ok) return PC_ERR_NO_DEVICE;
    hipIpcMemHandle_t h;
    HIPCHK(hipIpcGetMemHandle(&h, x->local));
    char* out = static_cast<char*>(handle_out);
    memset(out, 0, PC_XCHG_HANDLE_BYTES);
    memcpy(out, &h, sizeof(h));
    HIPCHK(hipDeviceGetPCIBusId(out + sizeof(h), PC_XCHG_HANDLE_BYTES - (int)sizeof(h) - 1, x->device));
    return PC_OK;
}

int pc_xchg_connect(pc_xchg* x, const void* all_handles) {
    if (!x || !all_handles) return PC_ERR_INVALID_ARG;
    DeviceGuard guard(x->device);
    if (!guard.ok) return PC_ERR_NO_DEVICE;
    g_hip_err.clear();
    const char* all = static_cast<const char*>(all_handles);
    // a failed connect leaves the handle as it was: the mappings opened by THIS call are closed again, and no HIP error stays
    // behind for an unrelated later call to report
    std::vector<int> opened;
    auto fail = [&](int code, const std::string& why) {
        for (int r : opened) { (void)hipIpcCloseMemHandle(x->peer[r]); x->peer[r] = nullptr; }
        (void)hipGetLastError();
        g_hip_err = why;
        return code;
    };
    // 1. every peer's DEVICE must be reachable from ours before a kernel of ours writes into its memory: resolve the published
    //    PCI bus id to this process's ordinal and verify / enable peer access -- a distinct error NOW, not a 20-second wait for a
    //    flag that can never arrive
    for (int r = 0; r < x->world; ++r) {
        if (r == x->rank || x->peer[r]) continue;
        char bus[PC_XCHG_HANDLE_BYTES - 64];
        memcpy(bus, all + (size_t)r * PC_XCHG_HANDLE_BYTES + 64, sizeof(bus));
        bus[sizeof(bus) - 1] = 0;
        int pdev = -1;
        const hipError_t be = bus[0] ? hipDeviceGetByPCIBusId(&pdev, bus) : hipErrorInvalidValue;
        if (be != hipSuccess || pdev < 0)
            return fail(PC_ERR_UNSUPPORTED, "pc_xchg_connect: rank " + std::to_string(r) + "'s device (PCI " + std::string(bus) +
                        ") is not visible to this process (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES hide it, or the rank is on another node): "
                        "peer-mapped exchange impossible, use PPOConfig.exchange = \"rccl\"");
        if (pdev == x->device) continue;     // the same device (a one-GPU rehearsal): nothing to enable
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, x->device, pdev) != hipSuccess || !can)
            return fail(PC_ERR_UNSUPPORTED, "pc_xchg_connect: device " + std::to_string(x->device) + " has no peer access to rank " + std::to_string(r) +
                        "'s device " + std::to_string(pdev) + " (PCI " + std::string(bus) + "; xGMI / PCIe P2P unavailable): use PPOConfig.exchange = \"rccl\"");
        const hipError_t pe = hipDeviceEnablePeerAccess(pdev, 0);
        (void)hipGetLastError();     // (hipErrorPeerAccessAlreadyEnabled is sticky otherwise)
        if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled)
            return fail(PC_ERR_HIP, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(pe));
    }
    // 2. map the peers' staging buffers
    for (int r = 0; r < x->world; ++r) {
        if (r == x->rank || x->peer[r]) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, all + (size_t)r * PC_XCHG_HANDLE_BYTES, sizeof(h));
        void* q = nullptr;
        const hipError_t oe = hipIpcOpenMemHandle(&q, h, hipIpcMemLazyEnablePeerAccess);
        if (oe != hipSuccess || !q) return fail(PC_ERR_HIP, std::string("hipIpcOpenMemHandle(rank ") + std::to_string(r) + "): " + hipGetErrorString(oe));
        x->peer[r] = static_cast<char*>(q);
        x->ipc[r] = true;
        opened.push_back(r);
    }
    x->connected = true;
    return PC_OK;
}

// The same exchange with every rank's handle in THIS process (one process driving several devices, or -- the tests' use -- several
// ranks on one device, each on its own stream): the peers' staging buffers are plain pointers here, no IPC handle is involved.
int pc_xchg_connect_local(pc_xchg* x, pc_xchg* const* ranks) {
    g_hip_err.clear();   // (pc_last_hip_error speaks of THIS call)
    if (!x || !ranks) return PC_ERR_INVALID_ARG;
    for (int r = 0; r < x->world; ++r) {
        const pc_xchg* q = ranks[r];
        if (!q || q->rank != r || q->world != x->world || q->n != x->n || (r == x->rank && q != x)) return PC_ERR_INVALID_ARG;
    }
    DeviceGuard guard(x->device);
    if (!guard.ok) return PC_ERR_NO_DEVICE;
    for (int r = 0; r < x->world; ++r) {
        if (r == x->rank || ranks[r]->device == x->device) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, x->device, ranks[r]->device) != hipSuccess || !can) {
            g_hip_err = "pc_xchg_connect_local: device " + std::to_string(x->device) + " has no peer access to device " + std::to_string(ranks[r]->device);
            return PC_ERR_UNSUPPORTED;
        }
        const hipError_t pe = hipDeviceEnablePeerAccess(ranks[r]->device, 0);
        (void)hipGetLastError();
        if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) {
            g_hip_err = std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(pe);
            return PC_ERR_HIP;
        }
    }
    for (int r = 0; r < x->world; ++r)
        if (r != x->rank) { x->peer[r] = ranks[r]->local; x->ipc[r] = false; }
    x->connected = true;
    return PC_OK;
}

int pc_xchg_set_timeout(pc_xchg* x, double seconds) {
    g_hip_err.clear();   // (pc_last_hip_error speaks of THIS call)
    if (!x || !(seconds > 0.0) || seconds > 3600.0) return PC_ERR_INVALID_ARG;
    x->timeout_s = seconds;
    return PC_OK;
}

int pc_xchg_allreduce(pc_xchg* x, float* bucket, void* stream) {
    g_hip_err.clear();   // (pc_last_hip_error speaks of THIS call)
    if (!x || !bucket) return PC_ERR_INVALID_ARG;
    if (!x->connected) return PC_ERR_INVALID_ARG;
    DeviceGuard guard(x->device);
    if (!guard.ok) return PC_ERR_NO_DEVICE;
    hipLaunchKernelGGL(xchg_allreduce_kernel, dim3(x->n_chunks), dim3(256), 0, (hipStream_t)stream, x->view(), bucket);
    HIPCHK(hipGetLastError());
    return PC_OK;
}

int pc_xchg_allreduce_group(pc_xchg* const* ranks, float* const* buckets, void* stream) {
    g_hip_err.clear();   // (pc_last_hip_error speaks of THIS call)
    if (!ranks || !buckets || !ranks[0]) return PC_ERR_INVALID_ARG;
    const int W = ranks[0]->world;
    XchgGroup g = {};
    for (int r = 0; r < W; ++r) {
        const pc_xchg* x = ranks[r];
        // one device, one launch: every rank's handle connected in this process, all on the launching device
        if (!x || !buckets[r] || x->rank != r || x->world != W || x->n != ranks[0]->n || !x->connected || x->device != ranks[0]->device) return PC_ERR_INVALID_ARG;
        for (int q = 0; q < W; ++q)
            if (x->peer[q] != ranks[q]->local) return PC_ERR_INVALID_ARG;      // (connected to THESE handles: pc_xchg_connect_local)
        g.v[r] = x->view();
        g.bucket[r] = buckets[r];
    }
    DeviceGuard guard(ranks[0]->device);
    if (!guard.ok) return PC_ERR_NO_DEVICE;
    // a workgroup waits for the same chunk's workgroups of the other ranks: the whole grid must be resident at once
    int per_cu = 0, cus = 0;
    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, xchg_allreduce_group_kernel, 256, 0));
    HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ranks[0]->device));
    if ((int64_t)ranks[0]->n_chunks * W > (int64_t)per_cu * cus) {
        g_hip_err = "pc_xchg_allreduce_group: " + std::to_string((int64_t)ranks[0]->n_chunks * W) + " workgroups cannot be co-resident on " +
                    std::to_string(cus) + " compute units";
        return PC_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(xchg_allreduce_group_kernel, dim3(ranks[0]->n_chunks, W), dim3(256), 0, (hipStream_t)stream, g);
    HIPCHK(hipGetLastError());
    return PC_OK;
}

int pc_xchg_status(pc_xchg* x) {
    g_hip_err.clear();   // (pc_last_hip_error speaks of THIS call)
    if (!x) return PC_ERR_INVALID_ARG;
    DeviceGuard guard(x->device);
    int err = 0;
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(&err, x->view().error, sizeof(int), hipMemcpyDeviceToHost));
    return err ? PC_ERR_TIMEOUT : PC_OK;
}

void pc_xchg_destroy(pc_xchg* x) {
    if (!x) return;
    DeviceGuard guard(x->device);
    (void)hipDeviceSynchronize();
    for (int r = 0; r < x->world; ++r)
        if (r != x->rank && x->peer[r] && x->ipc[r]) (void)hipIpcCloseMemHandle(x->peer[r]);
    (void)hipFree(x->local);
    delete x;
}

}  // extern "C"
