/*
 * ppocar.h -- C-ABI of the MI355X-native CarEnv hot path (libppocar.so).
 *
 * The reference (ProfessorNova/PPO-Car) is pure Python and has no FFI layer; the boundary
 * this library sits behind is the gymnasium vector-env protocol exactly as train.py uses
 * it, plus Buffer.calculate_advantages.  Each entry point names the reference interface
 * it replaces (file:line in the reference tree).  INTEGRATION.md shows the ctypes stub a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, int status return: 0 = PC_OK, negative = error
 *     (pc_strerror).  No exceptions cross the boundary.
 *   - Array arguments of pc_env_reset/pc_env_step/pc_gae/pc_sample are DEVICE pointers owned
 *     by the caller (e.g. torch tensors) on the device the handle was created for; they must
 *     stay alive until the stream has executed the call.  The library owns only its opaque
 *     handles and the env state inside them.
 *   - Launches are asynchronous on the caller-supplied hipStream_t (`stream`, NULL = the
 *     default stream).  The library never synchronises in reset/step/gae/sample.
 *   - There is NO CPU fallback: every compute entry point needs a gfx950 device and fails
 *     with PC_ERR_NO_DEVICE / PC_ERR_HIP otherwise.
 *   - One handle per device; a handle is not thread-safe; different handles are independent.  The library keeps NO process-wide
 *     state: every launch option lives in a handle (pc_policy: arithmetic form and work decomposition of the policy step;
 *     pc_env: pc_env_set_option).
 */
#ifndef PPOCAR_H
#define PPOCAR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PC_OK 0
#define PC_ERR_INVALID_ARG (-1)
#define PC_ERR_IO (-2)          /* track file missing / unreadable (the reference prints and returns None, car_env.py:624-628; here it is a hard error) */
#define PC_ERR_PARSE (-3)       /* track JSON malformed or schema violated */
#define PC_ERR_HIP (-4)         /* a HIP runtime call failed */
#define PC_ERR_UNSUPPORTED (-5) /* e.g. more rays than the kernel menu covers */
#define PC_ERR_NO_DEVICE (-6)   /* no usable gfx950 device */
#define PC_ERR_TIMEOUT (-7)     /* pc_xchg: a peer rank did not arrive at the gradient exchange */

#define PC_DTYPE_F32 0 /* float64 kinematic state; every ray's wall segment SELECTED in float32, its distance (the observation
                        * entry, the < 10 px tests) recomputed in float64 under the reference's strict test: the throughput path */
#define PC_DTYPE_F64 1 /* float64 throughout, in the reference's operation order, with glibc's cos / sin values looked up for every
                        * angle an episode can reach (up to 16 tracks per handle; other angles use the device's): the exact-parity
                        * path -- observations, rewards, events and the float64 state equal the reference's bit for bit.  (Where a
                        * float32 sweep runs first it only chooses WHICH wall's literal cast is evaluated; every produced value is
                        * the literal arithmetic's.) */

typedef struct pc_track pc_track;
typedef struct pc_env pc_env;
typedef struct pc_policy pc_policy;
typedef struct pc_xchg pc_xchg;

/* ---- track data: CarEnv.load_track (car_env.py:535-567) + the wall / gate lists that
 * CarEnv.reset builds (car_env.py:651-676).  Host-side, no GPU needed. ------------------ */

/* Parse a track JSON file (schema written by track_editor.py:50-56,126-127): scales x by
 * 1280 and y by 720, builds walls = outer segments then inner segments and gates = consecutive
 * point pairs. */
int pc_track_load_json(const char* path, pc_track** out);
/* Same from memory: walls [S][4], gates [G][4] = x1,y1,x2,y2 in pixels. */
int pc_track_from_arrays(const double* walls, int n_walls, const double* gates, int n_gates, double start_x,
                         double start_y, double start_angle_deg, pc_track** out);
/* *n_walls, *n_gates, start[3] = {x, y, angle_deg}; any pointer may be NULL. */
int pc_track_info(const pc_track* t, int* n_walls, int* n_gates, double* start);
/* Copy the geometry out (host): walls [S][4], gates [G][4]; either may be NULL. */
int pc_track_geometry(const pc_track* t, double* walls, double* gates);
void pc_track_destroy(pc_track* t);

/* ---- the vector environment: gym.vector.AsyncVectorEnv([make_env]*N) (train.py:138-139)
 * of CarEnv (car_env.py:472-760) wrapped in TransformReward (train.py:65,68). ------------ */

/* number of rays the reference generates for a nominal `num_rays`: len(range(0, 360, 360 // n))
 * (car_env.py:269) -- 12 -> 12, 16 -> 17, 32 -> 33.  Negative on invalid n. */
int pc_ray_count(int num_rays_nominal);

/* Create n_envs environments on `device` (HIP ordinal).  `tracks`/`n_tracks`: the track table;
 * `track_id` (host, [n_envs], may be NULL = all on track 0) picks each env's track -- the
 * reference allows a different track per env through reset(options=...) (car_env.py:621-628).
 * Replaces: CarEnv.__init__ (car_env.py:475-533) x N + AsyncVectorEnv construction.
 * F32 handles check what their float32 selector assumes of a track (car_env.py:155-184 puts no constraint on the walls): the
 * walls' bounding box must fit 2000 px and a track may have at most 8192 chain vertices (PC_ERR_UNSUPPORTED; pc_last_hip_error
 * names the track's index and which of the two limits it breaks: use PC_DTYPE_F64); walls that cross or touch without being chain neighbours, spikes sharper than ~13 degrees and
 * walls shorter than the corner margin are accepted and resolved exactly (every ray that selects one of them takes the float64
 * scan of the whole chain): slower on those rays, never different from the reference. */
int pc_env_create(int device, int64_t n_envs, int num_rays_nominal, const pc_track* const* tracks, int n_tracks,
                  const uint8_t* track_id, int dtype, pc_env** out);
void pc_env_destroy(pc_env* e); /* envs.close() (train.py:296) */

int pc_env_obs_dim(const pc_env* e);    /* 6 + R: envs.single_observation_space.shape[0] (train.py:141) as the reference actually produces it */
int pc_env_num_actions(const pc_env* e); /* 9: envs.single_action_space.n (train.py:142, car_env.py:525) */
int64_t pc_env_num_envs(const pc_env* e);

/* envs.reset(options={"track_path": p}) (train.py:159; CarEnv.reset car_env.py:605-691):
 * every env to its track's start state; obs [n_envs][D] float32 (device). */
int pc_env_reset(pc_env* e, float* obs, void* stream);

/* envs.step(actions) (train.py:185): CarEnv.step (car_env.py:693-760) for every env, reward *
 * reward_scale (TransformReward, train.py:65,68; computed in float64 then rounded to float32
 * exactly as `rew_buf[ptr] = rew` does, buffer.py:29), and gymnasium's same-step auto-reset:
 * an env that terminated or truncated is reset and `obs` holds its reset observation.
 *   actions     [N] int64, values 0..8 (anything else is treated as 8 = do nothing, car_env.py:721)
 *   obs         [N][D] float32   -- may point straight into Buffer.obs_buf[t+1]
 *   reward, terminated, truncated  [N] float32 (flags 0.0 / 1.0, what train.py:191-192 builds)
 *   gates_passed [N] int32 or NULL -- info["gates_passed"] of the step (before auto-reset)
 *   final_obs   [N][D] float32 or NULL -- the observation CarEnv.step itself returned
 *                                         (gymnasium's info["final_observation"] on done envs) */
int pc_env_step(pc_env* e, const int64_t* actions, double reward_scale, float* obs, float* reward,
                float* terminated, float* truncated, int32_t* gates_passed, float* final_obs, void* stream);

/* T successive pc_env_step calls as ONE call -- `for t in range(T): envs.step(actions[t])`, the env alone under pre-generated
 * actions (SURVEY 8(d)'s level (i)); train.py:185 without the policy in front of it.  actions [T][N] int64; row t of obs [T][N][D],
 * reward / terminated / truncated [T][N] = what the t-th pc_env_step writes, bit for bit, and the handle's state afterwards is the
 * state after those T calls.  Where the handle has the table-driven form (below) the T steps are ONE launch with the env state in
 * registers throughout; otherwise T launches of the per-step kernel, enqueued by this call. */
int pc_env_step_many(pc_env* e, const int64_t* actions, int64_t T, double reward_scale, float* obs, float* reward,
                     float* terminated, float* truncated, void* stream);
/* Which kernel the last pc_env_step / pc_env_step_many on this handle launched (0 before the first; both fill the same outputs bit for bit):
 *   PC_STEP_K1         env_step_kernel: any ray count 4..360, any track, per-env track ids; 2^k lanes per env chosen from the batch size
 *   PC_STEP_K1F        env_steps_fast_kernel: the table-driven env step of the persistent rollout kernel -- 12 / 16 / 32 nominal rays, the
 *                      track's gather tables staged in LDS per launch (mixed batches: one track per workgroup, or two tracks interleaved in
 *                      evenly split blocks of 64 envs, de-interleaved by wave), 2 lanes per env, the chain-packed / unrolled selector sweep; F64
 *                      handles: its literal form (tracks inside the selector's limits, rotations on the rotation table).  Taken by
 *                      pc_env_step from 8192 envs on, by pc_env_step_many at any batch size; PC_OPT_STEP_FORM decides otherwise
 *   PC_STEP_K1F_TABLE  the same with the track's 1/den table staged too (pc_env_step_many, T > 1, when it fits) */
#define PC_STEP_NONE 0
#define PC_STEP_K1 1
#define PC_STEP_K1F 2
#define PC_STEP_K1F_TABLE 3
int pc_env_last_step_kernel(const pc_env* e);

/* CarEnv._get_info (car_env.py:599-603) of every env's CURRENT state: info["gates_passed"] and info["time_passed"]
 * as the vector env's `infos` hold them after a step (an env auto-reset in that step reports its reset state, 0 / 0;
 * the finished episode's count is pc_env_step's `gates_passed`).  [N] int32 device arrays, either may be NULL. */
int pc_env_info(pc_env* e, int32_t* gates_passed, int32_t* time_passed, void* stream);

/* Per-handle launch options of pc_rollout (below).  Every choice gives bit-identical buffers; they exist so that the variant a
 * benchmark size takes can be checked at other batch sizes, and for A/B timing.
 *   PC_OPT_ROLLOUT_FORM  -1 (default) automatic: above 8192 envs independent waves of 32 envs, 256 envs per workgroup (128 up to
 *                        32768 envs), else 16 / 32 envs per workgroup with the policy's hidden tiles and the wall sweep split over the
 *                        waves; between 8193 and 32768 envs at 16 rays the first form with 16 envs per wave; 0 / 1 force the first /
 *                        second form, 4 the 16-envs-per-wave one (where the shape has it); 2 / 3 = forms 0 / 1 with the env step forming 1/den
 *                        arithmetically instead of reading the track's 1/den table from LDS (what happens anyway when it does not fit)
 *   PC_OPT_ROLLOUT_EPW   envs per workgroup: 0 (default) automatic; 128 / 256 force the big form's choice, 16 / 32 the small form's
 *   PC_OPT_ROLLOUT_FAST  1 (default) = batches whose workgroups each lie on one track take the mode whose gather tables sit in LDS
 *                        behind LDS pointers -- at 16 (17) rays in the kernels compiled for the chain layout of the reference's tracks
 *                        when every track of the batch has it; 2 = that mode but never those specialised kernels; 0 = always the
 *                        generic mode (what mixed-track batches whose workgroups straddle tracks take).  F64 handles: 1 / 2 = the
 *                        selector form (with / without the specialised sweeps), 0 = the filter form.  3 = as 1, but two tracks interleaved
 *                        in evenly split blocks keep the per-track passes inside every wave instead of being de-interleaved by wave
 *                        (what unevenly interleaved batches take anyway; a test / timing knob) */
#define PC_OPT_ROLLOUT_FORM 1
#define PC_OPT_ROLLOUT_EPW 2
#define PC_OPT_ROLLOUT_FAST 3
/*   PC_OPT_STEP_FORM     pc_env_step / pc_env_step_many: 0 (default) automatic (see PC_STEP_K1F), 1 = always PC_STEP_K1, 2 = PC_STEP_K1F
 *                        wherever the handle has it, at any batch size */
#define PC_OPT_STEP_FORM 4
int pc_env_set_option(pc_env* e, int option, int value);
int pc_env_get_option(const pc_env* e, int option, int* value);

/* Teacher forcing for parity tests: copy the env state from / to HOST arrays of length n_envs
 * (any pointer may be NULL).  `rot` is the heading in degrees.  Synchronous.
 * F32 handles store the heading as a count of 5-degree turns from the track's start angle
 * (Car.move_car only ever adds +-5.0, car_env.py:440-442), so set_state rounds `rot` to that grid. */
int pc_env_get_state(pc_env* e, double* px, double* py, double* vx, double* vy, double* rot, int64_t* time_step,
                     int64_t* next_gate, int64_t* passed);
int pc_env_set_state(pc_env* e, const double* px, const double* py, const double* vx, const double* vy,
                     const double* rot, const int64_t* time_step, const int64_t* next_gate, const int64_t* passed);

/* ---- Buffer.calculate_advantages (buffer.py:36-64): GAE(lambda) with separate terminated /
 * truncated masks, float32, same operation order as the reference's torch expression (bit-exact
 * with it).  rew, val, term, trunc, adv, ret: [T][N] row-major; last_*: [N].  All device. */
int pc_gae(int device, const float* rew, const float* val, const float* term, const float* trunc,
           const float* last_val, const float* last_term, const float* last_trunc, double gamma, double lam,
           int64_t T, int64_t N, float* adv, float* ret, void* stream);

/* ---- Agent.get_action_and_value's sampling tail (model.py:35-40): for logits [N][A] float32
 * draw action ~ Categorical(logits) and return log_prob(action) and (optionally) the entropy.
 * Counter-based RNG (Philox-4x32-10) keyed by (seed, offset): the same (seed, offset, N, A)
 * gives the same actions on any launch geometry.  actions [N] int64, logprob/entropy [N] float32
 * (entropy may be NULL).  All device. */
int pc_sample(int device, const float* logits, int64_t N, int A, uint64_t seed, uint64_t offset, int64_t* actions,
              float* logprob, float* entropy, void* stream);

/* ---- the whole of Agent.get_action_and_value(x) as the rollout calls it (model.py:34-41, train.py:181):
 * both 1-hidden-layer MLPs (actor D->H->A, critic D->H->1, ReLU), the categorical draw, log_prob and the
 * value, in one launch on the matrix cores.  A policy step's configuration is a HANDLE: shape (D, H, A), arithmetic form of the
 * two GEMMs and work decomposition.  Handles are immutable and independent: two of them with different forms can pack and run
 * side by side in one process; a weight image belongs to the handle that packed it.
 *   precision  2 (= -1, the default) fp16x2 in scaled domains: every fp32 operand v = h + l, h = fp16(v), l = fp16(v - h); a product
 *                is a_h b_l + a_l b_h + a_h b_h into ONE fp32 accumulator (v_mfma_f32_16x16x32_f16, three per K block); max error
 *                vs float64 on this MLP 1.1e-7 (a plain fp32 GEMM: 0.8e-7); operands saturate at fp16's finite range in their
 *                scaled domain.  Needs D <= 40, A <= 9, else the shape gets form 0.
 *              1 bf16x3: three bf16 pieces per operand, six piece products (v_mfma_f32_16x16x32_bf16); 1.0e-7; same shapes.
 *              0 fp32-input MFMA (v_mfma_f32_16x16x4_f32), bit-for-bit an fp32 fmaf chain.
 *   split      -1 automatic (hidden tiles split across the waves of a workgroup up to 8192 envs), 0 never, 1 always; the two
 *              forms differ in fp32 summation order (last-bit differences).
 * pc_policy_create: PC_ERR_UNSUPPORTED unless H == 256, A <= 15, D <= 40 (the caller then uses its own GEMMs + pc_sample).
 * pc_policy_get reports the form the shape actually got and the number of floats its weight image needs.
 * The weights do not change during a rollout, so they are packed ONCE into the kernel's LDS image:
 *   pc_policy_pack  builds the image (device buffer `image`) from the eight torch.nn.Linear tensors: aW1 [H][D], ab1 [H],
 *                   aW2 [A][H], ab2 [A], cW1 [H][D], cb1 [H], cW2 [1][H], cb2 [1]
 *   pc_policy_pack_checked  the same, and *range_status (a DEVICE int32, written asynchronously on `stream`) := 0 or a mask of
 *                   PC_POLICY_RANGE_*: the weights leave precision 2's numeric domain -- a weight saturates in its scaled fp16 domain
 *                   (|W1| > 4094, |W2| > 1023.5), or a hidden unit can reach the hidden layer's saturation point 255.87 on observations
 *                   within +-4 (|b1| + 4 sum_j |W1_uj| > 255.87; CarEnv's observations lie in [-1, 1.6]).  With the status 0 NO operand of
 *                   the policy pass can saturate, so the rollout kernels carry no run-time check.  A set bit means: results of
 *                   pc_policy_act / pc_rollout with this image are NOT model.py's within 4e-6 -- pack with a precision-0 handle instead
 *                   (the host layer does: Agent.pack_policy).  Forms 0 and 1 have no scaled domains: always 0.
 *   pc_policy_act   one policy step for obs [N][D] using the image.  RNG as pc_sample, with offset = `offset` + *offset_dev when
 *                   offset_dev != NULL (a device counter, so a captured HIP graph can be replayed with a fresh stream of draws).
 *                   action [N] int64; action_f32 [N] (the float copy Buffer.act_buf stores, buffer.py:13) or NULL; logprob,
 *                   value [N]; logits_out [N][A] or NULL. */
int pc_policy_create(int device, int D, int H, int A, int precision, int split, pc_policy** out);
void pc_policy_destroy(pc_policy* p);
int pc_policy_get(const pc_policy* p, int* precision, int* split, int64_t* image_floats);
int pc_policy_pack(const pc_policy* p, const float* aW1, const float* ab1, const float* aW2, const float* ab2, const float* cW1,
                   const float* cb1, const float* cW2, const float* cb2, float* image, void* stream);
#define PC_POLICY_RANGE_W1 1       /* a first-layer weight (actor or critic) saturates in its scaled fp16 domain */
#define PC_POLICY_RANGE_W2 2       /* an actor output-layer weight does */
#define PC_POLICY_RANGE_HIDDEN 4   /* a hidden activation can */
int pc_policy_pack_checked(const pc_policy* p, const float* aW1, const float* ab1, const float* aW2, const float* ab2, const float* cW1,
                           const float* cb1, const float* cW2, const float* cb2, float* image, int32_t* range_status, void* stream);
int pc_policy_act(const pc_policy* p, const float* obs, int64_t N, const float* image, uint64_t seed, uint64_t offset,
                  const uint64_t* offset_dev, int64_t* action, float* action_f32, float* logprob, float* value, float* logits_out,
                  void* stream);

/* ---- the whole rollout of one epoch (train.py:173-195) as ONE persistent launch: for t in 0..T-1
 *   (action, logprob, value) = Agent.get_action_and_value(obs_t)      [pc_policy_act's arithmetic and RNG, offset + t]
 *   obs_{t+1}, reward, terminated, truncated = envs.step(action)      [pc_env_step's arithmetic]
 *   Buffer.store(...)                                                 [rows written in place]
 * Inputs: the env handle (single track, or mixed tracks with every aligned block of 32 envs on one track -- when the
 * blocks are as large as the launch's workgroups, 128 / 256 envs in the large form and 16 / 32 in the small one, each
 * workgroup stages its track's tables in LDS as for a single track, else every wave reads them from global memory), the
 * policy handle and the weight image it packed, and next_obs / next_term / next_trunc [N] = observation and flags the rollout
 * starts from (the caller has copied them into row 0 of obs_buf / term_buf / trunc_buf, as Trainer does).
 * Outputs: obs_buf [T][N][D] rows 1..T-1, act_buf (float32, buffer.py:13), rew_buf, val_buf, logprob_buf [T][N] rows 0..T-1,
 * term_buf / trunc_buf rows 1..T-1, and next_obs / next_term / next_trunc overwritten with the state after step T-1 -- bit-identical
 * to T x (pc_policy_act; pc_env_step) -- plus two per-env outputs [N] float32 (each may be NULL) so that an epoch needs nothing
 * else between the rollout and Buffer.calculate_advantages:
 *   last_value : the critic's value of the FINAL observation -- agent.get_value(next_obs), train.py:200 -- from one more policy
 *                pass inside the launch (the fused policy step's arithmetic, i.e. what val_buf's rows hold for the other steps);
 *   reward_sum : the sum over the T steps of the env's (scaled) rewards, accumulated in float32 in step order -- the numerator
 *                of train.py:272's average reward without re-reading rew_buf.
 * F64 handles (PC_DTYPE_F64) take the same call: the launch then steps the env in the reference's own float64 (the per-step
 * kernel's env_step arithmetic, bit for bit: observations, rewards, events and the float64 state equal the reference's) -- Discrete(9),
 * the split-operand policy forms, 12, 16 or 32 nominal rays; other shapes are PC_ERR_UNSUPPORTED (the caller's per-step kernels run them).
 * Two kernels fill the same bits (pc_env_last_rollout_kernel says which ran): by default the selector form -- K9 with a float32
 * sweep that only SELECTS each ray's wall and the reference's literal arithmetic on that wall (fp16 x 2 policy arithmetic, tracks of
 * at most 8192 chain vertices inside 2000 px, every env's rotation one that reset and stepping produce; up to 8192 envs at 12 / 16 rays
 * inside the small form, whose policy arithmetic is the split policy step's) --, else the filter form,
 * which tests every (ray, wall) pair in float64 (any track, any state, bf16 x 3 at 12 rays; not built for 32 rays); tracks of more
 * than 64 chain vertices and rotations set off the table take the generic kernel with the selector step (PC_KERNEL_K9D_SELECTOR).
 * With a precision-0 policy handle an F64 handle at 16 rays runs the literal form with the fp32 weight image: every number of the
 * rollout in the reference's own arithmetic, still one launch.
 * Mixed-track handles: track ids constant inside aligned blocks of 32 envs (or more: whole workgroups) run the fast modes per block;
 * track ids that change INSIDE a block (car_env.py:621-628 puts every env on its own track: e.g. track_id = i & 1) run the big form with
 * the env step once per track present in a wave -- two tracks of the reference's layout (two equal loops of 13 or 9 chain vertices) at
 * 16 rays in the table-driven two-track form (both tracks' tables in LDS; F32 and F64 handles), anything else in the generic mode.
 * When every aligned block of 64 envs (32 in the 16-envs-per-wave form) is split evenly between the two tracks -- i & 1 is -- the
 * block's two waves DE-INTERLEAVE it: each wave steps the block's envs of ONE track (one pass; an env's rows, state and random stream stay
 * the env's own, so the buffers are the same bits).
 * PC_ERR_UNSUPPORTED for ray
 * counts whose slots per lane are not on the kernel menu (12 / 16 / 32 run the table-driven fast mode; 17 and 18 share the
 * slots of 16 and run the generic mode), shapes whose LDS footprint exceeds 160 KB (33 rays with the fp32 or bf16x3 weight
 * image in the large form), and 33 rays in the GENERIC mode with split operands (a mixed-track batch whose workgroups
 * straddle tracks, or the fast mode switched off: those kernels spilled and are not built): callers fall back to the
 * two-kernel loop, which fills the same buffers bit for bit. */
int pc_rollout(pc_env* e, const pc_policy* p, const float* image, int64_t T, double reward_scale, uint64_t seed, uint64_t offset,
               const uint64_t* offset_dev, float* obs_buf, float* act_buf, float* rew_buf, float* val_buf, float* term_buf,
               float* trunc_buf, float* logprob_buf, float* next_obs, float* next_term, float* next_trunc, float* last_value,
               float* reward_sum, void* stream);

/* ---- the non-GEMM work of one PPO minibatch step (train.py:230-261), three launches:
 * pc_ppo_gather : traj_*[batch_indices] (train.py:233-238,249): idx [B] int64 into the flattened trajectories
 *                 obs [M][D], act / logprob / adv / ret [M]  ->  o_obs [B][D], o_act / o_logprob / o_adv / o_ret [B].
 * pc_ppo_loss   : the clipped-PPO loss (train.py:235-255: ratio, per-minibatch advantage normalisation with the
 *                 unbiased std, max(-A r, -A clamp(r)), 0.5 (v - ret)^2, entropy) given the network outputs
 *                 logits [B][A], values [B]; writes its gradients dlogits [B][A], dvalues [B] (what autograd would
 *                 hand to the two MLPs) and adds (policy_loss, value_loss, entropy, total) to metrics[4]
 *                 (train.py:263-266).  2 <= B <= 1024.
 * pc_clip_adam  : nn.utils.clip_grad_norm_(max_norm) (train.py:260) + Adam.step() (train.py:261; eps/betas as
 *                 train.py:146 configures) over flat float32 buffers of n elements; lr and the step counter live on
 *                 the device (HIP-graph replay); grad_scale = 1/world_size folds the gradient average in. */
int pc_ppo_gather(int device, const int64_t* idx, int B, int D, const float* obs, const float* act, const float* logprob,
                  const float* adv, const float* ret, float* o_obs, float* o_act, float* o_logprob, float* o_adv, float* o_ret,
                  void* stream);
int pc_ppo_loss(int device, const float* logits, const float* values, const float* act, const float* old_logprob,
                const float* adv, const float* ret, int B, int A, double clip_ratio, double vf_coef, double ent_coef,
                float* dlogits, float* dvalues, float* metrics, void* stream);
int pc_clip_adam(int device, float* param, float* grad, float* exp_avg, float* exp_avg_sq, float* step_count, const float* lr_dev,
                 int64_t n, double max_norm, double grad_scale, double beta1, double beta2, double eps, void* stream);
/* The same step for the multi-rank form of pc_ppo_minibatch (apply = 2: the gradient kernels have already advanced the step
 * counter): n / 256 workgroups, each summing the bucket's squares itself in one fixed order (replicas stay bit-identical).
 * `grad` = the all-reduced SUM over ranks, left untouched; grad_scale = 1 / world_size.  (train.py:260-261) */
int pc_clip_adam_advanced(int device, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const float* step_count,
                          const float* lr_dev, int64_t n, double max_norm, double grad_scale, double beta1, double beta2, double eps,
                          void* stream);

/* ---- one whole PPO minibatch step (train.py:230-261) with no library GEMM: gather, both MLPs forward, the clipped-PPO
 * loss, both MLPs backward, and (apply != 0) clip_grad_norm_ + Adam, as three launches.  `param` / `grad` / `exp_avg` /
 * `exp_avg_sq` are flat float32 buffers in torch's module.parameters() order (actor.0.weight [H][D], actor.0.bias,
 * actor.2.weight [A][H], actor.2.bias, critic.0.weight, critic.0.bias, critic.2.weight [1][H], critic.2.bias); idx [B]
 * indexes the flattened trajectories obs [M][D], act / old_logprob / adv / ret [M].  grad receives the (clipped, when
 * applied) gradient; metrics[4] += (policy_loss, value_loss, entropy, total); step_count [1] float and lr_dev [1] live on
 * the device.  apply == 0 stops after the gradient; apply == 2 stops after the gradient but advances the step counter
 * (multi-rank: all-reduce the gradient, then pc_clip_adam_advanced).  workspace: device
 * buffer of pc_ppo_workspace_floats(B, D, H, A) floats.  Deterministic (fixed summation order, no atomics).
 * PC_ERR_UNSUPPORTED unless H == 256, A <= 15, D <= 40, 2 <= B <= 1024. */
int64_t pc_ppo_workspace_floats(int B, int D, int H, int A);
int pc_ppo_minibatch(int device, const int64_t* idx, int B, int D, int H, int A, const float* obs, const float* act,
                     const float* old_logprob, const float* adv, const float* ret, float* param, float* grad, float* exp_avg,
                     float* exp_avg_sq, float* step_count, const float* lr_dev, double clip_ratio, double vf_coef, double ent_coef,
                     double max_norm, double beta1, double beta2, double eps, float* metrics, float* workspace, int apply,
                     void* stream);

/* The same step on a minibatch gathered beforehand.  pc_ppo_prepare gathers n_mb minibatches in ONE launch: minibatch m
 * reads its B indices at idx[m * idx_ld ...] and writes, at prepared + m * pc_ppo_prepared_floats(B, D), the sample rows
 * obs[idx] [B][D], then act / old_logprob / adv / ret [B] each, then (mean, max(unbiased std, 1e-5)) of its advantages and
 * two pad floats -- what every workgroup of pc_ppo_minibatch otherwise fetches (an index load, then the dependent row
 * loads: two cold misses at the head of its critical path) and reduces for itself.  pc_ppo_minibatch_prepared(one such
 * block) == pc_ppo_minibatch on the same samples, bit for bit.  (train.py:225-240) */
int64_t pc_ppo_prepared_floats(int B, int D);
int pc_ppo_prepare(int device, const int64_t* idx, int64_t idx_ld, int n_mb, int B, int D, const float* obs, const float* act,
                   const float* old_logprob, const float* adv, const float* ret, float* prepared, void* stream);
int pc_ppo_minibatch_prepared(int device, const float* prepared_mb, int B, int D, int H, int A, float* param, float* grad, float* exp_avg,
                              float* exp_avg_sq, float* step_count, const float* lr_dev, double clip_ratio, double vf_coef,
                              double ent_coef, double max_norm, double beta1, double beta2, double eps, float* metrics,
                              float* workspace, int apply, void* stream);


/* The whole minibatch loop of one epoch (train.py:223-261) over n_mb prepared minibatches (consecutive blocks of
 * pc_ppo_prepared_floats(B, D) floats at `prepared`), with the clip + Adam step of minibatch i taken by the forward / backward
 * launch of minibatch i + 1 as it loads the parameters (every workgroup needs all of them anyway; workgroup 0 writes the new
 * generation into the other of two state buffers): TWO launches per minibatch instead of three, plus one clip + Adam launch for
 * the last gradient, which also brings the state home.  Bit-identical to n_mb x pc_ppo_minibatch_prepared(apply = 1): param,
 * exp_avg, exp_avg_sq, step_count and metrics; `grad` ends as the LAST minibatch's clipped gradient.  state2: device scratch of
 * pc_ppo_epoch_state_floats(D, H, A) floats; param / grad / exp_avg / exp_avg_sq / state2 16-byte aligned.  Single-rank only
 * (the multi-rank step has the gradient exchange between the backward pass and the clip). */
int64_t pc_ppo_epoch_state_floats(int D, int H, int A);
int pc_ppo_epoch_prepared(int device, const float* prepared, int n_mb, int B, int D, int H, int A, float* param, float* grad, float* exp_avg,
                          float* exp_avg_sq, float* step_count, const float* lr_dev, double clip_ratio, double vf_coef, double ent_coef,
                          double max_norm, double beta1, double beta2, double eps, float* metrics, float* workspace, float* state2,
                          void* stream);

/* ---- the per-minibatch gradient exchange (one all-reduce(SUM) of the flat gradient bucket between
 * loss.backward() and clip_grad_norm_, train.py:259-260) as a ONE-SHOT all-reduce over peer-mapped buffers -- the
 * latency-proof alternative to an RCCL all_reduce for a 49 - 92 KB message: every rank writes its bucket straight into a
 * slot of every peer's staging buffer (hipIpc-mapped device memory: W - 1 independent xGMI writes), raises an arrival flag,
 * waits for its own W flags and sums the W slots locally in rank order, so the reduced buckets are bit-identical on all
 * ranks.  One process per rank (ranks of one node; up to 8):
 *   pc_xchg_create(device, rank, world, n_floats)   allocates this rank's staging buffer (uncached device memory);
 *   pc_xchg_local_handle(x, out)                    PC_XCHG_HANDLE_BYTES bytes for the other ranks: the hipIpcMemHandle_t of the
 *                                                   staging buffer, then the PCI bus id of its device (text) -- the caller carries
 *                                                   them across (e.g. torch.distributed.all_gather_object);
 *   pc_xchg_connect(x, all)                         `all` = world x PC_XCHG_HANDLE_BYTES bytes in rank order: resolves every peer's
 *                                                   PCI bus id to this process's device ordinal, verifies / enables peer access
 *                                                   to it (PC_ERR_UNSUPPORTED with a message in pc_last_hip_error when the device
 *                                                   is not visible to this process or not reachable), THEN maps the peers'
 *                                                   buffers; a failed connect closes what it opened and leaves the handle as it was;
 *   pc_xchg_connect_local(x, ranks)                 the in-process form: `ranks` = the world's pc_xchg handles in rank order, all created
 *                                                   in THIS process (one process driving several devices -- peer access is verified /
 *                                                   enabled -- or several ranks on one device, each launching on its own stream);
 *   pc_xchg_allreduce_group(ranks, buckets, stream) the exchanges of ALL ranks of such an in-process group on ONE device as one launch
 *                                                   (bucket[r] := the rank-ordered sum, for every r): the ranks' workgroups are then
 *                                                   co-resident by construction -- separate launches on separate streams are not (HIP
 *                                                   multiplexes streams onto a few hardware queues) and must not be used with more
 *                                                   ranks per device than hardware queues; PC_ERR_UNSUPPORTED when world x
 *                                                   ceil(n_floats / 1024) workgroups exceed what the device holds at once;
 *   pc_xchg_set_timeout(x, seconds)                 patience of a wait inside the exchange kernel (default 20 s);
 *   pc_xchg_allreduce(x, bucket, stream)            in place, asynchronous on `stream`, capturable into a HIP graph: bucket[0..n)
 *                                                   := sum over ranks (rank order) of their buckets.  Every rank must make the
 *                                                   same sequence of calls;
 *   pc_xchg_status(x)                               synchronises the device; PC_ERR_TIMEOUT if a wait inside any call gave up
 *                                                   (the timeout passed without a peer's flag: the kernel then finishes with a
 *                                                   WRONG sum rather than hang the GPU, and later calls on the handle do not wait
 *                                                   again): the caller must check it wherever it synchronises and abort the job;
 *   pc_xchg_destroy(x)                              after every rank has finished using it (the caller synchronises the ranks). */
#define PC_XCHG_HANDLE_BYTES 128
int pc_xchg_create(int device, int rank, int world, int64_t n_floats, pc_xchg** out);
int pc_xchg_local_handle(pc_xchg* x, void* handle_out);
int pc_xchg_connect(pc_xchg* x, const void* all_handles);
int pc_xchg_connect_local(pc_xchg* x, pc_xchg* const* ranks);
int pc_xchg_allreduce_group(pc_xchg* const* ranks, float* const* buckets, void* stream);
int pc_xchg_set_timeout(pc_xchg* x, double seconds);
int pc_xchg_allreduce(pc_xchg* x, float* bucket, void* stream);
int pc_xchg_status(pc_xchg* x);
void pc_xchg_destroy(pc_xchg* x);

const char* pc_strerror(int code);
/* Detail of the last error on this thread: the HIP error string behind a PC_ERR_HIP, or what made pc_env_create answer
 * PC_ERR_UNSUPPORTED (e.g. an F32 handle for a track that does not fit 2000 px).  Empty when there is none. */
const char* pc_last_hip_error(void);
/* 0 in every shipped build.  Non-zero only in the separate developer library `make ablate` builds
 * (libppocar_ablate.so, -DPC_ABLATE=n: timing ablations that skip parts of the rollout kernel); bench.py and the
 * tests refuse to run on such a build.  There is no run-time switch that makes a kernel do less work. */
int pc_build_ablate(void);
/* Kernel-launch geometry of the last pc_env_step on this handle (lanes per env, rays per lane,
 * blocks, threads) -- for bench.py / DESIGN.md; any pointer may be NULL. */
int pc_env_launch_info(const pc_env* e, int* lanes_per_env, int* rays_per_lane, int* blocks, int* threads);
/* What pc_env_create made of track `track` of this handle: its wall count (car_env.py:653-670), the vertices of its wall chains,
 * and -- F32 handles -- how many wall segments carry the "resolve by the float64 chain scan" mark (walls that cross or touch
 * without being chain neighbours, spikes, walls shorter than the corner margin: exact, but every ray that selects one of them
 * costs an O(n_walls) float64 scan).  A track where that is a large share of the walls runs correctly and SLOWLY in dtype f32: the
 * Python host layer warns above 25 %.  F64 handles: the same count where the persistent kernel's selector form can run on the track
 * (PC_KERNEL_K9_LITERAL), else 0.  Any pointer may be NULL. */
int pc_env_track_info(const pc_env* e, int track, int* n_walls, int* n_chain_vertices, int* n_scan_segments);
/* Which persistent kernel the last successful pc_rollout on this handle launched (0 before the first; for bench.py, the tests and
 * DESIGN.md -- every kernel fills the same buffers bit for bit, this only says which one did):
 *   PC_KERNEL_K9          big form, float32-selector env step with float64 refinement (F32 handles)
 *   PC_KERNEL_K9M / _LITERAL  K9 with 16 envs per wave (4 lanes per env): 8193 .. 32768 envs at 16 rays (F32 / F64 handles)
 *   PC_KERNEL_K9S         small form (F32 handles, small batches)
 *   PC_KERNEL_K9_LITERAL  big form on an F64 handle: the float32 sweep selects each ray's wall, the reference's literal float64
 *                         arithmetic measures it (12 / 16 / 32 nominal rays, tracks inside the selector's limits, every env's rotation
 *                         on the track's rotation table: what reset and stepping produce)
 *   PC_KERNEL_K9S_LITERAL the same literal form inside the small form (F64 handles, 12 / 16 nominal rays, up to 8192 envs)
 *   PC_KERNEL_K9D_SELECTOR F64 handle, the generic kernel with the per-step kernel's selector step (sweep over the wall chain in
 *                         global memory + literal cast): tracks of more than 64 chain vertices, bf16 x 3, rotations off the table
 *   PC_KERNEL_K9D_FILTER  F64 handle, the filter form: every (ray, wall) pair in float64 (any track; 12 / 16 nominal rays) */
#define PC_KERNEL_NONE 0
#define PC_KERNEL_K9 1
#define PC_KERNEL_K9S 2
#define PC_KERNEL_K9_LITERAL 3
#define PC_KERNEL_K9D_FILTER 4
#define PC_KERNEL_K9S_LITERAL 5
#define PC_KERNEL_K9D_SELECTOR 6
#define PC_KERNEL_K9M 7
#define PC_KERNEL_K9M_LITERAL 8
int pc_env_last_rollout_kernel(const pc_env* e);
/* Override the lanes-per-env choice (power of two 1..64; 0 = automatic).  Tuning knob for bench.py. */
int pc_env_set_lanes_per_env(pc_env* e, int lanes_per_env);

#ifdef __cplusplus
}
#endif
#endif /* PPOCAR_H */
