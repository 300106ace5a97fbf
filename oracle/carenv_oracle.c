/*
 * oracle/carenv_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, double-precision, scalar restatement of the reference's CarEnv hot path
 * (ProfessorNova/PPO-Car, lib/car_env.py, lib/buffer.py).  Every function cites the
 * reference lines it follows and keeps the reference's OPERATION ORDER so that it is
 * bit-exact against the reference on this toolchain (compile with -ffp-contract=off:
 * numpy float64 scalars never fuse a multiply-add).
 *
 * Who may use it: tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg -- as the
 * checker / the timed CPU baseline only.  The product (ppo-car_amd/) never imports, links or
 * calls anything in this directory.
 *
 * Parity status: PINNED -- tests/test_oracle_golden.py checks every function below against
 * the golden vectors in tests/golden/ (.npz files), which tests/golden/make_golden.py recorded from
 * the unmodified reference classes (CarEnv, Car, Ray, Boundary, Buffer) in this container.
 * Not pinned by any reference artefact: the vector-env semantics of gymnasium 0.29.1
 * (same-step auto-reset, TransformReward) in oc_vec_step -- gymnasium is not under
 * /root/reference; they are restated from its documented behaviour.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define OC_PI 3.141592653589793238462643383279502884 /* NPY_PI */

/* np.radians(x) == x * (NPY_PI / 180.0)  (numpy npymath npy_deg2rad) */
static inline double oc_radians(double deg) { return deg * (OC_PI / 180.0); }

/*
 * np.linalg.norm of a 2-vector = sqrt(x.dot(x)).  numpy routes the dot to cblas_ddot; the
 * OpenBLAS x86-64 ddot tail loop `dot += y[i]*x[i]` is compiled with FMA contraction, so
 * on this machine norm = sqrt(fma(d1, d1, d0*d0)).  oc_norm_mode selects the variant:
 * 1 = fused tail (what the golden vectors pin here), 0 = two roundings.  The two differ by
 * <= 1 ulp of a float64 distance, which never reaches a float32 observation or a `< 10.0`
 * decision in practice; the switch exists so the pin is exact, not approximate.
 */
static int oc_norm_mode = 1;
void oc_set_norm_mode(int m) { oc_norm_mode = m; }

static inline double oc_norm2(double d0, double d1) {
    if (oc_norm_mode == 1) return sqrt(fma(d1, d1, d0 * d0));
    return sqrt(d0 * d0 + d1 * d1);
}

/*
 * Ray.cast (car_env.py:155-184) followed by the distance of Ray.get_distance
 * (car_env.py:205-207): returns 1 and *dist if the ray hits the segment, else 0.
 * (x3,y3) = ray origin, (dx,dy) = ray direction as stored by Ray.update (car_env.py:151-153).
 */
static inline int oc_cast(double x1, double y1, double x2, double y2, double x3, double y3, double dx, double dy,
                          double* dist) {
    double x4 = x3 + dx, y4 = y3 + dy;                                  /* :169 pos + dir */
    double den = (x1 - x2) * (y3 - y4) - (y1 - y2) * (x3 - x4);         /* :171 */
    if (den == 0) return 0;                                             /* :172 */
    double t = ((x1 - x3) * (y3 - y4) - (y1 - y3) * (x3 - x4)) / den;   /* :175 */
    double u = -((x1 - x2) * (y1 - y3) - (y1 - y2) * (x1 - x3)) / den;  /* :176 */
    if (0 < t && t < 1 && u > 0) {                                      /* :178 strict */
        double ptx = x1 + t * (x2 - x1);                                /* :180 */
        double pty = y1 + t * (y2 - y1);                                /* :181 */
        *dist = oc_norm2(x3 - ptx, y3 - pty);                           /* :205 norm(pos - pt) */
        return 1;
    }
    return 0;
}

/* Ray.get_distance over a list of S segments (car_env.py:186-213); segs = [S][4] x1,y1,x2,y2. */
static inline double oc_get_distance(double px, double py, double dx, double dy, const double* segs, int S) {
    double largest = 1000.0;                                            /* :198 */
    for (int s = 0; s < S; ++s) {
        double d;
        if (oc_cast(segs[4 * s], segs[4 * s + 1], segs[4 * s + 2], segs[4 * s + 3], px, py, dx, dy, &d) &&
            d < largest)                                                /* :203-207 */
            largest = d;
    }
    return largest;
}

/* Exported single-ray form, angle in degrees as Ray.__init__ takes it (car_env.py:124-134). */
double oc_ray_distance(double px, double py, double angle_deg, const double* segs, int S) {
    double a = oc_radians(angle_deg);
    return oc_get_distance(px, py, cos(a), sin(a), segs, S);
}

/* number of rays the reference generator produces: len(range(0, 360, 360 // n)) (car_env.py:269) */
int oc_ray_count(int n_nominal) {
    int step = 360 / n_nominal;
    return (360 + step - 1) / step;
}

/* Car.check_collision (car_env.py:376-392) with rays posed at (px,py,rot): any of the rays
 * r in range(0, n, n // 4) closer than 10.0 to the boundary list.  Early exit as the reference. */
static int oc_check_collision(double px, double py, double rot, int n, const double* segs, int S) {
    int step = 360 / n;
    for (int r = 0; r < n; r += n / 4) {                                /* :389 */
        double a = oc_radians(rot + (double)(r * step));                /* ray r has angle rot + r*step (:269,:463-466) */
        if (oc_get_distance(px, py, cos(a), sin(a), segs, S) < 10.0)    /* :387,:390 */
            return 1;
    }
    return 0;
}

/*
 * CarEnv.step (car_env.py:693-760) for n_envs independent environments, SoA state, NO
 * auto-reset.  State on entry must be consistent the way the reference's is: the rays hold
 * the pose (px,py,rot) of the last Car.update.  `destroyed` mirrors Car.__destroyed (set
 * by update :468-469, cleared only by reset :683).
 *   walls [S][4], gates [G][4]  pixel coordinates (load_track :549-565)
 *   obs   [n_envs][6+R] float32 (:595)         reward [n_envs] float64, UNSCALED
 * The active-gate set is {next_gate .. G-1} (a gate is deactivated only when its index
 * equals next_gate_index, :739-741, and all are restored together, :734-737), so
 * RewardGate.is_active() is `g >= next_gate`; the scan itself is the literal one of
 * Car.get_passed_gate (:405-408): first active gate, in index order, that is hit.
 */
void oc_env_step(const double* walls, int S, const double* gates, int G, int n, int64_t n_envs, double* px,
                 double* py, double* vx, double* vy, double* rot, int64_t* time_step, int64_t* next_gate,
                 int64_t* passed, uint8_t* destroyed, const int64_t* action, float* obs, double* reward,
                 uint8_t* terminated, uint8_t* truncated) {
    const int step = 360 / n;
    const int R = oc_ray_count(n);
    const int D = 6 + R;
    for (int64_t e = 0; e < n_envs; ++e) {
        double rw = 0.0;                                                /* :694 */
        const int64_t a = action[e];
        const double old_px = px[e], old_py = py[e], old_rot = rot[e]; /* pose the rays still hold */
        double ax = 0.0, ay = 0.0;                                      /* Car.__acceleration (zeroed at :461) */
        double r = rot[e];
        /* action translation :698-722; forward/backward first (uses the pre-turn heading), then the turn */
        const int fwd = (a == 0 || a == 4 || a == 5), bwd = (a == 1 || a == 6 || a == 7);
        const int left = (a == 2 || a == 4 || a == 6), right = (a == 3 || a == 5 || a == 7);
        if (fwd) {                                                      /* move_car("forward") :423-430 */
            ax = cos(oc_radians(r)) * 0.8;
            ay = sin(oc_radians(r)) * 0.8;
            rw += 0.01;                                                 /* :700,:710,:714 */
        } else if (bwd) {                                               /* move_car("backward") :431-438 */
            ax = -cos(oc_radians(r)) * 0.8;
            ay = -sin(oc_radians(r)) * 0.8;
        }
        if (left) r -= 5.0;                                             /* :440 */
        if (right) r += 5.0;                                            /* :442 */

        /* reward gates :725-741 -- rays are still at the pre-step pose (old_px, old_py, old_rot) */
        int hit_gate = -1;
        for (int g = (int)next_gate[e]; g < G; ++g) {                   /* gates below next are inactive */
            if (oc_check_collision(old_px, old_py, old_rot, n, gates + 4 * g, 1)) {
                hit_gate = g;
                break;                                                  /* :406-407 first active hit */
            }
        }
        if (hit_gate >= 0 && hit_gate == next_gate[e]) {                /* :726 */
            rw += 1.0;                                                  /* :727 */
            int64_t remaining = (int64_t)G - next_gate[e] - 1;          /* :728 remaining -= 1 */
            if (remaining == 0) {                                       /* :730 */
                rw += 10.0;                                             /* :732 */
                passed[e] += 1;                                         /* :733 */
                next_gate[e] = 0;                                       /* :734-737 */
            } else {
                passed[e] += 1;                                         /* :740 */
                next_gate[e] += 1;                                      /* :741 */
            }
        }

        /* Car.update :444-469 */
        double nvx = vx[e] + ax, nvy = vy[e] + ay;                      /* :452 */
        if (oc_norm2(ax, ay) == 0) {                                    /* :454 */
            nvx *= 1 - 0.2;                                             /* :455 */
            nvy *= 1 - 0.2;
        }
        nvx = nvx < -10.0 ? -10.0 : (nvx > 10.0 ? 10.0 : nvx);          /* :457 np.clip per component */
        nvy = nvy < -10.0 ? -10.0 : (nvy > 10.0 ? 10.0 : nvy);
        const double npx = old_px + nvx, npy = old_py + nvy;            /* :459 */
        if (oc_check_collision(npx, npy, r, n, walls, S)) destroyed[e] = 1; /* :468-469 */
        time_step[e] += 1;                                              /* :745 */
        uint8_t term = 0, trunc = 0;
        if (destroyed[e]) {                                             /* :746-748 */
            term = 1;
            rw -= 3.0;
        } else if (time_step[e] >= 1000) {                              /* :749-750 */
            trunc = 1;
        }
        px[e] = npx; py[e] = npy; vx[e] = nvx; vy[e] = nvy; rot[e] = r;

        /* _get_obs :569-597 */
        float* o = obs + (size_t)e * D;
        o[0] = (float)(npx / 1280);
        o[1] = (float)(npy / 720);
        o[2] = (float)(nvx / 10.0);
        o[3] = (float)(nvy / 10.0);
        o[4] = (float)cos(oc_radians(r));
        o[5] = (float)sin(oc_radians(r));
        for (int i = 0; i < R; ++i) {                                   /* get_distances :360-374 */
            double ang = oc_radians(r + (double)(i * step));
            o[6 + i] = (float)(oc_get_distance(npx, npy, cos(ang), sin(ang), walls, S) / 1000.0);
        }
        reward[e] = rw;
        terminated[e] = term;
        truncated[e] = trunc;
    }
}

/*
 * CarEnv.reset (car_env.py:605-691) after the track has been loaded: counters :677-680,
 * Car.reset :410-414, set_destroyed(False) :683, Car.update(boundaries) :686 (velocity and
 * acceleration are zero, so friction applies to zeros and the position does not move; the
 * collision test may set `destroyed` if the start pose already touches a wall), obs :688.
 */
void oc_env_reset(const double* walls, int S, int n, double start_x, double start_y, double start_rot,
                  int64_t n_envs, double* px, double* py, double* vx, double* vy, double* rot,
                  int64_t* time_step, int64_t* next_gate, int64_t* passed, uint8_t* destroyed, float* obs) {
    const int step = 360 / n;
    const int R = oc_ray_count(n);
    const int D = 6 + R;
    for (int64_t e = 0; e < n_envs; ++e) {
        time_step[e] = 0; next_gate[e] = 0; passed[e] = 0;
        double nvx = 0.0 + 0.0, nvy = 0.0 + 0.0;
        nvx *= 1 - 0.2; nvy *= 1 - 0.2;
        const double npx = start_x + nvx, npy = start_y + nvy;
        px[e] = npx; py[e] = npy; vx[e] = nvx; vy[e] = nvy; rot[e] = start_rot;
        destroyed[e] = (uint8_t)oc_check_collision(npx, npy, start_rot, n, walls, S);
        if (obs) {
            float* o = obs + (size_t)e * D;
            o[0] = (float)(npx / 1280);
            o[1] = (float)(npy / 720);
            o[2] = (float)(nvx / 10.0);
            o[3] = (float)(nvy / 10.0);
            o[4] = (float)cos(oc_radians(start_rot));
            o[5] = (float)sin(oc_radians(start_rot));
            for (int i = 0; i < R; ++i) {
                double ang = oc_radians(start_rot + (double)(i * step));
                o[6 + i] = (float)(oc_get_distance(npx, npy, cos(ang), sin(ang), walls, S) / 1000.0);
            }
        }
    }
}

/*
 * The vector-env call train.py makes (train.py:185): per env `CarEnv.step`, then
 * TransformReward `r * reward_scaling` (train.py:65,68), and gymnasium 0.29.1
 * AsyncVectorEnv's same-step auto-reset -- if terminated or truncated the returned
 * observation is the one of `env.reset()`.  `final_obs` (may be NULL) receives the obs of
 * CarEnv.step itself for every env (gymnasium's info["final_observation"] on done steps).
 * NOT pinned by the reference (gymnasium is a third-party dependency, requirements.txt:5).
 */
void oc_vec_step(const double* walls, int S, const double* gates, int G, int n, double start_x, double start_y,
                 double start_rot, double reward_scale, int64_t n_envs, double* px, double* py, double* vx,
                 double* vy, double* rot, int64_t* time_step, int64_t* next_gate, int64_t* passed,
                 uint8_t* destroyed, const int64_t* action, float* obs, double* reward, uint8_t* terminated,
                 uint8_t* truncated, float* final_obs) {
    const int D = 6 + oc_ray_count(n);
    oc_env_step(walls, S, gates, G, n, n_envs, px, py, vx, vy, rot, time_step, next_gate, passed, destroyed,
                action, obs, reward, terminated, truncated);
    if (final_obs) memcpy(final_obs, obs, (size_t)n_envs * D * sizeof(float));
    for (int64_t e = 0; e < n_envs; ++e) {
        reward[e] = reward[e] * reward_scale;
        if (terminated[e] || truncated[e])
            oc_env_reset(walls, S, n, start_x, start_y, start_rot, 1, px + e, py + e, vx + e, vy + e, rot + e,
                         time_step + e, next_gate + e, passed + e, destroyed + e, obs + (size_t)e * D);
    }
}

/*
 * Buffer.calculate_advantages (buffer.py:36-64) in float32, one rounding per torch op, in
 * torch's evaluation order:
 *   delta    = ((rew[t] + ((gamma * next_vals) * term_mask)) - val[t])                    :60
 *   last_gae = delta + ((((gamma*lambda) * term_mask) * trunc_mask) * last_gae)           :61
 * gamma and gamma*lambda are Python doubles that torch casts to float32 at the multiply.
 * Arrays are [T][N] row-major; last_* are [N]; adv, ret are [T][N].
 */
void oc_gae(const float* rew, const float* val, const float* term, const float* trunc, const float* last_val,
            const float* last_term, const float* last_trunc, double gamma, double lam, int64_t T, int64_t N,
            float* adv, float* ret) {
    const float g = (float)gamma;
    const float gl = (float)(gamma * lam);
    for (int64_t e = 0; e < N; ++e) {
        float last_gae = 0.0f;
        for (int64_t t = T - 1; t >= 0; --t) {
            const int last = (t == T - 1);
            const float next_vals = last ? last_val[e] : val[(t + 1) * N + e];        /* :53 */
            const float term_mask = 1.0f - (last ? last_term[e] : term[(t + 1) * N + e]);   /* :54 */
            const float trunc_mask = 1.0f - (last ? last_trunc[e] : trunc[(t + 1) * N + e]); /* :55 */
            float tmp = g * next_vals;
            tmp = tmp * term_mask;
            float delta = rew[t * N + e] + tmp;
            delta = delta - val[t * N + e];                                           /* :60 */
            float c = gl * term_mask;
            c = c * trunc_mask;
            c = c * last_gae;
            last_gae = delta + c;                                                     /* :61 */
            adv[t * N + e] = last_gae;                                                /* :62 */
            ret[t * N + e] = last_gae + val[t * N + e];                               /* :63 */
        }
    }
}

/* ------------------------------------------------------------------------------------------------------------------
 * Multi-threaded driver for the CPU BASELINE (bench.py's cpu_baseline leg; not used by the parity tests): T vector-env steps
 * (train.py:185 -- oc_vec_step: CarEnv.step + TransformReward + same-step auto-reset) of n_envs envs with pre-generated actions
 * [T][n_envs], the envs dealt to n_threads POSIX threads in contiguous static ranges.  Envs never interact (car_env.py has no
 * env-env coupling), so a thread runs ALL T steps of its own range without meeting the others: no per-step barrier, no shared
 * cache line but the two at a range's ends.  Returns the sum of all scaled rewards (a checksum; the final state is left in the
 * SoA arrays).  Same arithmetic as the single-threaded calls: a thread's range is stepped by oc_vec_step itself.
 * ------------------------------------------------------------------------------------------------------------------ */
#include <pthread.h>
#include <stdlib.h>

typedef struct {
    const double *walls, *gates;
    int S, G, n;
    double start_x, start_y, start_rot, reward_scale;
    int64_t lo, hi, n_envs, T;
    double *px, *py, *vx, *vy, *rot;
    int64_t *time_step, *next_gate, *passed;
    uint8_t* destroyed;
    const int64_t* actions;
    double reward_sum;
} oc_mt_job;

static void* oc_mt_worker(void* arg) {
    oc_mt_job* j = (oc_mt_job*)arg;
    const int64_t m = j->hi - j->lo;
    const int D = 6 + oc_ray_count(j->n);
    float* obs = (float*)malloc((size_t)m * D * sizeof(float));
    double* rew = (double*)malloc((size_t)m * sizeof(double));
    uint8_t* te = (uint8_t*)malloc((size_t)m);
    uint8_t* tr = (uint8_t*)malloc((size_t)m);
    double sum = 0.0;
    if (obs && rew && te && tr) {
        for (int64_t t = 0; t < j->T; ++t) {
            oc_vec_step(j->walls, j->S, j->gates, j->G, j->n, j->start_x, j->start_y, j->start_rot, j->reward_scale, m, j->px + j->lo,
                        j->py + j->lo, j->vx + j->lo, j->vy + j->lo, j->rot + j->lo, j->time_step + j->lo, j->next_gate + j->lo,
                        j->passed + j->lo, j->destroyed + j->lo, j->actions + t * j->n_envs + j->lo, obs, rew, te, tr, NULL);
            for (int64_t e = 0; e < m; ++e) sum += rew[e];
        }
    }
    free(obs); free(rew); free(te); free(tr);
    j->reward_sum = sum;
    return NULL;
}

double oc_vec_rollout_mt(const double* walls, int S, const double* gates, int G, int n, double start_x, double start_y,
                         double start_rot, double reward_scale, int64_t n_envs, int64_t T, int n_threads, double* px, double* py,
                         double* vx, double* vy, double* rot, int64_t* time_step, int64_t* next_gate, int64_t* passed,
                         uint8_t* destroyed, const int64_t* actions) {
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 1024) n_threads = 1024;
    if ((int64_t)n_threads > n_envs) n_threads = (int)n_envs;
    oc_mt_job* jobs = (oc_mt_job*)calloc((size_t)n_threads, sizeof(oc_mt_job));
    pthread_t* th = (pthread_t*)calloc((size_t)n_threads, sizeof(pthread_t));
    double total = 0.0;
    if (!jobs || !th) { free(jobs); free(th); return 0.0; }
    for (int k = 0; k < n_threads; ++k) {
        oc_mt_job* j = &jobs[k];
        j->walls = walls; j->gates = gates; j->S = S; j->G = G; j->n = n;
        j->start_x = start_x; j->start_y = start_y; j->start_rot = start_rot; j->reward_scale = reward_scale;
        j->lo = n_envs * k / n_threads; j->hi = n_envs * (k + 1) / n_threads; j->n_envs = n_envs; j->T = T;
        j->px = px; j->py = py; j->vx = vx; j->vy = vy; j->rot = rot;
        j->time_step = time_step; j->next_gate = next_gate; j->passed = passed; j->destroyed = destroyed; j->actions = actions;
        if (k == 0 || pthread_create(&th[k], NULL, oc_mt_worker, j) != 0) { if (k) oc_mt_worker(j); }
    }
    oc_mt_worker(&jobs[0]);      /* the calling thread takes the first range */
    for (int k = 1; k < n_threads; ++k) if (th[k]) pthread_join(th[k], NULL);
    for (int k = 0; k < n_threads; ++k) total += jobs[k].reward_sum;
    free(jobs); free(th);
    return total;
}
