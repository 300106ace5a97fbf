// env_steps.hpp -- part of the single translation unit ppocar.hip (included there after rollout.hpp; not a stand-alone header).
// K1f env_steps_fast_kernel: T successive CarEnv.step transitions (car_env.py:693-760, + TransformReward + gymnasium 0.29.1's same-step
// auto-reset) of every env as ONE launch, the actions read from the caller's [T, N] rows: pc_env_step (T = 1) and pc_env_step_many
// (SURVEY 8(d)'s level (i): the env alone under pre-generated actions).
#pragma once

// ------------------------------------------------------------------------------------------
// The table-driven env step of the persistent rollout kernel (env_step_fast, rollout.hpp) without a policy in front of it: a
// workgroup of 8 waves owns `epw` envs (256; 128: waves 4..7 only help to stage), a WAVE owns 32 of them with 2 lanes per env;
// the track's gather tables are staged in LDS once per launch (32 KB; with TAB also its 1/den table), the env state lives in
// registers from the first step to the last, every step's observation rows leave through the wave's 32 dense LDS rows as 16-byte
// stores.  After staging no barrier.  RPL: ray slots per lane (6 / 9 / 17 = 12 / 17 / 33 rays), SWP: the selector sweep (7: the
// chain-packed one for two loops of 13 vertices, big_track.json; 0: the generic ones), LIT: PC_DTYPE_F64 handles (the literal form).
// Every value is env_step_fast's, i.e. bit for bit what env_step_kernel computes (tests/test_env_gpu.py, test_env_steps_gpu.py);
// an action outside 0..7 is the no-op (car_env.py:721), as there.
// ------------------------------------------------------------------------------------------
// TWO: a batch of two tracks interleaved env by env with every aligned block of 64 envs split evenly between them (the host's check; i & 1
// is): both tracks' tables are staged (track 1's block `ts_floats` behind track 0's), the block's two waves de-interleave it -- wave 2k
// steps its track-0 envs, wave 2k + 1 its track-1 envs -- exactly as rollout_kernel's mode 7; rows leave by row.
template <int RPL, int SWP, bool TAB, bool LIT, bool TWO = false>
__global__ __launch_bounds__(512) void env_steps_fast_kernel(const EnvParams<float> p, const int64_t* __restrict__ actions, const int T,
                                                             const double reward_scale, float* __restrict__ obs,
                                                             float* __restrict__ reward, float* __restrict__ term_out,
                                                             float* __restrict__ trunc_out, const int epw, const int vec_ok,
                                                             int32_t* __restrict__ gates_passed, float* __restrict__ final_obs,
                                                             const int ts_floats) {
    constexpr int DC = RPL == 6 ? 18 : (RPL == 9 ? 23 : 39);
    static_assert(RPL == 6 || RPL == 9 || RPL == 17, "12 / 17 / 33 rays on two lanes per env");
    static_assert(SWP == 0 || ((SWP == 7 || SWP == 5) && RPL == 9), "the chain-packed sweeps: 17 rays");
    static_assert(!TWO || SWP == 5, "two tracks: each two equal loops of 13 or 9 vertices");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* sObs = lds;                     // [256 envs][DC]: the step's observation rows, dense (the caller's row layout)
    int* sMap = reinterpret_cast<int*>(sObs + 256 * DC);      // TWO: slot -> env of the block, per wave [8][32]
    float* sTab = sObs + 256 * DC + (TWO ? 256 : 0);          // the track's gather tables (ft_floats), then its 1/den table (TAB)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t N = p.N;
    // a mixed-track batch: the host checked that every workgroup's envs lie on ONE track, whose tables it stages
    const int trk_wg = (!TWO && p.track_id) ? __builtin_amdgcn_readfirstlane((int)p.track_id[min((int64_t)blockIdx.x * epw, N - 1)]) : 0;
    const TrackHdr h0 = cload(p.hdr + trk_wg);
    const FastTabs ft0 = stage_fast_tables<false, true, LIT>(p, h0, trk_wg, sTab, tid, 512);
    if constexpr (TWO) (void)stage_fast_tables<false, true, LIT>(p, cload(p.hdr + 1), 1, sTab + ts_floats, tid, 512);
    if constexpr (TAB) {
        const auto stage_rden = [&](const TrackHdr& h, float* sRden) {
            const float* src = p.rden + h.rden_off;
            const int nv = h.nV, n = 361 * nv;
            if constexpr (SWP == 7 || SWP == 5) {
                // wall_sweep_loops reads a row in the order (0, L, 1, L + 1, ...): entry 2 i = vertex i, entry 2 i + 1 = vertex L + i
                const int L = h.brk2;
                for (int i = tid; i < n; i += 512) {
                    const int row = i / nv, e = i - nv * row;
                    const int k = e < 2 * L ? (e >> 1) + ((e & 1) ? L : 0) : e;
                    sRden[i] = src[nv * row + k];
                }
            } else {
                for (int i = tid; i < n / 4; i += 512) reinterpret_cast<f32x4*>(sRden)[i] = reinterpret_cast<const f32x4*>(src)[i];   // nV is a multiple of 4
            }
        };
        stage_rden(h0, sTab + ft_floats(false, true));
        if constexpr (TWO) stage_rden(cload(p.hdr + 1), sTab + ts_floats + ft_floats(false, true));
    }
    // this wave's 32 envs: local rows [pbase, pbase + 32), 2 lanes per env
    const int pbase = wave * 32;
    const int el = pbase + (lane >> 1), g = lane & 1;
    const int64_t e_wave = (int64_t)blockIdx.x * epw + pbase;
    const int wtrk = TWO ? (wave & 1) : 0;                                      // TWO: the wave's track
    const int64_t e_block = (int64_t)blockIdx.x * epw + (wave >> 1) * 64;       // ... and its block of 64 envs
    int lj = lane >> 1;
    if constexpr (TWO) {
        const int64_t ei = e_block + lane;
        const bool is1 = ei < N && p.track_id[ei] != 0, is0 = ei < N && p.track_id[ei] == 0;
        uint64_t m = wtrk ? __builtin_amdgcn_ballot_w64(is1) : __builtin_amdgcn_ballot_w64(is0);
        for (int k = lane >> 1; k > 0; --k) m &= m - 1;      // slot s = the s-th env of the wave's track in the block (once per launch)
        lj = m ? __builtin_ctzll(m) : 64;
        if (g == 0) sMap[pbase + (lane >> 1)] = lj;
    }
    const int64_t e_env = TWO ? e_block + lj : e_wave + (lane >> 1);
    const bool e_valid = e_env < N && (!TWO || lj < 64);
    const TrackHdr hw = TWO ? cload(p.hdr + wtrk) : h0;
    const FastTabs ft = TWO ? ft_shift(ft0, ts_floats * wtrk) : ft0;
    using StateT = std::conditional_t<LIT, double, float>;
    const EnvParams<StateT> ps = p.template as<StateT>();
    EnvRegs st = {};
    if (e_valid) st = env_load<StateT>(ps, e_env);
    const FastLane fl = fast_lane<RPL, 2>(p, ft, g, sObs + el * DC);
    const int gq[2] = {(int)(size_t)ft.dir + 16 * g * p.q * p.step_deg,           // Car.get_passed_gate's rays j * (n // 4), j = g and g + 2,
                       (int)(size_t)ft.dir + 16 * (g + 2) * p.q * p.step_deg};    // as byte addresses into the direction table
    int k72 = Math<float>::mod72(st.k);
    f64x2 hcar = {1.0, 0.0};      // LIT: (cos, sin) of the env's current rotation (row st.k of the rotation table)
    if constexpr (LIT) {
        k72 = Math<float>::mod72((int)__builtin_rint((st.rot - hw.start_rot) / 5.0));
        if (e_valid) {
            const double2 e0 = p.dirtab64[hw.rot_off + st.k * (p.R + 2)];
            hcar = (f64x2){e0.x, e0.y};
        }
        st.rot = 0.0;     // (not kept: the rotation is the row's last entry, read again when the state is stored)
    }
    const lds_fp lrow = (lds_fp)(sObs + el * DC);
    int64_t a_next = e_valid ? actions[e_env] : 8;
    __syncthreads();      // the tables are in place; from here on the waves never synchronise again
    if (pbase >= epw) return;
    asm volatile("" : "+v"(st.px), "+v"(st.py), "+v"(st.vx), "+v"(st.vy), "+v"(st.k), "+v"(st.time), "+v"(st.next), "+v"(st.passed), "+v"(k72));
    if constexpr (LIT) asm volatile("" : "+v"(hcar));
    float* pa_rew = reward + e_env;
    float* pa_term = term_out + e_env;
    float* pa_trunc = trunc_out + e_env;
    const int64_t* pa_act = actions + e_env;
    float* dstg = obs + e_wave * DC;       // the wave's 32 rows are contiguous in the caller's [N, D] rows
    const int64_t left = N - e_wave;       // valid envs from this wave's first on
    const int n_rows = left >= 32 ? 32 : (int)left;
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
        const int a = (a_next >= 0 && a_next < 8) ? (int)a_next : 8;
        pa_act += N;
        if (t + 1 < T && e_valid) a_next = *pa_act;        // the next step's action row, under this step's arithmetic
        float rw, tf, cf;
        const bool done = env_step_fast<RPL, TAB, 1, 1, SWP, true, LIT, true>(p, hw, ft, fl, gq, g, st, k72, a, reward_scale, lrow, rw, tf, cf, t, lane, wave,
                                                                      0, nullptr, true, nullptr, &hcar);
        // pc_env_step's optional outputs (T = 1 only: the host passes them to no other launch): the finished episode's gate count
        // (info["gates_passed"] of CarEnv.step itself) and the observation CarEnv.step returned, before the same-step auto-reset
        if (gates_passed != nullptr && g == 0 && e_valid) gates_passed[e_env] = st.passed;
        if (final_obs != nullptr) {      // (uniform)
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const float* sl = sObs + pbase * DC;
            if constexpr (TWO) {
                float* fo = final_obs + e_block * DC;
                for (int i = lane; i < 32 * DC; i += 64) {
                    const int r = i / DC, c = i - r * DC, lr = sMap[pbase + r];
                    if (lr < 64 && e_block + lr < N) fo[lr * DC + c] = sl[i];
                }
            } else {
                float* fo = final_obs + e_wave * DC;
                for (int i = lane; i < n_rows * DC; i += 64) fo[i] = sl[i];
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (__builtin_amdgcn_ballot_w64(done) != 0) {   // wave-uniform: a finished env gets its reset observation and CarEnv.reset's state
            if (done) {
                float ro[(DC + 1) / 2];
#pragma unroll
                for (int j = 0; j < (DC + 1) / 2; ++j) ro[j] = ft.reset[g + 2 * j];     // (the table has 40 slots: in bounds)
#pragma unroll
                for (int j = 0; j < (DC + 1) / 2; ++j)
                    if (g + 2 * j < DC) lrow[g + 2 * j] = ro[j];
                env_reset_fast(hw, st, k72);
                if constexpr (LIT) {
                    const double2 e0 = p.dirtab64[hw.rot_off];      // row 0 = start_rot
                    hcar = (f64x2){e0.x, e0.y};
                }
            }
        }
        if (g == 0 && e_valid) {
            *pa_rew = rw;
            *pa_term = tf;
            *pa_trunc = cf;
        }
        pa_rew += N; pa_term += N; pa_trunc += N;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // the rows are complete (this wave wrote them all)
        __builtin_amdgcn_wave_barrier();
        const float* srcl = sObs + pbase * DC;
        if constexpr (TWO) {      // the wave's rows -> rows e_block + lj(slot) of the caller's buffer
            float* bg = obs + (int64_t)t * N * DC + e_block * DC;
#pragma unroll
            for (int j = 0; j < (32 * DC + 63) / 64; ++j) {
                const int i = lane + 64 * j;
                if (64 * j + 63 < 32 * DC || i < 32 * DC) {
                    const int r = i / DC, c = i - r * DC, lr = sMap[pbase + r];
                    if (lr < 64 && e_block + lr < N) bg[lr * DC + c] = srcl[i];
                }
            }
        } else if (vec_ok && n_rows == 32) {
            constexpr int NF4 = 8 * DC;                         // the wave's rows as float4s
#pragma unroll
            for (int j = 0; j < (NF4 + 63) / 64; ++j) {
                const int i = lane + 64 * j;
                if (64 * j + 63 < NF4 || i < NF4) reinterpret_cast<f32x4*>(dstg)[i] = reinterpret_cast<const f32x4*>(srcl)[i];
            }
        } else {
            for (int i = lane; i < n_rows * DC; i += 64) dstg[i] = srcl[i];
        }
        dstg += N * DC;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // (the next step overwrites the rows)
        __builtin_amdgcn_wave_barrier();
    }
    if (e_valid && g == 0) {
        if constexpr (LIT) st.rot = p.dirtab64[hw.rot_off + st.k * (p.R + 2) + p.R + 1].x;
        env_store<StateT>(ps, e_env, st);
    }
}
