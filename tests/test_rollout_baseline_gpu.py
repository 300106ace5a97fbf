"""GPU parity tests of the kernel the benchmark times -- the persistent rollout kernel (K9 / K9s, pc_rollout) in its
DEFAULT dispatch -- at the single-GPU BASELINE.json shapes (reference loop: train.py:173-195):

  (a) 65536 envs x big_track x 16 -> 17 rays x 128 steps   (the target size / the per-GPU shard of configs[3])
  (b) 65536 envs x 32 -> 33 rays x 128 steps               (configs[2])
  (c) 4096 envs x 16 rays x 1024 steps                     (configs[1])

For each: (i) every buffer of the persistent launch equals the per-step kernels' bit for bit, and (ii) the stored
actions of a strided population of envs (every 32-env wave of the launch represented) are replayed through the float64 CPU
oracle: rewards / flags must be exact up to an env's first near-tie (|d - 10 px| <= 1e-9 px on a collision ray or the gate
ray test) and observations within one float32 ulp, >= 99.99 % of them bit-equal.

Also here: the reference's ray / segment unit cases (tests/golden/ray_cases.npz: parallel, endpoint-exact, behind,
beyond 1000 px; car_env.py:155-184) pushed through the HIP env kernels themselves -- one single-wall track per case.
"""
import numpy as np
import pytest
import torch

import oracle
import ppo_car_amd as pc
from ppo_car_amd._capi import lib
from ppo_car_amd.ppo import PPOConfig, Trainer
from conftest import GOLDEN, TRACKS

pytestmark = pytest.mark.gpu

OBS_TOL = 1.2e-7        # one float32 ulp just below 1.0 (north_star: fp32 observations within 1e-5)
MARGIN_PX = 1e-9        # only below this margin may a threshold test fall on the other side than the reference's
BIT_EQUAL = 0.9999      # fraction of observation entries that must be the oracle's very bits
REPLAY = 512            # envs replayed through the oracle


def _snap(tr):
    b = tr.buffer
    return [t.clone() for t in (b.obs_buf, b.act_buf, b.rew_buf, b.val_buf, b.logprob_buf, b.term_buf, b.trunc_buf,
                                tr.next_obs, tr.next_term, tr.next_trunc)]


def _collision_rays(n):
    return list(range(0, n, n // 4))            # Car.check_collision: range(0, n, n // 4) (car_env.py:389)


def _near_tie(track, n, st, final_obs, e):
    """Smallest threshold margin |d - 10| (px) of env e's step: the 4 collision rays against the walls (from the
    oracle's own pre-reset observation of the step) and against gate[next] at the pre-step pose (Car.get_passed_gate
    uses the rays of the previous update, car_env.py:394-408)."""
    step_deg = 360 // n
    walls = [abs(float(final_obs[e, 6 + r]) * 1000.0 - 10.0) for r in _collision_rays(n)]
    gate = track.gates[int(st["next_gate"][e])]
    gates = [abs(oracle.ray_distance(st["px"][e], st["py"][e], st["rot"][e] + r * step_deg, gate) - 10.0)
             for r in _collision_rays(n)]
    return min(walls + gates)


def strided_population(n_envs, per_wave=4, wave=32, limit=None):
    """Env indices that touch EVERY 32-env wave of the launch (hence every workgroup of 16 .. 256 envs, and both waves of a
    SIMD): `per_wave` envs out of each block of `wave`, at offsets that rotate from block to block so that both lanes pairs /
    all lane quads of the env-step mapping are covered."""
    blocks = np.arange(0, n_envs, wave)
    offs = (np.arange(per_wave)[None, :] * (wave // per_wave) + (blocks[:, None] // wave) % (wave // per_wave))
    sel = (blocks[:, None] + offs).reshape(-1)
    sel = sel[sel < n_envs]
    if limit is not None and len(sel) > limit:
        sel = sel[np.linspace(0, len(sel) - 1, limit).astype(np.int64)]
    return np.unique(sel)


def _oracle_replay_check(cfg, snaps, first_obs, what, sel=None, stats=None, bit_equal=BIT_EQUAL, init_state=None, exact=False,
                         min_alive=0.97):
    """Replay the stored actions of the envs `sel` (default: the first REPLAY) through the oracle and compare with the buffers
    of the persistent rollout.  `stats` (a dict) receives the observation-error histogram, the list of departures and the counts of
    the bookkeeping events the replay went through (car_env.py:726-750: gates, laps, truncations, terminated at time_step >= 1000).
    init_state: the rollout did not start from reset -- a dict of full-batch state arrays (pc_env_set_state's fields) the oracle's
    envs start from (the launch's first policy input is then whatever next_obs held: not compared).
    exact: the bit-exact dtype -- every observation, reward and flag of every env must be the oracle's bits, no departures."""
    obs_buf, act_buf, rew_buf, _val, _lp, term_buf, trunc_buf, next_obs, next_term, next_trunc = snaps[:10]
    T, n = cfg.n_steps, cfg.num_rays
    if sel is None:
        sel = np.arange(min(REPLAY, obs_buf.shape[1]))
    P = len(sel)
    idx = torch.as_tensor(sel, device=obs_buf.device)
    acts = act_buf[:, idx].cpu().numpy().astype(np.int64)
    assert acts.min() >= 0 and acts.max() <= 8
    track = oracle.Track(cfg.track)
    ora = oracle.OracleVecEnv(track, P, num_rays=n, reward_scaling=cfg.reward_scaling, threads=8)
    o = ora.reset()
    if init_state is None:
        assert np.abs(first_obs[idx].cpu().numpy() - o).max() <= (0.0 if exact else 1e-6)
    else:
        ora.set_state(**{k: np.asarray(v)[sel] for k, v in init_state.items()})
    alive = np.ones(P, bool)                 # env has not yet left the oracle's trajectory
    worst, ties, n_done, n_cmp, n_eq = 0.0, 0, 0, 0, 0
    ev = dict(gates=0, laps=0, truncations=0, terminated_at_time_limit=0, max_abs_turns=0)
    edges = np.array([0.0, 1e-9, 3e-8, 6e-8, 1.2e-7, 2.5e-7, 5e-7, 1e-6, 1e-5, 1e-4, np.inf])
    hist = np.zeros(len(edges) - 1, np.int64)
    departures = []
    for t in range(T):      # (row by row: the strided slices of the big buffers are gathered on the device)
        OBt = (obs_buf[t + 1] if t + 1 < T else next_obs)[idx].cpu().numpy()
        TEt = (term_buf[t + 1] if t + 1 < T else next_term)[idx].cpu().numpy() != 0
        TRt = (trunc_buf[t + 1] if t + 1 < T else next_trunc)[idx].cpu().numpy() != 0
        RWt = rew_buf[t][idx].cpu().numpy()
        st = {k: getattr(ora, k).copy() for k in ("px", "py", "rot", "next_gate", "time_step", "passed")}
        o, r, te, trn, fin = ora.step(acts[t], want_final_obs=True)
        # the bookkeeping events of this step, from the oracle's own counters (on the envs still compared)
        x = r.astype(np.float64) / cfg.reward_scaling                # the step's raw reward: 0.01 forward + 1 gate + 10 lap - 3 crash (car_env.py:700-748)
        lap = x > 5.0                                                # +10 survives even a crash's -3 in the same step
        gate = (x - 10.0 * lap + 3.0 * te) > 0.5
        assert not (lap & (st["next_gate"] != track.G - 1)).any()    # a lap is the pass of the LAST gate (:730-737)
        ev["gates"] += int((gate & alive).sum())
        ev["laps"] += int((lap & alive).sum())
        ev["truncations"] += int((trn & alive).sum())
        ev["terminated_at_time_limit"] += int((te & (st["time_step"] >= 999) & alive).sum())        # Q7: terminated wins (car_env.py:746-750)
        ev["max_abs_turns"] = max(ev["max_abs_turns"], int(np.abs(np.rint((st["rot"] - track.start_rot) / 5.0)).max()))
        ev_bad = (TEt != te) | (TRt != trn) | (RWt != r.astype(np.float32))
        if exact:
            assert not ev_bad.any(), f"{what}: step {t}: flags / rewards differ from the oracle's in the bit-exact dtype (envs {sel[np.nonzero(ev_bad)[0][:8]]})"
            assert np.array_equal(OBt, o), f"{what}: step {t}: observations differ from the oracle's in the bit-exact dtype"
        for e in np.nonzero(ev_bad & alive)[0]:
            m = _near_tie(track, n, st, fin, e)
            departures.append((int(sel[e]), t, float(m)))
            assert m <= MARGIN_PX, f"{what}: env {sel[e]} step {t}: event mismatch away from a threshold (margin {m} px)"
            ties += 1
        alive &= ~ev_bad
        if alive.any():
            err = np.abs(OBt[alive].astype(np.float64) - o[alive].astype(np.float64))
            worst = max(worst, float(err.max()))
            hist += np.histogram(err, bins=edges)[0]
            n_cmp += err.size
            n_eq += int((err == 0).sum())
        n_done += int((te | trn)[alive].sum())
    assert worst <= OBS_TOL, f"{what}: obs error {worst} before the first near-tie"
    assert alive.mean() > min_alive, f"{what}: {P - alive.sum()} of {P} envs left the oracle's trajectory"
    assert n_done > 0                       # episodes ended (auto-reset rows were compared)
    assert n_eq >= bit_equal * n_cmp, f"{what}: only {n_eq} of {n_cmp} observation entries bit-equal"
    if stats is not None:
        stats.update(events=ev)
        stats.update(envs=P, steps=T, entries=n_cmp, bit_equal=n_eq, obs_max_err=worst, hist_edges=[float(x) for x in edges[:-1]],
                     hist=[int(x) for x in hist], departures=departures, episodes_ended=n_done, on_trajectory=float(alive.mean()))
    return worst, ties, float(alive.mean())


@pytest.mark.parametrize("n_envs,num_rays,n_steps", [(65536, 16, 1024), (65536, 32, 128), (4096, 16, 1024)],
                         ids=["target_65536x1024x17rays", "cfg2_65536x128x33rays", "cfg1_4096x1024x17rays"])
def test_default_dispatch_rollout_vs_step_kernels_and_oracle(n_envs, num_rays, n_steps):
    res, first = {}, None
    assert lib.pc_build_ablate() == 0
    for mode in ("mega", "steps"):
        cfg = PPOConfig(n_envs=n_envs, n_steps=n_steps, num_rays=num_rays, track=TRACKS["big_track"], rollout_kernel=mode,
                        use_graphs=False, seed=11)
        tr = Trainer(cfg, device="cuda")
        if first is None:
            first = tr.next_obs.clone()
        tr.rollout()
        torch.cuda.synchronize()
        assert tr.rollout_mode == ("mega" if mode == "mega" else "steps-eager")     # default dispatch accepted the shape
        res[mode] = _snap(tr)
        res[mode + "_state"] = tr.envs.get_state()
        tr.close()
        del tr
    for i, (a, b) in enumerate(zip(res["mega"], res["steps"])):
        assert torch.equal(a, b), f"buffer {i} differs between pc_rollout and the per-step kernels"
    for k in res["mega_state"]:
        assert np.array_equal(res["mega_state"][k], res["steps_state"][k]), k
    sel = strided_population(n_envs, per_wave=1, limit=1024)      # (tools/population_replay.py: 4 per wave = 8192 envs, by hand)
    worst, ties, alive = _oracle_replay_check(cfg, res["mega"], first, f"N={n_envs} rays={num_rays} T={n_steps}", sel=sel)
    print(f"K9 vs oracle: obs max err {worst:.2e}, near-tie flips {ties}, envs on the oracle trajectory {alive:.3f}")
    del res
    torch.cuda.empty_cache()


@pytest.mark.parametrize("n_envs", [1000, 20000], ids=["small_form", "big_form"])
@pytest.mark.parametrize("num_rays", [17, 18])
def test_ray_counts_that_share_the_slot_count_of_17_rays(num_rays, n_envs):
    """num_rays 17 / 18 give 18 actual rays (range(0, 360, 360 // n), car_env.py:269): the same ray slots per lane as 16 -> 17
    rays, but an observation of 24 floats.  The fast kernels carry their observation width as a compile-time constant, so the
    dispatch must send these shapes to the generic mode: default dispatch, bitwise the per-step kernels, and the oracle."""
    assert oracle.ray_count(num_rays) == 18
    res, first = {}, None
    for mode in ("steps", "mega"):
        cfg = PPOConfig(n_envs=n_envs, n_steps=64, num_rays=num_rays, track=TRACKS["big_track"], rollout_kernel=mode,
                        use_graphs=False, seed=5)
        tr = Trainer(cfg, device="cuda")
        if first is None:
            first = tr.next_obs.clone()
        for _ in range(2):
            tr.rollout()
            tr.buffer.ptr = 0
        torch.cuda.synchronize()
        assert tr.rollout_mode == ("mega" if mode == "mega" else "steps-eager") and tr.obs_dim == (24,)
        res[mode] = _snap(tr)
        res[mode + "_state"] = tr.envs.get_state()
        tr.close()
    for i, (a, b) in enumerate(zip(res["steps"], res["mega"])):
        assert torch.equal(a, b), i
    for k in res["mega_state"]:
        assert np.array_equal(res["mega_state"][k], res["steps_state"][k]), k
    assert float(res["mega"][5].sum()) > 0
    # one rollout from reset against the oracle
    cfg = PPOConfig(n_envs=n_envs, n_steps=200, num_rays=num_rays, track=TRACKS["big_track"], rollout_kernel="mega", use_graphs=False, seed=5)
    tr = Trainer(cfg, device="cuda")
    tr.rollout()
    torch.cuda.synchronize()
    _oracle_replay_check(cfg, _snap(tr), first, f"rays={num_rays} N={n_envs}", sel=strided_population(n_envs, per_wave=4, limit=256))
    tr.close()


@pytest.mark.parametrize("fast", [1, 2, 0], ids=["chain28_kernels", "fast_generic_sweep", "generic_mode"])
@pytest.mark.parametrize("epw", [128, 256])
@pytest.mark.parametrize("n_envs", [1000, 512])
def test_both_workgroup_sizes_of_the_big_form_at_small_n(n_envs, epw, fast):
    """The 256-env-per-workgroup variant (what 65536 envs take) and the 128-env one, forced at a small batch: bitwise the
    per-step kernels, with the 1/den table in LDS (form 0) -- the configuration the benchmark runs -- in the kernels compiled
    for big_track's chain length (the default), in the fast mode's generic-sweep kernels and in the generic mode."""
    res = {}
    for mode in ("steps", "mega"):      # every option through this trainer's own handles (pc_policy_create, pc_env_set_option)
        tr = Trainer(PPOConfig(n_envs=n_envs, n_steps=96, num_rays=16, track=TRACKS["big_track"], rollout_kernel=mode,
                               use_graphs=False, seed=3, policy_split=0, rollout_form=0, rollout_epw=epw, rollout_fast=fast), device="cuda")
        assert (tr.envs.get_option("rollout_form"), tr.envs.get_option("rollout_epw"), tr.envs.get_option("rollout_fast")) == (0, epw, fast)
        with pytest.raises(pc.PpoCarError):
            tr.envs.set_option("rollout_fast", 4)
        for _ in range(2):
            tr.rollout()
            tr.buffer.ptr = 0
        torch.cuda.synchronize()
        assert tr.rollout_mode == ("mega" if mode == "mega" else "steps-eager")
        res[mode] = _snap(tr)
        tr.close()
    for i, (a, b) in enumerate(zip(res["steps"], res["mega"])):
        assert torch.equal(a, b), i
    assert float(res["mega"][5].sum()) > 0


@pytest.mark.parametrize("epw", [16, 32])
@pytest.mark.parametrize("n_envs,num_rays", [(1000, 16), (8000, 16), (500, 12)])
def test_both_workgroup_sizes_of_the_small_form(n_envs, num_rays, epw):
    """The small form (hidden tiles and sweep parts split over the eight waves of a workgroup) with 16 envs per workgroup --
    what BASELINE configs[1] (4096 envs) takes: one workgroup on every CU -- and with 32, forced at other batch sizes: bitwise
    the per-step kernels (split policy form), two rollouts so that auto-resets are inside the window."""
    res = {}
    for mode in ("steps", "mega"):
        tr = Trainer(PPOConfig(n_envs=n_envs, n_steps=96, num_rays=num_rays, track=TRACKS["big_track"], rollout_kernel=mode,
                               use_graphs=False, seed=4, policy_split=1, rollout_form=1, rollout_epw=epw), device="cuda")
        for _ in range(2):
            tr.rollout()
            tr.buffer.ptr = 0
        torch.cuda.synchronize()
        assert tr.rollout_mode == ("mega" if mode == "mega" else "steps-eager")
        res[mode] = _snap(tr)
        res[mode + "_state"] = tr.envs.get_state()
        tr.close()
    for i, (a, b) in enumerate(zip(res["steps"], res["mega"])):
        assert torch.equal(a, b), i
    for k in res["mega_state"]:
        assert np.array_equal(res["mega_state"][k], res["steps_state"][k]), k
    assert float(res["mega"][5].sum()) > 0


@pytest.mark.parametrize("n_envs,num_rays,n_steps", [(65536, 16, 64), (4096, 16, 200), (1000, 32, 50), (3000, 12, 33)])
def test_rollout_ex_delivers_bootstrap_values_and_reward_totals(n_envs, num_rays, n_steps):
    """pc_rollout with its two optional outputs = pc_rollout without them (same buffers, bit for bit) + the critic's value of the
    final observation (train.py:200) and every env's reward total (train.py:272's numerator), from inside the same launch -- big
    and small form."""
    outs = {}
    for ex in (True, False):
        cfg = PPOConfig(n_envs=n_envs, n_steps=n_steps, num_rays=num_rays, track=TRACKS["big_track"], rollout_kernel="mega",
                        use_graphs=False, seed=21)
        tr = Trainer(cfg, device="cuda")
        if ex:
            tr.rollout()
            assert tr.rollout_mode == "mega" and tr._aux_valid
            boot, rsum = tr._boot_val.clone(), tr._rew_sum.clone()
            with torch.no_grad():
                ref_v = tr.agent.get_value(tr.next_obs).reshape(-1)
            assert float((boot - ref_v).abs().max()) <= 1e-5                     # the fused policy step's arithmetic vs torch fp32
            ref_sum = tr.buffer.rew_buf.double().sum(0)
            assert float((rsum.double() - ref_sum).abs().max()) <= 1e-4 * max(1.0, float(ref_sum.abs().max()))
            mean_k = float(rsum.sum() / (n_steps * n_envs))
            assert mean_k == pytest.approx(float(tr.buffer.rew_buf.mean()), rel=1e-4, abs=1e-7)
        else:               # without the two optional outputs, called directly
            agent, buf = tr.agent, tr.buffer
            assert agent.pack_policy()
            buf.obs_buf[0].copy_(tr.next_obs)
            buf.term_buf[0].copy_(tr.next_term)
            buf.trunc_buf[0].copy_(tr.next_trunc)
            rc = lib.pc_rollout(tr.envs._h, agent._image_handle, agent._image.data_ptr(), cfg.n_steps, float(cfg.reward_scaling), int(agent.rng_seed),
                                0, tr.rng_base.data_ptr(), buf.obs_buf.data_ptr(), buf.act_buf.data_ptr(), buf.rew_buf.data_ptr(),
                                buf.val_buf.data_ptr(), buf.term_buf.data_ptr(), buf.trunc_buf.data_ptr(), buf.logprob_buf.data_ptr(),
                                tr.next_obs.data_ptr(), tr.next_term.data_ptr(), tr.next_trunc.data_ptr(), None, None,
                                torch.cuda.current_stream().cuda_stream)
            assert rc == 0
        torch.cuda.synchronize()
        outs[ex] = _snap(tr)
        tr.close()
    for i, (a, b) in enumerate(zip(outs[True], outs[False])):
        assert torch.equal(a, b), i


# ------------------------------------------------------------------------------------------------
# the reference's ray / segment unit cases through the HIP kernels
# ------------------------------------------------------------------------------------------------
def _ray_case_envs(rc, lo, hi, dtype):
    """One single-wall track per case: the car starts at the ray origin heading along the ray, so ray 0 of the reset
    observation (per-segment cast) and of a no-op step's observation (wall sweep) is the case's distance."""
    far_gate = np.array([[-5000.0, -5000.0, -5001.0, -5000.0]])
    tracks = [pc.Track(walls=[[rc["x1"][i], rc["y1"][i], rc["x2"][i], rc["y2"][i]]], gates=far_gate,
                       start=(rc["px"][i], rc["py"][i], rc["angle"][i])) for i in range(lo, hi)]
    return pc.VecCarEnv(hi - lo, tracks, num_rays=12, dtype=dtype, track_id=np.arange(hi - lo, dtype=np.uint8))


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_ray_segment_unit_cases_through_the_hip_kernels(dtype):
    rc = np.load(f"{GOLDEN}/ray_cases.npz")
    m = len(rc["px"])
    want = (rc["dist"] / 1000.0).astype(np.float32)          # obs entry 6 = d / 1000 (car_env.py:593)
    got_reset, got_step = np.zeros(m, np.float32), np.zeros(m, np.float32)
    for lo in range(0, m, 252):
        hi = min(lo + 252, m)
        env = _ray_case_envs(rc, lo, hi, dtype)
        obs, _ = env.reset()
        got_reset[lo:hi] = obs[:, 6].cpu().numpy()
        fin = torch.empty(hi - lo, env.obs_dim, device="cuda")
        env.step(torch.full((hi - lo,), 8, dtype=torch.int64, device="cuda"), final_obs=fin)   # no-op: the car stays put
        got_step[lo:hi] = fin[:, 6].cpu().numpy()
        env.close()
    # the hand-written tail (make_golden.ray_unit_cases): parallel x3, endpoint t == 0, t == 1, mid hit, behind, > 1000, 999.5,
    # origin on the segment end, d == 10 twice
    tail = slice(m - 12, m)
    if dtype == "f64":
        # float64 instantiation: the reference's own arithmetic.  Device cos/sin (ocml) differ from glibc by <= 1 ulp on a
        # few arguments, which can move a distance by an ulp of float64 -- invisible after the float32 cast except exactly
        # at a rounding boundary: require equality on >= 99.5 % and <= 1 float32 ulp elsewhere.
        for got in (got_reset, got_step):
            exact = got == want
            assert exact.mean() >= 0.995, exact.mean()
            assert np.all(np.abs(got - want) <= np.spacing(want))
            assert np.array_equal(got[tail], want[tail])
    else:
        # F32 mode = float32 selection + float64 refinement under the reference's strict test: the three hand-written EXACT
        # degeneracies -- a segment endpoint exactly on the ray line (t == 0, t == 1) and the ray origin exactly on the segment's
        # end -- give the reference's "no hit" like everything else; the distances come from float64 arithmetic (un / den instead of
        # the reference's norm of the hit point: a few float64 ulps apart), so the float32 entries are the reference's bits except
        # where that difference straddles a float32 rounding boundary.
        for got in (got_reset, got_step):
            assert np.all(np.abs(got - want) <= np.spacing(want)), np.abs(got - want).max()
            assert (got == want).mean() >= 0.995, (got == want).mean()
            assert np.array_equal(got[tail], want[tail])


def test_f32_ray_origin_on_a_wall_line_follows_the_reference():
    """u == 0 (the ray origin exactly on the wall, strictly between its endpoints): the reference's `u > 0` (car_env.py:178)
    rejects the hit, for either orientation of the wall.  Round 2's float32 sweep accepted u == +0.0 (a documented deviation);
    the float64 refinement applies the strict test, so F32 mode now answers as the reference and the F64 kernel do."""
    far_gate = np.array([[-5000.0, -5000.0, -5001.0, -5000.0]])
    out = {}
    for name, wall in (("down", [[100.0, 50.0, 100.0, 150.0]]), ("up", [[100.0, 150.0, 100.0, 50.0]])):   # through the ray origin
        assert oracle.ray_distance(100.0, 100.0, 0.0, wall[0]) == 1000.0
        for dtype in ("f64", "f32"):
            env = pc.VecCarEnv(1, pc.Track(walls=wall, gates=far_gate, start=(100.0, 100.0, 0.0)), num_rays=12, dtype=dtype)
            obs, _ = env.reset()
            fin = torch.empty(1, env.obs_dim, device="cuda")
            env.step(torch.full((1,), 8, dtype=torch.int64, device="cuda"), final_obs=fin)
            out[name, dtype] = (float(obs[0, 6]), float(fin[0, 6]))
            env.close()
    for key, v in out.items():
        assert v == (1.0, 1.0), (key, v)


def test_f32_ray_through_a_shared_corner_follows_the_reference():
    """A ray EXACTLY through the vertex two walls share: the reference's strict 0 < t < 1 (car_env.py:178) lets it pass between
    them and it hits whatever lies behind.  (Round 2's float32 sweep hit one of the two walls by design.)"""
    far_gate = np.array([[-5000.0, -5000.0, -5001.0, -5000.0]])
    walls = [[200.0, 50.0, 200.0, 100.0], [200.0, 100.0, 200.0, 150.0],      # two collinear walls meeting at (200, 100)
             [300.0, 50.0, 300.0, 150.0]]                                      # a wall 100 px behind them
    trk = pc.Track(walls=walls, gates=far_gate, start=(100.0, 100.0, 0.0))
    assert oracle.ray_distance(100.0, 100.0, 0.0, walls[0]) == 1000.0 and oracle.ray_distance(100.0, 100.0, 0.0, walls[1]) == 1000.0
    for dtype in ("f64", "f32"):
        env = pc.VecCarEnv(1, trk, num_rays=12, dtype=dtype)
        obs, _ = env.reset()
        fin = torch.empty(1, env.obs_dim, device="cuda")
        env.step(torch.full((1,), 8, dtype=torch.int64, device="cuda"), final_obs=fin)
        assert float(obs[0, 6]) == float(np.float32(0.2)) and float(fin[0, 6]) == float(np.float32(0.2)), (dtype, obs[0, 6], fin[0, 6])
        env.close()


@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("n_envs,n_steps,num_rays", [(1, 1, 16), (1, 40, 12), (33, 3, 16), (257, 2, 32), (8193, 1, 16), (32769, 2, 16), (65537, 3, 16)])
def test_ragged_and_minimal_shapes_through_the_persistent_kernels(n_envs, n_steps, num_rays, dtype):
    """The smallest and the most ragged launches: one env, one step, env counts one past a workgroup / wave / form boundary (33, 257, 8193 =
    the first batch of the 16-envs-per-wave form, 32769 and 65537 = a last workgroup with a single env), observation rows whose
    16-byte alignment the vector stores cannot assume.  pc_rollout with the trainer's 256-env threshold lifted: bitwise the per-step
    kernels, float64 state included, and the oracle on every env (up to 256) for all steps."""
    res, first = {}, None
    for mode in ("mega", "steps"):
        cfg = PPOConfig(n_envs=n_envs, n_steps=n_steps, num_rays=num_rays, track=TRACKS["big_track"], rollout_kernel=mode, use_graphs=False, seed=41,
                        env_dtype=dtype, policy_split=1 if (n_envs <= 8192 and not (dtype == "f64" and num_rays == 32)) else 0)
        tr = Trainer(cfg, device="cuda")
        if first is None:
            first = tr.next_obs.clone()
        for _ in range(2):
            tr.rollout()
            tr.buffer.ptr = 0
        torch.cuda.synchronize()
        assert tr.rollout_mode == ("mega" if mode == "mega" else "steps-eager"), tr.rollout_mode
        res[mode] = _snap(tr)
        res[mode + "_state"] = tr.envs.get_state()
        tr.close()
    for i, (a, b) in enumerate(zip(res["mega"], res["steps"])):
        assert torch.equal(a, b), f"buffer {i} differs (N={n_envs}, T={n_steps}, rays={num_rays}, {dtype})"
    for k in res["mega_state"]:
        assert np.array_equal(res["mega_state"][k], res["steps_state"][k]), k
