"""Pins the CPU oracle (oracle/carenv_oracle.c) to the golden vectors recorded from the
unmodified reference (tests/golden/make_golden.py).  Everything here is BIT-EXACT: float64
state, float32 observations, float64 rewards, flags and counters."""
import numpy as np
import pytest

import oracle
from conftest import ENV_CONFIGS, GOLDEN, TRACKS

STATE = ("px", "py", "vx", "vy", "rot", "time_step", "next_gate", "passed")


def _load(track, n):
    return np.load(f"{GOLDEN}/env_{track}_n{n}.npz")


@pytest.mark.parametrize("track,n", ENV_CONFIGS)
def test_track_loader_matches_reference_geometry(track, n):
    g = _load(track, n)
    tr = oracle.Track(TRACKS[track])
    assert np.array_equal(tr.walls, g["walls"])      # outer-then-inner order, x*1280 / y*720
    assert np.array_equal(tr.gates, g["gates"])
    assert (tr.S, tr.G) == {"big_track": (24, 55), "track": (16, 45), "oval64": (128, 40)}[track]
    assert np.array_equal(np.array([tr.start_x, tr.start_y, 0.0, 0.0, tr.start_rot]), g["reset_state"])


@pytest.mark.parametrize("n,R", [(12, 12), (16, 17), (32, 33), (4, 4), (8, 8), (24, 24), (36, 36)])
def test_ray_count_quirk(n, R):
    # len(range(0, 360, 360 // n)) -- car_env.py:269; "16 rays" are 17, "32 rays" are 33
    assert oracle.ray_count(n) == len(range(0, 360, 360 // n)) == R


@pytest.mark.parametrize("track,n", ENV_CONFIGS)
def test_reset_obs(track, n):
    g = _load(track, n)
    env = oracle.OracleVecEnv(oracle.Track(TRACKS[track]), 3, num_rays=n)
    obs = env.reset()
    assert obs.dtype == np.float32 and obs.shape == (3, len(g["reset_obs"]))
    for i in range(3):
        assert np.array_equal(obs[i], g["reset_obs"])
    assert not env.destroyed.any()


@pytest.mark.parametrize("track,n", ENV_CONFIGS)
@pytest.mark.parametrize("grp", ["long", "short"])
def test_teacher_forced_step(track, n, grp):
    """Every recorded transition: inject the reference pre-state, apply the action, compare all."""
    g = _load(track, n)
    T, N = g[f"{grp}_action"].shape
    env = oracle.OracleVecEnv(oracle.Track(TRACKS[track]), T * N, num_rays=n)
    env.set_state(destroyed=0, **{k: g[f"{grp}_pre_{k}"].reshape(-1) for k in STATE})
    obs, rew, term, trunc = env.raw_step(g[f"{grp}_action"].reshape(-1))
    assert np.array_equal(obs, g[f"{grp}_step_obs"].reshape(T * N, -1))
    assert np.array_equal(rew, g[f"{grp}_reward"].reshape(-1))
    assert np.array_equal(term, g[f"{grp}_terminated"].reshape(-1))
    assert np.array_equal(trunc, g[f"{grp}_truncated"].reshape(-1))
    for k in STATE:
        assert np.array_equal(getattr(env, k), g[f"{grp}_post_{k}"].reshape(-1)), k


@pytest.mark.parametrize("track,n", ENV_CONFIGS)
@pytest.mark.parametrize("grp", ["long", "short"])
def test_free_running_vector_env(track, n, grp):
    """Replay the recorded action streams from reset through the vector-env call
    (auto-reset + reward scaling): the oracle must reproduce the whole trajectory."""
    g = _load(track, n)
    act = g[f"{grp}_action"]
    T, N = act.shape
    env = oracle.OracleVecEnv(oracle.Track(TRACKS[track]), N, num_rays=n, reward_scaling=float(g["reward_scaling"]))
    obs = env.reset()
    n_done = 0
    for t in range(T):
        for k in STATE:
            assert np.array_equal(getattr(env, k), g[f"{grp}_pre_{k}"][t]), (t, k)
        obs, rew, term, trunc, fin = env.step(act[t], want_final_obs=True)
        assert np.array_equal(obs, g[f"{grp}_ret_obs"][t]), t
        assert np.array_equal(fin, g[f"{grp}_step_obs"][t]), t
        assert np.array_equal(rew, g[f"{grp}_reward_scaled"][t]), t
        assert np.array_equal(term, g[f"{grp}_terminated"][t]), t
        assert np.array_equal(trunc, g[f"{grp}_truncated"][t]), t
        n_done += int(term.sum() + trunc.sum())
    assert n_done > 0


@pytest.mark.parametrize("track,n", ENV_CONFIGS)
def test_golden_coverage(track, n):
    """The fixtures exercise what the reference can do: crash, truncation, gate, full lap,
    every action, unreduced heading drift."""
    g = _load(track, n)
    rew = np.concatenate([g["long_reward"].reshape(-1), g["short_reward"].reshape(-1)])
    assert g["long_truncated"].sum() >= 2 and g["short_terminated"].sum() > 10
    assert (rew > 10.5).sum() >= 1                      # lap completion: +1 +10 (+0.01)
    assert ((rew > 0.5) & (rew < 2)).sum() > 50         # ordinary gate
    assert (rew < -2.5).sum() > 10                      # collision -3
    assert set(np.unique(g["short_action"])) == set(range(9))
    assert np.abs(g["long_pre_rot"]).max() > 4000       # ~1000 turns of 5 degrees, never wrapped
    # known-answer constants (README.md:90-93 / car_env.py:700,727,732,748)
    vals = np.unique(np.round(rew, 6))
    assert set(vals) <= {-3.0, -2.99, -2.0, -1.99, 0.0, 0.01, 1.0, 1.01, 11.0, 11.01, 8.0, 8.01}


def test_ray_segment_cases():
    rc = np.load(f"{GOLDEN}/ray_cases.npz")
    m = len(rc["px"])
    d = np.array([oracle.ray_distance(rc["px"][i], rc["py"][i], rc["angle"][i],
                                      [rc["x1"][i], rc["y1"][i], rc["x2"][i], rc["y2"][i]]) for i in range(m)])
    assert np.array_equal(d, rc["dist"])
    # the hand-written tail: parallel / collinear / behind / endpoint-exact / beyond 1000 -> 1000.0
    tail = rc["dist"][-12:]
    assert tail[0] == 1000.0 and tail[1] == 1000.0 and tail[2] == 1000.0
    assert tail[5] == 200.0 and tail[6] == 1000.0 and tail[7] == 1000.0 and tail[8] == 999.5


def test_norm_mode_switch_is_the_only_unpinned_degree_of_freedom():
    """np.linalg.norm's fused ddot tail is what this machine's reference run does (mode 1).
    Mode 0 (two roundings) differs from it by <= 1 ulp and only on a few cases."""
    rc = np.load(f"{GOLDEN}/ray_cases.npz")
    try:
        oracle.lib().oc_set_norm_mode(0)
        d0 = np.array([oracle.ray_distance(rc["px"][i], rc["py"][i], rc["angle"][i],
                                           [rc["x1"][i], rc["y1"][i], rc["x2"][i], rc["y2"][i]]) for i in range(len(rc["px"]))])
    finally:
        oracle.lib().oc_set_norm_mode(1)
    assert np.max(np.abs(d0 - rc["dist"]) / rc["dist"]) < 2.3e-16


def test_gae_matches_reference_buffer():
    g = np.load(f"{GOLDEN}/gae_cases.npz")
    for c in range(int(g["n_cases"])):
        adv, ret = oracle.gae(g[f"c{c}_rew"], g[f"c{c}_val"], g[f"c{c}_term"], g[f"c{c}_trunc"], g[f"c{c}_last_val"],
                              g[f"c{c}_last_term"], g[f"c{c}_last_trunc"], float(g["gamma"]), float(g["lam"]))
        assert np.array_equal(adv, g[f"c{c}_adv"]), c
        assert np.array_equal(ret, g[f"c{c}_ret"]), c
