"""Property tests on the CPU oracle (hypothesis): the invariants SURVEY section 8(a) derives from the reference
source, checked over random action streams.  They guard the oracle itself -- the thing every GPU parity test
leans on -- beyond the golden trajectories."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

import oracle
from conftest import TRACKS


@pytest.fixture(scope="module")
def tracks():
    return {k: oracle.Track(v) for k, v in TRACKS.items()}


@settings(max_examples=25, deadline=None)
@given(seed=st.integers(0, 2**31 - 1), n=st.sampled_from([12, 16, 32]), name=st.sampled_from(["track", "big_track"]),
       p_fwd=st.floats(0.0, 0.8))
def test_invariants_over_random_rollouts(tracks, seed, n, name, p_fwd):
    rng = np.random.default_rng(seed)
    N, T = 24, 160
    tr = tracks[name]
    env = oracle.OracleVecEnv(tr, N, num_rays=n, reward_scaling=0.1)
    reset_obs = env.reset()
    assert np.all(reset_obs == reset_obs[0])                       # E2: the reset observation is a per-track constant
    prev_passed = env.passed.copy()
    prev_next = env.next_gate.copy()
    for t in range(T):
        a = rng.integers(0, 9, N)
        fw = rng.random(N) < p_fwd
        a[fw] = rng.choice([0, 4, 5], int(fw.sum()))
        obs, rew, term, trunc, fin = env.step(a, want_final_obs=True)
        done = term | trunc
        assert not (term & trunc).any()                            # Q7: terminated wins over truncated
        assert np.all(np.abs(fin[:, 2:4]) <= 1.0 + 1e-7)           # Q2: per-component speed clip, |v| <= 10 -> obs <= 1
        assert np.all(fin[:, 6:] <= 1.0 + 1e-7) and np.all(fin[:, 6:] >= 0.0)        # ray distances in [0, 1000]
        assert np.allclose(fin[:, 4] ** 2 + fin[:, 5] ** 2, 1.0, atol=1e-6)            # heading is a unit vector
        assert np.all(obs[done] == reset_obs[0]) and np.all(obs[~done] == fin[~done])  # same-step auto-reset
        # reward decomposition (car_env.py:700,727,732,748; TransformReward x0.1): forward bonus + gate + lap - crash
        base = np.where(np.isin(a, [0, 4, 5]), 0.01, 0.0)
        extra = np.round((rew / 0.1 - base - np.where(term, -3.0, 0.0)), 6)
        assert set(np.unique(extra)) <= {0.0, 1.0, 11.0}
        gate = extra > 0.5
        # E1: a gate fires iff next_gate advanced (mod G); passed counts them; counters reset on done
        nxt = np.where(done, 0, env.next_gate)
        assert np.all(env.time_step[done] == 0) and np.all(env.passed[done] == 0) and np.all(env.next_gate[done] == 0)
        live = ~done
        assert np.all((env.passed[live] - prev_passed[live]) == gate[live].astype(np.int64))
        assert np.all(env.next_gate[live] == (prev_next[live] + gate[live]) % tr.G)
        assert np.all((extra == 11.0)[live] == ((prev_next[live] == tr.G - 1) & gate[live]))
        assert np.all(env.next_gate >= 0) and np.all(env.next_gate < tr.G)
        prev_passed, prev_next = env.passed.copy(), env.next_gate.copy()


@settings(max_examples=40, deadline=None)
@given(px=st.floats(50, 1200), py=st.floats(50, 650), ang=st.floats(-720, 720), d=st.floats(1.0, 900.0))
def test_ray_distance_to_a_perpendicular_wall_is_the_offset(px, py, ang, d):
    """A wall perpendicular to the ray at distance d, long enough to be hit: Ray.get_distance returns d."""
    c, s = np.cos(np.radians(ang)), np.sin(np.radians(ang))
    hx, hy = px + d * c, py + d * s
    seg = [hx - 500 * s, hy + 500 * c, hx + 500 * s, hy - 500 * c]
    got = oracle.ray_distance(px, py, ang, seg)
    assert got == pytest.approx(d, rel=1e-9, abs=1e-9)
    assert oracle.ray_distance(px, py, ang + 180.0, seg) == 1000.0        # behind the ray: no hit


@settings(max_examples=20, deadline=None)
@given(seed=st.integers(0, 10**6), T=st.integers(1, 40), N=st.integers(1, 7))
def test_gae_matches_the_closed_form(seed, T, N):
    """adv_t = sum_k (gamma*lam)^k * prod(masks) * delta_{t+k} evaluated in float64 agrees with the float32 scan."""
    rng = np.random.default_rng(seed)
    rew, val = rng.standard_normal((T, N)).astype(np.float32), rng.standard_normal((T, N)).astype(np.float32)
    term = (rng.random((T, N)) < 0.2).astype(np.float32)
    trunc = ((rng.random((T, N)) < 0.1) * (1 - term)).astype(np.float32)
    lv, lt, ltr = rng.standard_normal(N).astype(np.float32), (rng.random(N) < 0.3).astype(np.float32), np.zeros(N, np.float32)
    adv, ret = oracle.gae(rew, val, term, trunc, lv, lt, ltr, 0.99, 0.95)
    ref = np.zeros((T, N))
    last = np.zeros(N)
    for t in reversed(range(T)):
        nv = lv if t == T - 1 else val[t + 1]
        tm = 1 - (lt if t == T - 1 else term[t + 1])
        trm = 1 - (ltr if t == T - 1 else trunc[t + 1])
        delta = rew[t].astype(np.float64) + 0.99 * nv * tm - val[t]
        last = delta + 0.99 * 0.95 * tm * trm * last
        ref[t] = last
    assert np.allclose(adv, ref, atol=2e-5) and np.allclose(ret, ref + val, atol=2e-5)
