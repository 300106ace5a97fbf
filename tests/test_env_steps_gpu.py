"""K1f (env_steps_fast_kernel): pc_env_step's table-driven form and pc_env_step_many -- CarEnv.step (car_env.py:693-760) as the
persistent rollout kernel computes it (env_step_fast: gather tables in LDS, two lanes per env, the chain-packed / unrolled selector
sweep, float64 refinement or the literal cast), with the actions taken from the caller instead of a policy pass.

Bars: every output and the env state afterwards equal the generic per-step kernel K1's bit for bit (K1 is pinned to the reference's
golden vectors in test_env_gpu.py) -- F32 and F64 handles, 12 / 16 / 32 nominal rays, both reference tracks, ragged batch sizes,
out-of-range actions, mixed tracks in blocks; F64 handles reproduce the golden vectors themselves; the actions of a trained policy's
2048-step rollout (laps, time limits) replayed through pc_env_step_many reproduce that rollout's buffers."""
import numpy as np
import pytest
import torch

import ppo_car_amd as pc
from conftest import GOLDEN, TRACKS

pytestmark = pytest.mark.gpu

STATE = ("px", "py", "vx", "vy", "rot", "time_step", "next_gate", "passed")
FAST_CONFIGS = [(t, n) for t in ("big_track", "track") for n in (12, 16, 32)]


def _env(N, track, n, dtype, form, track_id=None, rs=0.1):
    e = pc.VecCarEnv(N, track, num_rays=n, reward_scaling=rs, dtype=dtype, track_id=track_id)
    e.set_option("step_form", form)
    return e


def _actions(T, N, seed, wild=True):
    g = torch.Generator().manual_seed(seed)
    a = torch.randint(0, 9, (T, N), generator=g)
    fwd = torch.rand(T, N, generator=g) < 0.35           # biased forward: cars reach gates as well as walls
    a = torch.where(fwd, torch.zeros_like(a), a)
    if wild:                                             # anything outside 0..7 is the no-op (car_env.py:721)
        w = torch.rand(T, N, generator=g) < 0.01
        a = torch.where(w, torch.randint(-3, 200, (T, N), generator=g), a)
    return a.cuda()


@pytest.mark.parametrize("track,n", FAST_CONFIGS)
@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("N", [1, 33, 257, 4099])
def test_table_driven_step_is_bitwise_the_generic_step(track, n, dtype, N):
    T = 160
    acts = _actions(T, N, 1000 * n + N)
    a, b = _env(N, TRACKS[track], n, dtype, 1), _env(N, TRACKS[track], n, dtype, 2)
    oa, _ = a.reset()
    ob, _ = b.reset()
    assert torch.equal(oa, ob)
    n_done = 0
    for t in range(T):
        ra, rb = a.step(acts[t]), b.step(acts[t])
        for x, y, what in zip(ra[:4], rb[:4], ("obs", "reward", "terminated", "truncated")):
            assert torch.equal(x, y), (t, what, int((x != y).sum()))
        n_done += int(ra[2].sum())
    assert a.last_step_kernel() == "K1" and b.last_step_kernel() == "K1f"
    sa, sb = a.get_state(), b.get_state()
    for k in STATE:
        assert np.array_equal(sa[k], sb[k]), k
    assert n_done > 0 or N == 1        # episodes ended and restarted inside the run
    a.close()
    b.close()


@pytest.mark.parametrize("track,n", FAST_CONFIGS)
@pytest.mark.parametrize("grp", ["long", "short"])
def test_table_driven_step_reproduces_the_reference_goldens_f64(track, n, grp):
    """The reference's own transitions (tests/golden/env_*.npz, recorded from car_env.py), teacher-forced through K1f on an F64 handle:
    returned observation, scaled reward, flags and the float64 state afterwards, bit for bit."""
    g = np.load(f"{GOLDEN}/env_{track}_n{n}.npz")
    T, N = g[f"{grp}_action"].shape
    M = T * N
    env = _env(M, TRACKS[track], n, "f64", 2, rs=float(g["reward_scaling"]))
    env.reset()
    env.set_state(**{k: g[f"{grp}_pre_{k}"].reshape(-1) for k in STATE})
    obs, rew, term, trunc, _ = env.step(torch.from_numpy(g[f"{grp}_action"].reshape(-1)).cuda())
    torch.cuda.synchronize()
    assert env.last_step_kernel() == "K1f"
    done = g[f"{grp}_terminated"].reshape(-1) | g[f"{grp}_truncated"].reshape(-1)
    assert np.array_equal(obs.cpu().numpy(), g[f"{grp}_ret_obs"].reshape(M, -1))
    assert np.array_equal(rew.cpu().numpy(), g[f"{grp}_reward_scaled"].reshape(-1).astype(np.float32))
    assert np.array_equal(term.cpu().numpy() != 0, g[f"{grp}_terminated"].reshape(-1))
    assert np.array_equal(trunc.cpu().numpy() != 0, g[f"{grp}_truncated"].reshape(-1))
    st = env.get_state()
    for k in STATE:
        assert np.array_equal(st[k][~done], g[f"{grp}_post_{k}"].reshape(-1)[~done]), k
    assert np.all(st["time_step"][done] == 0) and np.all(st["px"][done] == g["reset_state"][0])
    env.close()


@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("N,form,kernel", [(9000, 0, "K1f-table"), (24, 0, "K1f-table"), (4099, 2, "K1f-table"), (300, 1, "K1")])
def test_step_many_is_T_steps(dtype, N, form, kernel):
    """pc_env_step_many == T x pc_env_step: one launch where the handle has the table-driven form (at any batch size, with the 1/den
    table staged for T > 1), T launches of K1 otherwise -- the same rows and the same state either way."""
    T, n = 96, 16
    acts = _actions(T, N, 7 + N)
    a, b = _env(N, TRACKS["big_track"], n, dtype, 1), _env(N, TRACKS["big_track"], n, dtype, form)
    a.reset()
    b.reset()
    rows = [a.step(acts[t])[:4] for t in range(T)]
    many = b.step_many(acts)
    assert b.last_step_kernel() == kernel
    for i, what in enumerate(("obs", "reward", "terminated", "truncated")):
        want = torch.stack([r[i] for r in rows])
        assert many[i].shape == want.shape and torch.equal(many[i], want), what
    sa, sb = a.get_state(), b.get_state()
    for k in STATE:
        assert np.array_equal(sa[k], sb[k]), k
    # ... and on from there: the state the launch left is a state the per-step kernel continues from
    more = _actions(8, N, 99, wild=False)
    for t in range(8):
        ra, rb = a.step(more[t]), b.step(more[t])
        assert torch.equal(ra[0], rb[0]) and torch.equal(ra[1], rb[1])
    a.close()
    b.close()


@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("N", [1024, 4160])
def test_mixed_tracks_in_blocks_and_evenly_interleaved_take_the_table_driven_step(dtype, N):
    """Two tracks in halves: a workgroup stages its own track.  Interleaved i & 1 (every block of 64 envs split evenly): both tracks staged,
    the block's two waves de-interleave it (K1f's TWO form, as rollout_kernel's mode 7) -- with gates_passed / final_obs as well.  An
    uneven interleave (i % 3 == 0) has no table-driven form: K1.  All the generic kernel's rows and state."""
    n, T = 16, 64
    tracks = [TRACKS["track"], TRACKS["big_track"]]
    acts = _actions(T, N, 5)
    halves = (np.arange(N) >= N // 2 // 256 * 256).astype(np.uint8)
    inter = (np.arange(N) & 1).astype(np.uint8)
    uneven = (np.arange(N) % 3 == 0).astype(np.uint8)
    for tid, kernel in ((halves, "K1f-table"), (inter, "K1f-table"), (uneven, "K1")):
        a, b = _env(N, tracks, n, dtype, 1, track_id=tid), _env(N, tracks, n, dtype, 2, track_id=tid)
        a.reset()
        b.reset()
        rows = [a.step(acts[t])[:4] for t in range(T)]
        many = b.step_many(acts)
        assert b.last_step_kernel() == kernel, (kernel, b.last_step_kernel())
        for i in range(4):
            assert torch.equal(many[i], torch.stack([r[i] for r in rows])), i
        for k in STATE:
            assert np.array_equal(a.get_state()[k], b.get_state()[k]), k
        # one more step with the optional outputs
        ga, gb = torch.empty(N, dtype=torch.int32, device="cuda"), torch.empty(N, dtype=torch.int32, device="cuda")
        fa, fb = torch.empty(N, a.obs_dim, device="cuda"), torch.empty(N, a.obs_dim, device="cuda")
        ra, rb = a.step(acts[0], gates_passed=ga, final_obs=fa), b.step(acts[0], gates_passed=gb, final_obs=fb)
        assert b.last_step_kernel() == ("K1f" if kernel != "K1" else "K1")
        assert torch.equal(ra[0], rb[0]) and torch.equal(ra[1], rb[1]) and torch.equal(ga, gb) and torch.equal(fa, fb)
        a.close()
        b.close()


def _oval_track(tmp_path, n_points):
    import json
    from ppo_car_amd.track_tool import make_oval
    path = str(tmp_path / f"oval{n_points}.json")
    with open(path, "w") as f:
        json.dump(make_oval(n_points=n_points, n_gates=24, wobble=0.06, seed=5), f)
    return path


@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("n_points,nv", [(30, 64), (20, 44)])
def test_long_chains_fill_the_wave_of_a_careful_job(tmp_path, dtype, n_points, nv):
    """The whole-wave careful jobs put chain segment j on lane j: the reference's tracks (28 / 20 padded vertices) only ever use the
    first two rows of 16 lanes.  Synthetic circuits of 2 x 30 walls (62 -> 64 chain vertices: EVERY lane, all four rows of the reduction)
    and 2 x 20 walls (44: three rows), irregular, random actions: K1f's rows and state equal the generic kernel's, whose careful path is
    per lane; the persistent kernel (big form) equals the per-step kernels."""
    from ppo_car_amd.ppo import PPOConfig, Trainer
    path = _oval_track(tmp_path, n_points)
    N, T = 3000, 160
    acts = _actions(T, N, n_points)
    a, b = _env(N, path, 16, dtype, 1), _env(N, path, 16, dtype, 2)
    assert a.track_info[0]["n_chain_vertices"] + (-a.track_info[0]["n_chain_vertices"]) % 4 == nv
    a.reset()
    b.reset()
    rows = [a.step(acts[t])[:4] for t in range(T)]
    many = b.step_many(acts)
    assert a.last_step_kernel() == "K1" and b.last_step_kernel().startswith("K1f")
    for i in range(4):
        assert torch.equal(many[i], torch.stack([r[i] for r in rows])), i
    for k in STATE:
        assert np.array_equal(a.get_state()[k], b.get_state()[k]), k
    assert float(many[2].sum()) > 0        # episodes ended: cars reached walls
    a.close()
    b.close()
    res = {}
    for mode in ("mega", "steps"):
        tr = Trainer(PPOConfig(n_envs=20000, n_steps=96, num_rays=16, track=path, rollout_kernel=mode, use_graphs=False, seed=23, env_dtype=dtype),
                     device="cuda")
        for _ in range(2):
            tr.rollout()
            tr.buffer.ptr = 0
        torch.cuda.synchronize()
        assert tr.rollout_mode == ("mega" if mode == "mega" else "steps-eager")
        if mode == "mega":
            # (the 1/den table of these tracks -- 92 / 63 KB -- does not fit beside the weight image: both dtypes form 1/den arithmetically)
            assert tr.envs.last_rollout_kernel() == ("K9" if dtype == "f32" else "K9-literal"), tr.envs.last_rollout_kernel()
        bf = tr.buffer
        res[mode] = [x.clone() for x in (bf.obs_buf, bf.act_buf, bf.rew_buf, bf.val_buf, bf.logprob_buf, bf.term_buf, bf.trunc_buf, tr.next_obs)]
        tr.close()
    for i, (x, y) in enumerate(zip(res["mega"], res["steps"])):
        assert torch.equal(x, y), i


def test_automatic_choice_of_the_step_kernel():
    for N, kernel in ((8191, "K1"), (8192, "K1f")):
        e = pc.VecCarEnv(N, TRACKS["big_track"], num_rays=16, reward_scaling=0.1)
        e.reset()
        e.step(torch.zeros(N, dtype=torch.int64, device="cuda"))
        assert e.last_step_kernel() == kernel and e.get_option("step_form") == 0
        e.close()


def test_shapes_without_a_table_driven_form_take_the_generic_kernel():
    """17 nominal rays (18 actual), a track of 128 walls (oval64): K1, same API."""
    for kw, trk in ((dict(num_rays=17), "big_track"), (dict(num_rays=16), "oval64")):
        e = pc.VecCarEnv(512, TRACKS[trk], reward_scaling=0.1, **kw)
        e.set_option("step_form", 2)
        e.reset()
        e.step(torch.zeros(512, dtype=torch.int64, device="cuda"))
        assert e.last_step_kernel() == "K1", (kw, trk)
        out = e.step_many(torch.zeros(3, 512, dtype=torch.int64, device="cuda"))
        assert e.last_step_kernel() == "K1" and out[0].shape == (3, 512, e.obs_dim)
        e.close()
    e = pc.VecCarEnv(512, TRACKS["big_track"], num_rays=16, reward_scaling=0.1)
    e.set_option("step_form", 2)
    e.reset()
    e.step(torch.zeros(512, dtype=torch.int64, device="cuda"))
    assert e.last_step_kernel() == "K1f"
    e.step(torch.zeros(512, dtype=torch.int64, device="cuda"), gates_passed=torch.empty(512, dtype=torch.int32, device="cuda"),
           final_obs=torch.empty(512, e.obs_dim, device="cuda"))
    assert e.last_step_kernel() == "K1f"       # (the optional outputs are K1f's as well: test_env_gpu.py's goldens run through both kernels)
    assert e.get_option("step_form") == 2
    e.close()


@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_trained_policy_rollout_replayed_through_step_many(dtype):
    """Laps, time limits and ~100 turns of heading inside K1f: the actions a TRAINED policy drew in a 2048-step persistent rollout
    (tests/golden/policy_trained.npz: ~3 laps per episode), replayed from reset through pc_env_step_many, reproduce that rollout's
    observation / reward / flag rows and its final env state bit for bit."""
    from ppo_car_amd.ppo import PPOConfig, Trainer
    from oracle.scenarios import load_trained_policy
    N, T = 16384, 2048
    tr = Trainer(PPOConfig(n_envs=N, n_steps=T, num_rays=16, track=TRACKS["big_track"], rollout_kernel="mega", use_graphs=False, seed=31,
                           env_dtype=dtype), device="cuda")
    load_trained_policy(tr.agent)
    tr.rollout()
    torch.cuda.synchronize()
    b = tr.buffer
    rew = b.rew_buf.clone()
    assert float(rew.max()) > 1.05 and float(b.trunc_buf.sum()) > 0           # laps (+1 +10, scaled 0.1) and time limits happened
    env = _env(N, TRACKS["big_track"], 16, dtype, 0)
    env.reset()
    obs, r, te, tc = env.step_many(b.act_buf.to(torch.int64))
    assert env.last_step_kernel() == "K1f-table"
    assert torch.equal(obs[:-1], b.obs_buf[1:]) and torch.equal(obs[-1], tr.next_obs)
    assert torch.equal(r, rew)
    assert torch.equal(te[:-1], b.term_buf[1:]) and torch.equal(te[-1], tr.next_term)
    assert torch.equal(tc[:-1], b.trunc_buf[1:]) and torch.equal(tc[-1], tr.next_trunc)
    sa, sb = tr.envs.get_state(), env.get_state()
    for k in STATE:
        assert np.array_equal(sa[k], sb[k]), k
    env.close()
    tr.close()


def test_env_only_example_runs_and_reports_identical_rows():
    """examples/env_only.py: the env alone step by step and as one call, compared inside the script."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "examples", "env_only.py"), "--n-envs", "8192", "--n-steps", "64"], capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "identical rows: True" in out.stdout and "(K1f)" in out.stdout and "(K1f-table)" in out.stdout, out.stdout


@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_table_driven_step_inside_a_hip_graph(dtype):
    """pc_env_step / pc_env_step_many never synchronise: a loop of steps (K1f) and a step_many launch captured into ONE HIP graph -- the
    very first launches of their kernels on this handle happen inside the capture -- and replayed twice give what the eager calls give."""
    N, T = 8192, 24
    acts = _actions(3 * T, N, 21, wild=False)
    a, b = _env(N, TRACKS["big_track"], 16, dtype, 0), _env(N, TRACKS["big_track"], 16, dtype, 0)
    a.reset()
    b.reset()
    torch.cuda.synchronize()
    D = a.obs_dim
    cur = torch.zeros(T, N, dtype=torch.int64, device="cuda")
    rows = (torch.empty(T, N, D, device="cuda"), torch.empty(T, N, device="cuda"), torch.empty(T, N, device="cuda"), torch.empty(T, N, device="cuda"))
    many = tuple(torch.empty(T // 2, *r.shape[1:], device="cuda") for r in rows)
    s_ = torch.cuda.Stream()
    with torch.cuda.stream(s_):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for t in range(T // 2):
                b.step(cur[t], out=tuple(r[t] for r in rows))
            b.step_many(cur[T // 2:], out=many)
    torch.cuda.synchronize()
    b.reset()
    for rep in range(2):
        chunk = acts[rep * T:(rep + 1) * T]
        want = [a.step(chunk[t])[:4] for t in range(T)]
        cur.copy_(chunk)
        g.replay()
        torch.cuda.synchronize()
        for i in range(4):
            assert torch.equal(rows[i][:T // 2], torch.stack([w[i] for w in want[:T // 2]])), (rep, i)
            assert torch.equal(many[i], torch.stack([w[i] for w in want[T // 2:]])), (rep, i)
    for k in STATE:
        assert np.array_equal(a.get_state()[k], b.get_state()[k]), k
    a.close()
    b.close()
