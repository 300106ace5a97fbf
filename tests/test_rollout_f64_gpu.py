"""K9d: pc_rollout on PC_DTYPE_F64 handles -- the whole rollout (train.py:173-195) as one persistent launch with the env in the
reference's own float64 (car_env.py:693-760 literally: env_step_core<double>).  Two bars:
  (i)  every buffer and the final env state equal the per-step kernels' (policy_kernel; env_step_kernel<double>) bit for bit;
  (ii) the stored actions replayed through the CPU oracle reproduce EVERY observation, reward and flag of EVERY env bit for bit
       (this dtype has no tolerance: it is the configuration whose index / event exactness holds by construction)."""
import numpy as np
import pytest
import torch

import oracle
from ppo_car_amd.ppo import PPOConfig, Trainer
from conftest import TRACKS

pytestmark = pytest.mark.gpu


def _split(num_rays, n_envs, precision=2):
    """the policy step's work decomposition whose bits the persistent launch reproduces: up to 4096 envs at 16 rays an F64 handle takes
    the SMALL form (12 / 16 rays, up to 8192 envs; K9s, hidden tiles split over the waves: policy_kernel<SPLIT>'s summation order), else the big form (unsplit)"""
    return 1 if (num_rays in (12, 16) and n_envs <= 8192 and precision == 2) else 0


def _expected_kernel(num_rays, n_envs, precision=2, two_equal_loops=True):
    if precision != 2:
        return "K9d-filter"
    if _split(num_rays, n_envs):
        return "K9s-literal"
    # 8193 .. 32768 envs at 16 rays on the chain-packed sweeps' track layouts: 16 envs per wave
    return "K9m-literal" if (num_rays == 16 and 8192 < n_envs <= 32768 and two_equal_loops) else "K9-literal"


def _snap(tr):
    b = tr.buffer
    return [t.clone() for t in (b.obs_buf, b.act_buf, b.rew_buf, b.val_buf, b.logprob_buf, b.term_buf, b.trunc_buf,
                                tr.next_obs, tr.next_term, tr.next_trunc, tr._boot_val if tr._boot_val is not None else tr.next_term)]


def _oracle_exact(cfg, snap, first, track, sel):
    """all T steps of the envs `sel` through the oracle, teacher-forced by the stored actions: bit equality, no exceptions"""
    obs, act, rew, _, _, term, trunc, nobs, nterm, ntrunc = [t.cpu().numpy() for t in snap[:10]]
    T = cfg.n_steps
    ora = oracle.OracleVecEnv(oracle.Track(track), len(sel), num_rays=cfg.num_rays, reward_scaling=cfg.reward_scaling, threads=4)
    o = ora.reset()
    assert np.array_equal(first.cpu().numpy()[sel], o) and np.array_equal(obs[0][:, :][sel], o)
    done = 0
    for t in range(T):
        o, r, te, trn = ora.step(act[t][sel].astype(np.int64))
        nxt_o = obs[t + 1][sel] if t + 1 < T else nobs[sel]
        nxt_te = term[t + 1][sel] if t + 1 < T else nterm[sel]
        nxt_tr = trunc[t + 1][sel] if t + 1 < T else ntrunc[sel]
        assert np.array_equal(nxt_o, o), t
        assert np.array_equal(rew[t][sel], r.astype(np.float32)), t
        assert np.array_equal(nxt_te != 0, te) and np.array_equal(nxt_tr != 0, trn), t
        done += int((te | trn).sum())
    return done


@pytest.mark.parametrize("precision", [2, 1])
@pytest.mark.parametrize("num_rays,n_envs,n_steps", [(16, 1000, 80), (12, 512, 80), (16, 20000, 200), (12, 40000, 64)])
def test_f64_persistent_rollout_is_bitwise_the_per_step_kernels_and_the_oracle(num_rays, n_envs, n_steps, precision):
    if precision == 1 and num_rays != 12:
        pytest.skip("bf16x3 beside nine float64 ray slots per lane is not on the persistent float64 kernel's menu (the per-step kernels run it)")
    res, first = {}, None
    for mode in ("mega", "steps"):
        cfg = PPOConfig(n_envs=n_envs, n_steps=n_steps, num_rays=num_rays, track=TRACKS["big_track"], rollout_kernel=mode, env_dtype="f64",
                        use_graphs=False, seed=21, policy_precision=precision, policy_split=_split(num_rays, n_envs, precision))
        tr = Trainer(cfg, device="cuda")
        if first is None:
            first = tr.next_obs.clone()
        for ep in range(2):                      # two rollouts: the second starts from mid-episode states and fresh Philox counters
            tr.rollout()
            torch.cuda.synchronize()
            assert tr.rollout_mode == ("mega" if mode == "mega" else "steps-eager"), tr.rollout_mode
            if mode == "mega":
                assert tr.envs.last_rollout_kernel() == _expected_kernel(num_rays, n_envs, precision)
            res[(mode, ep)] = _snap(tr)
            tr.buffer.ptr = 0
        res[mode + "_state"] = tr.envs.get_state()
        tr.close()
    for ep in range(2):
        for i, (a, b) in enumerate(zip(res[("mega", ep)][:10], res[("steps", ep)][:10])):
            assert torch.equal(a, b), f"rollout {ep}: buffer {i} differs between the persistent float64 launch and the per-step kernels"
    for k in res["mega_state"]:
        assert np.array_equal(res["mega_state"][k], res["steps_state"][k]), k       # the float64 state itself, heading included
    sel = np.arange(0, n_envs, max(1, n_envs // 256))[:256]
    done = _oracle_exact(cfg, res[("mega", 0)], first, TRACKS["big_track"], sel)
    assert done > 0      # episodes ended inside the replay: auto-reset rows were compared too


def test_f64_persistent_rollout_at_the_target_shape_against_the_oracle():
    """65536 envs x 17 rays (the size BASELINE's target is quoted on), 96 steps, default dispatch: one env of every 32-env wave --
    2048 envs -- replayed through the oracle, everything bit-equal.  (bench.py reports this configuration as exact_f64_value.)"""
    cfg = PPOConfig(n_envs=65536, n_steps=96, num_rays=16, track=TRACKS["big_track"], env_dtype="f64", use_graphs=False, seed=4)
    tr = Trainer(cfg, device="cuda")
    first = tr.next_obs.clone()
    tr.rollout()
    torch.cuda.synchronize()
    assert tr.rollout_mode == "mega"
    snap = _snap(tr)
    tr.close()
    sel = np.arange(0, 65536, 32) + (np.arange(2048) % 32)
    assert _oracle_exact(cfg, snap, first, TRACKS["big_track"], sel) > 0


def test_f64_persistent_rollout_mixed_tracks_and_whole_epochs():
    """track.json + big_track.json in blocks (BASELINE configs[4]'s layout) through K9d: bitwise the per-step kernels; then whole
    epochs (rollout + GAE + update, bootstrap value from the launch's own critic pass) run and stay finite."""
    tracks = [TRACKS["track"], TRACKS["big_track"]]
    res = {}
    for mode in ("mega", "steps"):
        tr = Trainer(PPOConfig(n_envs=2048, n_steps=64, num_rays=16, track=tracks, rollout_kernel=mode, env_dtype="f64", use_graphs=False, seed=9,
                               policy_split=_split(16, 2048)), device="cuda")
        tr.rollout()
        torch.cuda.synchronize()
        assert tr.rollout_mode == ("mega" if mode == "mega" else "steps-eager")
        res[mode] = _snap(tr)
        tr.close()
    for i, (a, b) in enumerate(zip(res["mega"][:10], res["steps"][:10])):
        assert torch.equal(a, b), i
    tr = Trainer(PPOConfig(n_envs=1024, n_steps=64, batch_size=64, train_iters=2, num_rays=16, track=TRACKS["big_track"], env_dtype="f64"), device="cuda")
    for _ in range(3):
        s = tr.run_epoch()
        assert tr.rollout_mode == "mega" and np.isfinite(s["losses/total_loss"])
    tr.close()


def test_f64_at_33_rays_selector_form_and_the_fallback_to_the_per_step_kernels():
    """33 rays in float64: the selector form runs it as one persistent launch (17 ray slots per lane in two sweep passes, no 1/den
    table: K9's cfg2 kernel with the literal arithmetic); the FILTER form is not built at that width (17 float64 slots beside the
    policy state spilled) -- with PC_OPT_ROLLOUT_FAST = 0 pc_rollout answers PC_ERR_UNSUPPORTED and the trainer runs the per-step
    kernels.  Bitwise each other and bit-exact against the oracle."""
    res = {}
    for fast in (1, 0):
        cfg = PPOConfig(n_envs=2048, n_steps=64, num_rays=32, track=TRACKS["big_track"], env_dtype="f64", use_graphs=False, seed=2, rollout_fast=fast,
                        policy_split=0)
        tr = Trainer(cfg, device="cuda")
        first = tr.next_obs.clone()
        tr.rollout()
        torch.cuda.synchronize()
        assert tr.rollout_mode == ("mega" if fast else "steps-eager")
        if fast:
            assert tr.envs.last_rollout_kernel() == "K9-literal"
        res[fast] = _snap(tr)
        tr.close()
    for i, (a, b) in enumerate(zip(res[1][:10], res[0][:10])):
        assert torch.equal(a, b), i
    assert _oracle_exact(cfg, res[1], first, TRACKS["big_track"], np.arange(0, 2048, 8)) > 0


# ---- the SELECTOR form of the persistent float64 rollout (PC_KERNEL_K9_LITERAL: K9 with the literal arithmetic behind the float32 sweep)

def _cross_track(path, start=(560.0, 135.0), angle=0.0, inner_kind="plus"):
    """Two plus-shaped wall loops (12 walls each: the chain-packed sweep's layout) on INTEGER pixel coordinates, the start pose on the
    line of an inner wall, heading along an axis: rays run exactly through vertices, exactly along walls and exactly parallel to
    them -- every tie the literal arithmetic decides by its roundings (cast_exact's comment) -- at every reset and beyond."""
    import json
    W, H = 1280.0, 720.0
    outer = [(480, 90), (800, 90), (800, 270), (1120, 270), (1120, 450), (800, 450), (800, 630), (480, 630), (480, 450), (160, 450), (160, 270),
             (480, 270), (480, 90)]
    inner = [(560, 180), (720, 180), (720, 315), (960, 315), (960, 405), (720, 405), (720, 540), (560, 540), (560, 405), (320, 405), (320, 315),
             (560, 315), (560, 180)]
    if inner_kind == "octagon":      # 12 + 8 walls: NOT two equal loops -- the generic sweeps instead of the chain-packed one
        inner = [(600, 180), (680, 180), (760, 315), (760, 405), (680, 540), (600, 540), (520, 405), (520, 315), (600, 180)]
    gates = [(704, 90), (704, 180), (880, 270), (880, 315), (1120, 360), (960, 360), (880, 450), (880, 405)]
    frac = lambda pts: [[x / W, y / H] for x, y in pts]
    for x, y in outer + inner:
        assert (x / W) * W == x and (y / H) * H == y      # the loader's x * 1280, y * 720 give the integers back
    with open(path, "w") as f:
        json.dump({"outer_track_points": frac(outer), "inner_track_points": frac(inner), "reward_gates": frac(gates),
                   "initial_position": [start[0] / W, start[1] / H], "initial_angle": angle}, f)
    return str(path)


def _rollouts(cfg_kw, n_rollouts=2):
    tr = Trainer(PPOConfig(**cfg_kw), device="cuda")
    first = tr.next_obs.clone()
    snaps, kernels = [], []
    for _ in range(n_rollouts):
        tr.rollout()
        torch.cuda.synchronize()
        snaps.append(_snap(tr))
        kernels.append(tr.envs.last_rollout_kernel() if tr.rollout_mode == "mega" else tr.rollout_mode)
        tr.buffer.ptr = 0
    state = tr.envs.get_state()
    tr.close()
    return first, snaps, kernels, state


def test_f64_selector_form_runs_by_default_and_equals_the_filter_form_bit_for_bit():
    """big_track, 16 rays, 20480 envs (the big form): the default dispatch of an F64 handle is the selector form; PC_OPT_ROLLOUT_FAST = 0 takes the filter form
    (every ray x wall pair in float64); both fill every buffer and leave the float64 state with the same bits -- two rollouts each,
    the second from mid-episode states -- and a sample of the envs replays through the oracle bit for bit."""
    kw = dict(n_envs=20480, n_steps=160, num_rays=16, track=TRACKS["big_track"], env_dtype="f64", use_graphs=False, seed=33, rollout_kernel="mega")
    first, sel, k_sel, st_sel = _rollouts(dict(kw))
    _, fil, k_fil, st_fil = _rollouts(dict(kw, rollout_fast=0))
    assert k_sel == ["K9m-literal"] * 2 and k_fil == ["K9d-filter"] * 2, (k_sel, k_fil)
    for ep in range(2):
        for i, (a, b) in enumerate(zip(sel[ep][:10], fil[ep][:10])):
            assert torch.equal(a, b), (ep, i)
    for k in st_sel:
        assert np.array_equal(st_sel[k], st_fil[k]), k
    cfg = PPOConfig(**kw)
    assert _oracle_exact(cfg, sel[0], first, TRACKS["big_track"], np.arange(0, 20480, 80)) > 0


@pytest.mark.parametrize("start,angle,inner,num_rays", [((560.0, 135.0), 0.0, "plus", 16), ((640.0, 135.0), 90.0, "plus", 16),
                                                        ((520.0, 180.0), 45.0, "plus", 16), ((600.0, 135.0), 0.0, "octagon", 16),
                                                        ((560.0, 135.0), 0.0, "plus", 12), ((600.0, 135.0), 180.0, "octagon", 12)])
@pytest.mark.parametrize("n_envs", [4096, 8192, 20480])
def test_f64_selector_form_on_a_track_of_ties(tmp_path, start, angle, inner, num_rays, n_envs):
    """The cross track (integer coordinates, axis-parallel walls, the start pose on a wall's line): rays through vertices, along walls,
    parallel to walls.  The selector form flags what float32 cannot decide and resolves it with the literal loop over all walls:
    bitwise the per-step float64 kernels, and bitwise the oracle for every env replayed.  Two equal loops take the chain-packed sweep,
    the octagon variant (12 + 8 walls) and 12 rays the generic ones; 4096 / 8192 envs at 16 rays the small form's two variants (K9s,
    the sweep out of LDS), 20480 the big form."""
    track = _cross_track(tmp_path / "cross.json", start, angle, inner)
    kw = dict(n_envs=n_envs, n_steps=128, num_rays=num_rays, track=track, env_dtype="f64", use_graphs=False, seed=5,
              policy_split=_split(num_rays, n_envs))
    first, mega, k_mega, st_mega = _rollouts(dict(kw, rollout_kernel="mega"))
    _, steps, k_steps, st_steps = _rollouts(dict(kw, rollout_kernel="steps"))
    assert k_mega == [_expected_kernel(num_rays, n_envs, 2, inner == "plus")] * 2 and k_steps == ["steps-eager"] * 2, (k_mega, k_steps)
    for ep in range(2):
        for i, (a, b) in enumerate(zip(mega[ep][:10], steps[ep][:10])):
            assert torch.equal(a, b), (ep, i)
    for k in st_mega:
        assert np.array_equal(st_mega[k], st_steps[k]), k
    cfg = PPOConfig(**kw)
    assert _oracle_exact(cfg, mega[0], first, track, np.arange(0, n_envs, n_envs // 256)) > 0


def test_f64_selector_form_needs_rotations_on_the_table_and_set_state_can_take_them_off():
    """pc_env_set_state with a rotation no episode reaches (or a row further from start_rot than the env's time step allows): the
    next pc_rollout takes the generic kernel K9d -- which hashes / evaluates such angles -- and still equals the per-step kernels; after
    pc_env_reset the literal form of K9 is back."""
    res = {}
    for mode in ("mega", "steps"):
        tr = Trainer(PPOConfig(n_envs=2048, n_steps=48, num_rays=16, track=TRACKS["big_track"], env_dtype="f64", use_graphs=False, seed=8,
                               rollout_kernel=mode, policy_split=0, rollout_form=0), device="cuda")
        st = tr.envs.get_state()
        rot = st["rot"].copy()
        rot[::7] += 0.125                 # not start_rot + 5 k: no row of the rotation table
        tr.envs.set_state(rot=rot)
        tr.rollout()
        torch.cuda.synchronize()
        res[mode] = _snap(tr)
        if mode == "mega":
            assert tr.envs.last_rollout_kernel() == "K9d-selector"      # (the generic kernel: angles off the table are hashed / evaluated)
            tr.next_obs.copy_(tr.envs.reset()[0])
            tr.buffer.ptr = 0
            tr.rollout()
            torch.cuda.synchronize()
            assert tr.envs.last_rollout_kernel() == "K9-literal"      # (rollout_form = 0: the big form at any batch size)
        tr.close()
    for i, (a, b) in enumerate(zip(res["mega"][:10], res["steps"][:10])):      # (obs row 0 is the pre-set_state observation in both)
        assert torch.equal(a, b), i


@pytest.mark.parametrize("n_envs", [4096, 8192, 20480])
def test_f64_selector_form_on_walls_that_cross_and_touch(tmp_path, n_envs):
    """test_env_gpu's junction track (a T-junction and two walls that cross: segments the host marks PC_SEG_SCAN, where a float32
    selector cannot order hits by looking at chain neighbours): every ray that selects one of them takes the literal loop over all
    walls.  Bitwise the per-step float64 kernels and the oracle -- with NO tolerance, unlike the float32 dtype on this track."""
    from test_env_gpu import _junction_track_json
    track = _junction_track_json(str(tmp_path / "junction.json"))
    kw = dict(n_envs=n_envs, n_steps=160, num_rays=16, track=track, env_dtype="f64", use_graphs=False, seed=19, policy_split=_split(16, n_envs))
    first, mega, k_mega, st_mega = _rollouts(dict(kw, rollout_kernel="mega"), 1)
    _, steps, k_steps, st_steps = _rollouts(dict(kw, rollout_kernel="steps"), 1)
    assert k_mega == [_expected_kernel(16, n_envs, 2, False)] and k_steps == ["steps-eager"]      # (the junction track: one loop of 4, one chain of 5)
    for i, (a, b) in enumerate(zip(mega[0][:10], steps[0][:10])):
        assert torch.equal(a, b), i
    for k in st_mega:
        assert np.array_equal(st_mega[k], st_steps[k]), k
    assert _oracle_exact(PPOConfig(**kw), mega[0], first, track, np.arange(0, n_envs, n_envs // 256)) > 0


@pytest.mark.parametrize("n_envs,n_steps", [(20000, 160), (65536, 96), (1000, 200)])
def test_strictest_cell_float64_env_and_exact_fp32_policy_chain_in_one_persistent_launch(n_envs, n_steps):
    """dtype f64 AND policy_precision 0 (v_mfma_f32_16x16x4_f32: the fp32 fmaf chain of model.py's Linear layers): every number of the
    rollout in the reference's own arithmetic.  pc_rollout runs it as K9's literal form with the fp32 weight image
    (rollout_kernel<6, 9, 0, 1, true>, 16 -> 17 rays, the big form): bitwise the per-step kernels (policy_kernel<6, false, 0>;
    env_step_kernel<double>), float64 state included, and every observation / reward / flag of 256 envs equal to the oracle's."""
    res, first = {}, None
    for mode in ("mega", "steps"):
        cfg = PPOConfig(n_envs=n_envs, n_steps=n_steps, num_rays=16, track=TRACKS["big_track"], rollout_kernel=mode, env_dtype="f64",
                        use_graphs=False, seed=23, policy_precision=0, policy_split=0)
        tr = Trainer(cfg, device="cuda")
        if first is None:
            first = tr.next_obs.clone()
        tr.rollout()
        torch.cuda.synchronize()
        assert tr.rollout_mode == ("mega" if mode == "mega" else "steps-eager"), tr.rollout_mode
        if mode == "mega":
            assert tr.envs.last_rollout_kernel() == "K9-literal" and tr.agent.policy_form()[0] == 0
        res[mode] = _snap(tr)
        res[mode + "_state"] = tr.envs.get_state()
        tr.close()
    for i, (a, b) in enumerate(zip(res["mega"][:10], res["steps"][:10])):
        assert torch.equal(a, b), f"buffer {i} differs between the persistent launch and the per-step kernels (f64 env, fp32 policy chain)"
    for k in res["mega_state"]:
        assert np.array_equal(res["mega_state"][k], res["steps_state"][k]), k
    sel = np.arange(0, n_envs, max(1, n_envs // 256))[:256]
    assert _oracle_exact(cfg, res["mega"], first, TRACKS["big_track"], sel) > 0
