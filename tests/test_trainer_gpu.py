"""GPU tests of the whole loop: rollout plumbing against the reference's loop semantics, the HIP-graph
update against the eager one, and a short end-to-end training run."""
import numpy as np
import pytest
import torch

import oracle
import ppo_car_amd as pc
from ppo_car_amd.ppo import PPOConfig, PPOLearner, Trainer
from conftest import TRACKS

pytestmark = pytest.mark.gpu


def _cfg(**kw):
    base = dict(n_envs=256, n_steps=64, batch_size=32, train_iters=3, track=TRACKS["big_track"], num_rays=16, seed=5)
    base.update(kw)
    return PPOConfig(**base)


@pytest.mark.parametrize("policy", ["fused", "sample", "torch"])
def test_rollout_buffer_is_consistent_with_the_env(policy):
    """Replaying the actions the rollout stored through the float64 oracle reproduces the stored rewards, flags and
    observations (float32 kernel: within tolerance), with the reference's row conventions: flags stored at row t are
    those that preceded obs t (train.py:176-177,195)."""
    cfg = _cfg(policy=policy, env_dtype="f64")
    tr = Trainer(cfg, device="cuda")
    first_obs = tr.next_obs.clone()
    tr.rollout()
    torch.cuda.synchronize()
    buf = tr.buffer
    T, N = cfg.n_steps, cfg.n_envs
    acts = buf.act_buf.cpu().numpy().astype(np.int64)
    assert acts.min() >= 0 and acts.max() <= 8
    ora = oracle.OracleVecEnv(oracle.Track(cfg.track), N, num_rays=cfg.num_rays, reward_scaling=cfg.reward_scaling, threads=4)
    o = ora.reset()
    assert np.array_equal(first_obs.cpu().numpy(), o) and np.array_equal(buf.obs_buf[0].cpu().numpy(), o)
    assert float(buf.term_buf[0].abs().sum()) == 0 and float(buf.trunc_buf[0].abs().sum()) == 0
    for t in range(T):
        o, r, te, trn = ora.step(acts[t])
        nxt_obs = buf.obs_buf[t + 1] if t + 1 < T else tr.next_obs
        nxt_te = buf.term_buf[t + 1] if t + 1 < T else tr.next_term
        nxt_tr = buf.trunc_buf[t + 1] if t + 1 < T else tr.next_trunc
        assert np.array_equal(nxt_obs.cpu().numpy(), o), t
        assert np.array_equal(buf.rew_buf[t].cpu().numpy(), r.astype(np.float32)), t
        assert np.array_equal(nxt_te.cpu().numpy() != 0, te) and np.array_equal(nxt_tr.cpu().numpy() != 0, trn)
    # stored log-probs / values are those of the stored action under the rollout policy
    with torch.no_grad():
        _, lp, _, v = tr.agent.get_action_and_value(buf.obs_buf.view(-1, *tr.obs_dim), buf.act_buf.view(-1))
    assert torch.allclose(lp, buf.logprob_buf.view(-1), atol=2e-5)
    assert torch.allclose(v.view(-1), buf.val_buf.view(-1), atol=2e-5)
    assert tr.global_step_idx == T * N
    tr.close()


def test_graph_update_equals_eager_update():
    """Same initial parameters, same trajectories, same index draws: the HIP-graph minibatch steps and the eager
    ones give the same parameters (capturable Adam computes its bias correction on the device: 1e-6 slack)."""
    outs = {}
    for use_graphs in (False, True):
        cfg = _cfg(use_graphs=use_graphs)
        tr = Trainer(cfg, device="cuda")
        tr.rollout()
        tr.update()
        tr.rollout()          # second epoch: graphs are replayed, not rebuilt
        tr.update()
        torch.cuda.synchronize()
        outs[use_graphs] = (tr.learner.flat_param.clone(), tr.learner.metrics.clone(), tr.learner.current_lr())
        if use_graphs:
            assert tr.learner._graph_key is not None or tr.learner._epoch_graph is not None
        tr.close()
    (p0, m0, lr0), (p1, m1, lr1) = outs[False], outs[True]
    assert lr0 == pytest.approx(lr1, rel=1e-6) and lr0 == pytest.approx(3e-4 * 0.99 ** 2, rel=1e-6)
    assert torch.allclose(p0, p1, atol=2e-5, rtol=1e-4)
    assert torch.allclose(m0, m1, atol=1e-3, rtol=1e-3)


def test_learner_on_gpu_matches_cpu_reference_arithmetic():
    """One eager minibatch step on the GPU vs the same step on the CPU (same torch ops, different device)."""
    torch.manual_seed(1)
    a_cpu = pc.Agent(23, 9)
    a_gpu = pc.Agent(23, 9)
    a_gpu.load_state_dict(a_cpu.state_dict())
    a_gpu.cuda()
    cfg = PPOConfig(n_envs=4, n_steps=64, batch_size=64, train_iters=1, use_graphs=False)
    L0, L1 = PPOLearner(a_cpu, cfg, "cpu"), PPOLearner(a_gpu, cfg, "cuda")
    g = torch.Generator().manual_seed(0)
    batch = (torch.randn(64, 23, generator=g), torch.randint(0, 9, (64,), generator=g).float(), -torch.rand(64, generator=g),
             torch.randn(64, generator=g), torch.randn(64, generator=g))
    L0.minibatch_step(*batch)
    L1.minibatch_step(*[t.cuda() for t in batch])
    assert torch.allclose(L0.flat_param, L1.flat_param.cpu(), atol=1e-6)
    assert torch.allclose(L0.metrics, L1.metrics.cpu(), atol=1e-4)


def test_short_training_run_improves_reward_and_logs_reference_scalars():
    cfg = _cfg(n_envs=1024, n_steps=128, batch_size=256, train_iters=10, num_rays=12, learning_rate=1e-3, full_sweep=False)
    tr = Trainer(cfg, device="cuda")
    hist = [tr.run_epoch() for _ in range(12)]
    for k in ("losses/policy_loss", "losses/value_loss", "losses/entropy", "losses/total_loss", "charts/avg_reward",
              "charts/learning_rate", "charts/SPS"):
        assert k in hist[0] and np.isfinite(hist[-1][k])
    assert hist[-1]["global_step"] == 12 * 1024 * 128
    assert hist[-1]["charts/learning_rate"] == pytest.approx(1e-3 * 0.99 ** 12, rel=1e-5)
    first, last = np.mean([h["charts/avg_reward"] for h in hist[:3]]), np.mean([h["charts/avg_reward"] for h in hist[-3:]])
    assert last > first                     # per-step reward goes up (fewer crashes, more forward / gates)
    sd = tr.agent.state_dict()              # what train.py:283 saves
    assert set(sd) == {"actor.0.weight", "actor.0.bias", "actor.2.weight", "actor.2.bias",
                       "critic.0.weight", "critic.0.bias", "critic.2.weight", "critic.2.bias"}
    tr.close()


def test_graph_rollout_is_bitwise_the_eager_rollout():
    """The captured-and-replayed rollout (one HIP graph for all steps) and the eager one draw from the same
    Philox counters (device-side base + step index), so whole buffers must agree bit for bit."""
    res = {}
    for use_graphs in (False, True):
        cfg = _cfg(use_graphs=use_graphs, n_envs=512, n_steps=48, rollout_kernel="steps")
        tr = Trainer(cfg, device="cuda")
        snaps = []
        for ep in range(3):            # graph mode: epoch 0 eager, capture before epoch 1, replay epochs 1 and 2
            tr.rollout()
            torch.cuda.synchronize()
            snaps.append([t.clone() for t in (tr.buffer.obs_buf, tr.buffer.act_buf, tr.buffer.rew_buf, tr.buffer.val_buf,
                                              tr.buffer.logprob_buf, tr.buffer.term_buf, tr.next_obs)])
            tr.buffer.ptr = 0          # no update in between: weights stay fixed, so both modes see the same policy
        res[use_graphs] = snaps
        assert (tr._rollout_graph is not None) == use_graphs
        assert int(tr.rng_base) == 3 * 48
        tr.close()
    for ep in range(3):
        for a, b in zip(res[False][ep], res[True][ep]):
            assert torch.equal(a, b), ep
    assert not torch.equal(res[True][0][1], res[True][1][1])     # different epochs draw different actions


@pytest.mark.parametrize("B,D", [(512, 23), (64, 18), (1000, 39)])
def test_fused_update_kernels_match_torch_minibatch_steps(B, D):
    """pc_ppo_gather + pc_ppo_loss (+ torch autograd through the MLPs) + pc_clip_adam  vs  the reference's torch
    ops (ppo_loss, clip_grad_norm_, Adam): same parameters and metric sums after several minibatch steps."""
    M = 4096
    g = torch.Generator().manual_seed(B)
    obs = torch.rand(M, D, generator=g).cuda()
    act = torch.randint(0, 9, (M,), generator=g).float().cuda()
    lp = (-torch.rand(M, generator=g) * 2.5).cuda()
    adv = (torch.randn(M, generator=g) * 3 + 0.5).cuda()
    ret = torch.randn(M, generator=g).cuda()
    idxs = [torch.randperm(M, generator=g)[:B].cuda() for _ in range(6)]
    res = {}
    for fused in (False, True):
        torch.manual_seed(7)
        agent = pc.Agent(D, 9).cuda()
        cfg = PPOConfig(n_envs=8, n_steps=B, batch_size=B, train_iters=1, use_graphs=False, fused_update=fused, max_grad_norm=0.5)
        L = PPOLearner(agent, cfg, "cuda")
        if fused:
            L._fused_alloc(D, 9)
        grads = []
        for i in idxs:
            if fused:
                L.fused_minibatch_step(i, obs, act, lp, adv, ret)
            else:
                L.minibatch_step(obs[i], act[i], lp[i], adv[i], ret[i])
            grads.append(L.flat_grad.clone())
        res[fused] = (L.flat_param.clone(), L.metrics.clone(), grads)
    (p0, m0, g0), (p1, m1, g1) = res[False], res[True]
    assert torch.allclose(g0[0], g1[0], atol=1e-6, rtol=1e-4)          # clipped gradients of the first step
    assert torch.allclose(m0, m1, atol=1e-4, rtol=1e-5)                # metric sums (policy, value, entropy, total)
    assert torch.allclose(p0, p1, atol=3e-6, rtol=1e-5)                # parameters after 6 clip+Adam steps
    assert float((p0 - p1).abs().max()) < 3e-6


def test_fused_and_torch_updates_agree_inside_the_trainer():
    outs = {}
    for fused in (False, True):
        cfg = _cfg(fused_update=fused, use_graphs=True)
        tr = Trainer(cfg, device="cuda")
        tr.run_epoch(sync=False)
        tr.run_epoch(sync=False)
        torch.cuda.synchronize()
        outs[fused] = (tr.learner.flat_param.clone(), tr.learner.current_lr())
        tr.close()
    assert outs[False][1] == pytest.approx(outs[True][1], rel=1e-6)
    # identical rollouts in epoch 1 (same seeds, same init); the second rollout runs on slightly different weights,
    # so only a loose agreement is meaningful after two epochs
    assert torch.allclose(outs[False][0], outs[True][0], atol=5e-3)


@pytest.mark.parametrize("form", [0, 1, 2, 3])
@pytest.mark.parametrize("precision", [2, 1, 0])
@pytest.mark.parametrize("num_rays,n_envs", [(16, 1000), (12, 512), (32, 300)])
def test_persistent_rollout_kernel_is_bitwise_the_two_kernel_rollout(num_rays, n_envs, precision, form):
    """pc_rollout (one persistent launch, weights in LDS, env state in registers) must fill the buffer with exactly the
    bits of the per-step policy_kernel / env_step_kernel sequence, over several epochs (auto-resets included), in both
    of its forms: 0 = 256 envs per workgroup, every wave independent (the whole-tile policy kernel's summation order);
    1 = 32 envs per workgroup, hidden tiles split over the waves (the split policy kernel's summation order); 2 / 3 = the
    same two without the 1/den table in LDS (the sweep forms den and its reciprocal itself: same bits by construction)."""
    res = {}
    for mode in ("steps", "mega"):
        # policy_split = form & 1: the same fp32 summation order in the per-step policy kernel as in this rollout form
        cfg = _cfg(rollout_kernel=mode, use_graphs=False, n_envs=n_envs, n_steps=80, num_rays=num_rays, policy_split=form & 1,
                   policy_precision=precision, rollout_form=form)
        tr = Trainer(cfg, device="cuda")
        snaps = []
        for ep in range(3):
            tr.rollout()
            torch.cuda.synchronize()
            # 32 -> 33 rays with the fp32 weight image: image + a 256-env observation tile exceed 160 KB of LDS -> form 0
            # reports PC_ERR_UNSUPPORTED and the trainer falls back to the two-kernel loop; form 1 fits.  The 95 KB fp16x2
            # image fits both forms, the 141 KB bf16x3 image (two K blocks at D = 39) neither.
            mega_ok = mode == "mega" and (num_rays != 32 or precision == 2 or (precision == 0 and form & 1))
            assert tr.rollout_mode == ("mega" if mega_ok else "steps-eager")
            b = tr.buffer
            snaps.append([t.clone() for t in (b.obs_buf, b.act_buf, b.rew_buf, b.val_buf, b.logprob_buf, b.term_buf,
                                              b.trunc_buf, tr.next_obs, tr.next_term, tr.next_trunc)])
            tr.buffer.ptr = 0
        st = tr.envs.get_state()
        res[mode] = (snaps, st)
        tr.close()
    for ep in range(3):
        for i, (a, b) in enumerate(zip(res["steps"][0][ep], res["mega"][0][ep])):
            assert torch.equal(a, b), (ep, i)
    for k in res["steps"][1]:
        assert np.array_equal(res["steps"][1][k], res["mega"][1][k]), k
    assert float(res["mega"][0][2][5].sum()) > 0        # episodes ended (terminations) inside the window


def test_rollout_forms_agree_with_default_dispatch():
    """Automatic dispatch: up to 8192 envs pc_rollout takes the 32-env-per-workgroup form and the per-step policy
    kernel its split form -- the two must still be bit-identical (this is what a user switching rollout_kernel sees)."""
    res = {}
    for mode in ("steps", "mega"):
        tr = Trainer(_cfg(rollout_kernel=mode, use_graphs=False, n_envs=4096, n_steps=48, num_rays=16), device="cuda")
        tr.rollout()
        torch.cuda.synchronize()
        assert tr.rollout_mode == ("mega" if mode == "mega" else "steps-eager")
        b = tr.buffer
        res[mode] = [t.clone() for t in (b.obs_buf, b.act_buf, b.rew_buf, b.val_buf, b.logprob_buf, b.term_buf, b.trunc_buf,
                                         tr.next_obs)]
        tr.close()
    for i, (a, b) in enumerate(zip(res["steps"], res["mega"])):
        assert torch.equal(a, b), i


@pytest.mark.parametrize("env_dtype", ["f32", "f64"])
@pytest.mark.parametrize("n_envs,mixed", [(20480, False), (32768, True), (12345, False)])
def test_sixteen_envs_per_wave_form_fills_the_same_buffers(n_envs, mixed, env_dtype):
    """8193 .. 32768 envs at 16 rays: pc_rollout takes K9 with 16 envs per wave (4 lanes per env, one policy column tile) -- two waves
    per SIMD where 32-env waves leave one.  Bitwise the 32-envs-per-wave form (PC_OPT_ROLLOUT_FORM = 0) and the per-step kernels, on
    big_track and on the mixed batch of BASELINE configs[4]'s shard, with a ragged last workgroup (12345 envs), in both dtypes; the
    form can be forced at any size (PC_OPT_ROLLOUT_FORM = 4)."""
    track = [TRACKS["track"], TRACKS["big_track"]] if mixed else TRACKS["big_track"]
    res, kern = {}, {}
    for name, kw in (("auto", dict(rollout_kernel="mega")), ("classic", dict(rollout_kernel="mega", rollout_form=0)), ("steps", dict(rollout_kernel="steps"))):
        tr = Trainer(_cfg(track=track, use_graphs=False, n_envs=n_envs, n_steps=40, num_rays=16, env_dtype=env_dtype, policy_split=0, **kw), device="cuda")
        for _ in range(2):
            tr.rollout()
            tr.buffer.ptr = 0
        torch.cuda.synchronize()
        kern[name] = tr.envs.last_rollout_kernel() if tr.rollout_mode == "mega" else tr.rollout_mode
        b = tr.buffer
        res[name] = [t.clone() for t in (b.obs_buf, b.act_buf, b.rew_buf, b.val_buf, b.logprob_buf, b.term_buf, b.trunc_buf, tr.next_obs)]
        st = tr.envs.get_state()
        res[name].append(torch.from_numpy(np.stack([st[k].astype(np.float64) for k in sorted(st)])))
        tr.close()
    lit = "-literal" if env_dtype == "f64" else ""
    assert kern == {"auto": "K9m" + lit, "classic": "K9" + lit, "steps": "steps-eager"}, kern
    for other in ("classic", "steps"):
        for i, (a, b) in enumerate(zip(res["auto"], res[other])):
            assert torch.equal(a, b), (other, i)


def test_sixteen_envs_per_wave_form_can_be_forced_at_any_size():
    res = {}
    for form in (-1, 4):
        tr = Trainer(_cfg(use_graphs=False, n_envs=65536, n_steps=24, num_rays=16, rollout_kernel="mega", rollout_form=form), device="cuda")
        tr.rollout()
        torch.cuda.synchronize()
        assert tr.envs.last_rollout_kernel() == ("K9m" if form == 4 else "K9")
        b = tr.buffer
        res[form] = [t.clone() for t in (b.obs_buf, b.act_buf, b.rew_buf, b.val_buf, b.logprob_buf, b.term_buf, b.trunc_buf, tr.next_obs)]
        tr.close()
    for i, (a, b) in enumerate(zip(res[-1], res[4])):
        assert torch.equal(a, b), i


@pytest.mark.parametrize("n_envs,form", [(4096, -1), (1000, 0), (8192, 0), (1000, 1), (65536, -1)])
def test_mixed_track_batches_through_the_persistent_rollout_kernel(n_envs, form):
    """BASELINE configs[4] (track.json and big_track.json in one batch): with every aligned block of 32 envs on one track
    pc_rollout steps the batch bit-identically to the per-step kernels -- in the fast modes when every WORKGROUP's envs lie on
    one track (it stages that track's tables in LDS: 4096 / 65536 default dispatch, 8192 big form, 1000 small form), else
    with every wave reading its own track's tables from global memory (1000 envs, big form: the halves meet at env 480);
    with the tracks interleaved env by env it reports PC_ERR_UNSUPPORTED and the trainer takes the per-step path."""
    tracks = [TRACKS["track"], TRACKS["big_track"]]
    res = {}
    for mode in ("steps", "mega"):
        tr = Trainer(_cfg(track=tracks, rollout_kernel=mode, use_graphs=False, n_envs=n_envs, n_steps=48, num_rays=16,
                          rollout_form=form, policy_split=-1 if form < 0 else form & 1), device="cuda")
        for _ in range(2):
            tr.rollout()
            tr.buffer.ptr = 0
        torch.cuda.synchronize()
        assert tr.rollout_mode == ("mega" if mode == "mega" else "steps-eager")
        b = tr.buffer
        res[mode] = [t.clone() for t in (b.obs_buf, b.act_buf, b.rew_buf, b.val_buf, b.logprob_buf, b.term_buf, tr.next_obs)]
        res[mode + "_state"] = tr.envs.get_state()
        tr.close()
    for i, (a, b) in enumerate(zip(res["steps"], res["mega"])):
        assert torch.equal(a, b), i
    for k in res["steps_state"]:
        assert np.array_equal(res["steps_state"][k], res["mega_state"][k]), k
    st = res["mega_state"]                                   # both tracks really ran: 45 vs 55 gates bound next_gate
    half = n_envs // 2 // 32 * 32
    assert st["next_gate"][:half].max() < 45 and float(res["mega"][5].sum()) > 0
    tr = Trainer(_cfg(track=tracks, track_interleave=True, rollout_kernel="mega", use_graphs=False, n_envs=512, n_steps=8, num_rays=16),
                 device="cuda")
    tr.rollout()
    assert tr.rollout_mode == "mega" and tr.envs.last_rollout_kernel() == "K9"      # (round 6: the two-track fast form, the env step once per track of a wave)
    tr.close()


def test_generic_mode_at_33_rays_is_not_built_and_falls_back_to_the_per_step_kernels():
    """pc_rollout's generic mode (here: the fast mode switched off on the env handle) exists at 12 and 17 rays; at 33 rays with
    split operands those kernels spilled and are not built: PC_ERR_UNSUPPORTED, and the trainer runs the two-kernel loop instead
    (train.py:173-195 either way; the forms are bit-identical wherever both exist)."""
    for num_rays, want in ((16, "mega"), (32, "steps-eager")):
        tr = Trainer(_cfg(rollout_kernel="mega", use_graphs=False, n_envs=20000, n_steps=4, num_rays=num_rays), device="cuda")
        tr.envs.set_option("rollout_fast", 0)
        tr.rollout()
        torch.cuda.synchronize()
        assert tr.rollout_mode == want, (num_rays, tr.rollout_mode)
        assert torch.isfinite(tr.buffer.obs_buf[:4]).all()
        tr.close()


@pytest.mark.parametrize("num_rays,batch,graphs", [(16, 512, True), (12, 100, False), (32, 64, False)])
def test_prepared_minibatches_are_bitwise_the_in_kernel_gather(num_rays, batch, graphs):
    """pc_ppo_prepare + pc_ppo_minibatch_prepared (samples, scalars and advantage statistics of all minibatches gathered
    once per epoch) against pc_ppo_minibatch gathering for itself: identical parameters, Adam state and logged sums."""
    res = {}
    for prep in (False, True):
        cfg = _cfg(prepared_minibatches=prep, use_graphs=graphs, n_envs=256, n_steps=64, batch_size=batch, train_iters=3,
                   num_rays=num_rays, seed=7)
        tr = Trainer(cfg, device="cuda")
        scal = [tr.run_epoch() for _ in range(3)]
        L = tr.learner
        res[prep] = ([t.clone() for t in (L.flat_param, L.exp_avg, L.exp_avg_sq, L.step_count, L.metrics, L.flat_grad)], scal)
        assert (getattr(L, "_prep", None) is not None) == prep
        tr.close()
    for i, (a, b) in enumerate(zip(res[False][0], res[True][0])):
        assert torch.equal(a, b), i
    for a, b in zip(res[False][1], res[True][1]):
        for k in ("losses/policy_loss", "losses/value_loss", "losses/entropy", "losses/total_loss", "charts/avg_reward"):
            assert a[k] == b[k], k



def test_reference_style_loop_runs_on_the_drop_in_surface():
    """examples/dropin_loop.py is the reference's train.py loop written against VecCarEnv / Buffer / Agent only (plain torch
    for the loss and Adam, torch.randperm minibatches, the reference's loop bounds): it must run unchanged and learn."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("dropin_loop", os.path.join(os.path.dirname(os.path.dirname(__file__)), "examples",
                                                                            "dropin_loop.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    agent, hist = mod.run(TRACKS["big_track"], n_envs=512, n_steps=128, n_epochs=12, batch_size=512, train_iters=20, num_rays=12,
                          seed=1, log=lambda *_: None)
    assert len(hist) == 12 and all(np.isfinite(list(h.values())).all() for h in hist)
    first, last = np.mean([h["avg_reward"] for h in hist[:3]]), np.mean([h["avg_reward"] for h in hist[-3:]])
    assert last > first + 0.01                                   # per-step reward goes up within a dozen epochs
    assert set(agent.state_dict()) == {"actor.0.weight", "actor.0.bias", "actor.2.weight", "actor.2.bias",
                                       "critic.0.weight", "critic.0.bias", "critic.2.weight", "critic.2.bias"}



def _two_rank_worker(rank, world, port, out_dir, use_graphs, exchange="rccl"):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)     # both ranks share cuda:0 here; RCCL needs one GPU per rank
    torch.cuda.set_device(0)
    cfg = _cfg(n_envs=512, n_steps=64, batch_size=32, train_iters=2, use_graphs=use_graphs, seed=11, exchange=exchange,
               capture_collectives=exchange == "p2p" and use_graphs)
    tr = Trainer(cfg, device="cuda:0", rank=rank, world_size=world)
    s1 = tr.run_epoch()
    s2 = tr.run_epoch()
    torch.cuda.synchronize()
    torch.save({"param": tr.learner.flat_param.cpu(), "acts": tr.buffer.act_buf.cpu(), "scalars": s2, "step": tr.global_step_idx,
                "captured": tr.learner._epoch_graph is not None},
               os.path.join(out_dir, f"r{rank}_{int(use_graphs)}_{exchange}.pt"))
    tr.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("use_graphs", [True, False])
def test_two_ranks_on_one_gpu_keep_replicas_identical(tmp_path, use_graphs):
    """world_size 2 with the real kernels (both ranks on cuda:0, gloo as the transport): different env shards and
    action streams per rank, ONE all-reduce of the flat gradient bucket per minibatch (between the two captured
    graphs in graph mode), identical parameters afterwards."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_two_rank_worker, args=(2, port, str(tmp_path), use_graphs), nprocs=2, join=True)
    r0 = torch.load(tmp_path / f"r0_{int(use_graphs)}_rccl.pt")
    r1 = torch.load(tmp_path / f"r1_{int(use_graphs)}_rccl.pt")
    assert torch.equal(r0["param"], r1["param"])                 # replicas bit-identical after 2 epochs
    assert not torch.equal(r0["acts"], r1["acts"])               # but the shards sampled different actions
    assert r0["step"] == r1["step"] == 2 * 2 * 512 * 64          # global_step counts the whole job (train.py:174)
    assert r0["scalars"]["charts/avg_reward"] == pytest.approx(r1["scalars"]["charts/avg_reward"])   # all-reduced scalars


@pytest.mark.parametrize("use_graphs", [True, False])
def test_one_shot_p2p_exchange_between_two_ranks_on_one_gpu(tmp_path, use_graphs):
    """PPOConfig.exchange = "p2p": the per-minibatch gradient all-reduce as the library's one-shot exchange over hipIpc-mapped
    staging buffers (pc_xchg_*), two processes sharing cuda:0 (an IPC mapping works between processes on one device, so
    correctness and bit-identity are provable here; the xGMI latency is not).  Replicas stay bit-identical, the result equals
    the all_reduce path's bit for bit (two ranks: a + b is the same sum in either order), and with graphs the whole epoch's
    update -- exchanges included -- can be ONE captured graph (capture_collectives) even over gloo.  (train.py:259-260)"""
    import socket
    import torch.multiprocessing as mp
    res = {}
    for exchange in ("p2p", "rccl"):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        try:
            mp.spawn(_two_rank_worker, args=(2, port, str(tmp_path), use_graphs, exchange), nprocs=2, join=True)
        except Exception as ex:      # noqa: BLE001
            # Two processes replaying whole-epoch graphs on ONE device time-slice it; an exchange kernel that spins for its peer can
            # (rarely) be left alone on the device until its patience runs out: ExchangeTimeout, raised by run_epoch, the grid drains,
            # nothing hangs.  That is a property of this same-device rehearsal (the product runs one process per GPU), so the captured
            # case gets ONE more attempt in fresh processes -- and fails if that times out as well.  Anything else fails at once.
            if not (use_graphs and exchange == "p2p" and "ExchangeTimeout" in str(ex)):
                raise
            with socket.socket() as s2:
                s2.bind(("127.0.0.1", 0))
                port = s2.getsockname()[1]
            mp.spawn(_two_rank_worker, args=(2, port, str(tmp_path), use_graphs, exchange), nprocs=2, join=True)
        res[exchange] = [torch.load(tmp_path / f"r{r}_{int(use_graphs)}_{exchange}.pt") for r in (0, 1)]
    p0, p1 = res["p2p"]
    assert torch.equal(p0["param"], p1["param"])                          # replicas bit-identical
    assert not torch.equal(p0["acts"], p1["acts"])
    assert torch.equal(p0["param"], res["rccl"][0]["param"])              # and the very bits of the all_reduce path
    assert p0["captured"] == use_graphs and not res["rccl"][0]["captured"]


def _xchg_worker(rank, world, port, out_dir, n, epochs):
    """Drives pc_xchg_allreduce DIRECTLY (no trainer, no graphs): `epochs` exchanges of a bucket whose contents depend on (rank,
    epoch, index); after each one the bucket must be the rank-ordered sum, bit for bit, and the handle must report PC_OK."""
    import ctypes as C
    import os
    import torch.distributed as dist
    from ppo_car_amd._capi import check, lib
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    h = C.c_void_p()
    check(lib.pc_xchg_create(0, rank, world, n, C.byref(h)), "pc_xchg_create")
    check(lib.pc_xchg_set_timeout(h, 60.0), "pc_xchg_set_timeout")
    assert lib.pc_xchg_set_timeout(h, -1.0) != 0
    from ppo_car_amd._capi import PC_XCHG_HANDLE_BYTES as HB
    mine = (C.c_char * HB)()
    check(lib.pc_xchg_local_handle(h, mine), "pc_xchg_local_handle")
    assert HB == 128 and b":" in bytes(mine.raw)[64:]          # the IPC handle, then the PCI bus id of the buffer's device (text)
    handles = [None] * world
    dist.all_gather_object(handles, bytes(mine.raw))
    if world > 1:
        # a peer whose device this process cannot resolve (here: a PCI bus id no device has) is refused AT CONNECT TIME with its own
        # message, nothing stays mapped, no HIP error stays behind -- and the handle then connects normally
        bogus = list(handles)
        peer = (rank + 1) % world
        bogus[peer] = bogus[peer][:64] + b"ffff:ff:1f.7".ljust(HB - 64, b"\0")
        assert lib.pc_xchg_connect(h, C.c_char_p(b"".join(bogus))) == -5          # PC_ERR_UNSUPPORTED
        msg = lib.pc_last_hip_error().decode()
        assert "not visible" in msg and "ffff:ff:1f.7" in msg and "rccl" in msg, msg
        assert lib.pc_xchg_allreduce(h, torch.zeros(n, device="cuda").data_ptr(), None) == -1   # still unconnected: refused, not launched
        torch.zeros(4, device="cuda").sum().item()                                  # (a HIP call after the failure: no stale error)
    check(lib.pc_xchg_connect(h, C.c_char_p(b"".join(handles))), "pc_xchg_connect")
    assert lib.pc_last_hip_error().decode() == ""
    dist.barrier()
    idx = torch.arange(n, device="cuda", dtype=torch.float32)
    gen = lambda r, ep: torch.sin(idx * (0.37 + r) + ep * 1.7) * (1.0 + 1000.0 * (ep % 3)) + r       # irregular magnitudes
    bad = 0
    st = torch.cuda.current_stream().cuda_stream
    for ep in range(epochs):
        bucket = gen(rank, ep)
        check(lib.pc_xchg_allreduce(h, bucket.data_ptr(), st), "pc_xchg_allreduce")
        want = gen(0, ep)
        for r in range(1, world):
            want = want + gen(r, ep)           # rank order, float32: what every rank must hold
        bad += int((bucket != want).sum())
        if ep % 50 == 49:
            assert lib.pc_xchg_status(h) == 0
    assert lib.pc_xchg_status(h) == 0          # PC_OK: no wait ever gave up
    torch.save({"bad": bad}, os.path.join(out_dir, f"x{rank}.pt"))
    torch.cuda.synchronize()
    dist.barrier()
    lib.pc_xchg_destroy(h)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2, 4])
def test_one_shot_exchange_kernel_directly_many_epochs(tmp_path, world):
    """pc_xchg_allreduce on its own, 300 back-to-back exchanges (both epoch parities, the double buffering, 15 independent chunks
    for a 14858-float bucket): bit-equal rank-ordered sums on every rank and PC_OK -- on one rank (the sum of one bucket is the
    bucket) and between 2 and 4 processes sharing cuda:0 (a GPU box of this pool lets one job put six processes on its card, the test
    runner included; the slot / flag layout at the full 8 ranks runs in ONE process: test_one_shot_exchange_eight_ranks_in_one_process).  Plain launches, no whole-epoch graphs: nothing here can starve a peer, so
    a PC_ERR_TIMEOUT in THIS test is a synchronisation bug in the kernel.  (train.py:259-260)"""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_xchg_worker, args=(world, port, str(tmp_path), 14858, 300), nprocs=world, join=True)
    for r in range(world):
        assert torch.load(tmp_path / f"x{r}.pt")["bad"] == 0


@pytest.mark.parametrize("world,n", [(8, 14858), (8, 23050), (5, 12298), (3, 1000)])
def test_one_shot_exchange_eight_ranks_in_one_process(world, n):
    """K13's slot / flag layout at the FULL world size (XCHG_MAX_RANKS = 8: 2 parities x 8 writers x n_pad floats and 2 x 8 x n_chunks
    flags per rank) -- which no multi-process rehearsal on a one-GPU box can reach (at most six processes per card): all `world` ranks'
    handles live in this process (pc_xchg_connect_local) and their exchange kernels go out as ONE launch (pc_xchg_allreduce_group:
    blockIdx.y = rank), so the grids are co-resident and wait for each other exactly as eight devices would.  100 exchanges, buckets of irregular magnitude: every
    rank's bucket must be the rank-ordered float32 sum, bit for bit, and every handle PC_OK.  (train.py:259-260)"""
    import ctypes as C
    from ppo_car_amd._capi import check, lib
    hs = []
    for r in range(world):
        h = C.c_void_p()
        check(lib.pc_xchg_create(0, r, world, n, C.byref(h)), "pc_xchg_create")
        check(lib.pc_xchg_set_timeout(h, 10.0), "pc_xchg_set_timeout")
        hs.append(h)
    arr = (C.c_void_p * world)(*[h.value for h in hs])
    assert lib.pc_xchg_allreduce(hs[1], torch.zeros(n, device="cuda").data_ptr(), None) == -1      # unconnected: refused, not launched
    wrong = (C.c_void_p * world)(*[hs[(r + 1) % world].value for r in range(world)])
    assert lib.pc_xchg_connect_local(hs[0], wrong) == -1                                            # handles out of rank order: refused
    for h in hs:
        check(lib.pc_xchg_connect_local(h, arr), "pc_xchg_connect_local")
    idx = torch.arange(n, device="cuda", dtype=torch.float32)
    gen = lambda r, ep: torch.sin(idx * (0.37 + r) + ep * 1.7) * (1.0 + 1000.0 * (ep % 3)) + r
    torch.cuda.synchronize()
    bad = 0
    st = torch.cuda.current_stream().cuda_stream
    for ep in range(100):
        buckets = [gen(r, ep) for r in range(world)]
        want = buckets[0].clone()
        for r in range(1, world):
            want = want + buckets[r]
        ptrs = (C.c_void_p * world)(*[b.data_ptr() for b in buckets])
        check(lib.pc_xchg_allreduce_group(arr, ptrs, st), "pc_xchg_allreduce_group")      # all ranks' grids in ONE launch: co-resident
        for r in range(world):
            bad += int((buckets[r] != want).sum())
    for h in hs:
        assert lib.pc_xchg_status(h) == 0
    assert bad == 0
    for h in hs:
        lib.pc_xchg_destroy(h)


def test_grouped_exchange_refuses_a_grid_that_cannot_be_resident():
    """A workgroup of the grouped exchange waits for the same chunk's workgroups of the other ranks: world x ceil(n / 1024) workgroups
    must be on the device at once.  8 ranks x 1 M floats (8192 workgroups) is refused with PC_ERR_UNSUPPORTED and a reason, not
    launched into a timeout."""
    import ctypes as C
    from ppo_car_amd._capi import check, lib
    world, n = 8, 1 << 20
    hs = []
    for r in range(world):
        h = C.c_void_p()
        check(lib.pc_xchg_create(0, r, world, n, C.byref(h)), "pc_xchg_create")
        hs.append(h)
    arr = (C.c_void_p * world)(*[h.value for h in hs])
    for h in hs:
        check(lib.pc_xchg_connect_local(h, arr), "pc_xchg_connect_local")
    buckets = [torch.zeros(n, device="cuda") for _ in range(world)]
    ptrs = (C.c_void_p * world)(*[b.data_ptr() for b in buckets])
    rc = lib.pc_xchg_allreduce_group(arr, ptrs, torch.cuda.current_stream().cuda_stream)
    assert rc == -5 and b"co-resident" in lib.pc_last_hip_error()
    for h in hs:
        assert lib.pc_xchg_status(h) == 0
        lib.pc_xchg_destroy(h)


@pytest.mark.parametrize("world", [4])
def test_four_ranks_on_one_gpu_keep_replicas_identical_both_exchanges(tmp_path, world):
    """The trainer at world size 4 (four processes sharing cuda:0, gloo for the rendezvous): env shards and action streams differ per
    rank, the per-minibatch exchange -- torch.distributed's all_reduce and the library's one-shot exchange, eagerly enqueued --
    leaves all four replicas bit-identical, and both exchanges give the same parameters up to the summation order of four terms
    (the one-shot exchange sums in rank order; gloo's ring does not promise one).  (train.py:259-261)"""
    import socket
    import torch.multiprocessing as mp
    res = {}
    for exchange in ("rccl", "p2p"):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        mp.spawn(_two_rank_worker, args=(world, port, str(tmp_path), False, exchange), nprocs=world, join=True)
        res[exchange] = [torch.load(tmp_path / f"r{r}_0_{exchange}.pt") for r in range(world)]
        for r in range(1, world):
            assert torch.equal(res[exchange][0]["param"], res[exchange][r]["param"]), (exchange, r)     # replicas bit-identical
            assert not torch.equal(res[exchange][0]["acts"], res[exchange][r]["acts"])
        assert res[exchange][0]["step"] == 2 * world * 512 * 64
    assert float((res["rccl"][0]["param"] - res["p2p"][0]["param"]).abs().max()) < 1e-5


@pytest.mark.parametrize("B,D,A", [(512, 23, 9), (100, 18, 9), (1024, 39, 9), (256, 23, 6), (64, 39, 13)])
def test_custom_minibatch_kernels_match_torch_autograd(B, D, A):
    """pc_ppo_minibatch (gather + both MLPs forward + loss + backward, no library GEMM, then clip + Adam) against torch:
    the raw gradient against autograd's, then parameters and metric sums over several optimizer steps.  A = 9 runs the kernels
    with CarEnv's action count compiled in, the other counts the generic ones."""
    from ppo_car_amd.ppo import ppo_loss
    M = 5000
    g = torch.Generator().manual_seed(B + D)
    obs = (torch.rand(M, D, generator=g) * 2 - 0.3).cuda()
    act = torch.randint(0, A, (M,), generator=g).float().cuda()
    lp = (-torch.rand(M, generator=g) * 2.5).cuda()
    adv = (torch.randn(M, generator=g) * 3 + 0.5).cuda()
    ret = torch.randn(M, generator=g).cuda()
    idxs = [torch.randperm(M, generator=g)[:B].cuda() for _ in range(5)]
    res = {}
    for custom in (False, True):
        torch.manual_seed(3)
        agent = pc.Agent(D, A).cuda()
        with torch.no_grad():
            for p_ in agent.parameters():
                p_.add_(torch.randn_like(p_) * 0.05)
        cfg = PPOConfig(n_envs=8, n_steps=B, batch_size=B, train_iters=1, use_graphs=False, fused_update=custom, custom_mlp=custom,
                        max_grad_norm=0.5)
        L = PPOLearner(agent, cfg, "cuda")
        assert L.custom == custom
        if custom:   # raw (unclipped) gradient of the first minibatch: apply = 0
            from ppo_car_amd._capi import check, lib
            i = idxs[0]
            check(lib.pc_ppo_minibatch(0, i.data_ptr(), B, D, 256, A, obs.data_ptr(), act.data_ptr(), lp.data_ptr(), adv.data_ptr(),
                                       ret.data_ptr(), L.flat_param.data_ptr(), L.flat_grad.data_ptr(), None, None, None, None, 0.2,
                                       0.5, 0.001, 0.5, 0.9, 0.999, 1e-5, L.metrics.data_ptr(), L._ws.data_ptr(), 0,
                                       torch.cuda.current_stream().cuda_stream), "pc_ppo_minibatch")
            raw = L.flat_grad.clone()
            L.metrics.zero_()
        else:
            i = idxs[0]
            loss, *_ = ppo_loss(agent, obs[i], act[i], lp[i], adv[i], ret[i], 0.2, 0.5, 0.001)
            L.flat_grad.zero_()
            loss.backward()
            raw = L.flat_grad.clone()
        for i in idxs:
            if custom:
                L.custom_minibatch_step(i, obs, act, lp, adv, ret)
            else:
                L.minibatch_step(obs[i], act[i], lp[i], adv[i], ret[i])
        torch.cuda.synchronize()
        res[custom] = (raw, L.flat_param.clone(), L.metrics.clone())
    (g0, p0, m0), (g1, p1, m1) = res[False], res[True]
    assert torch.allclose(g0, g1, atol=2e-6, rtol=2e-4), float((g0 - g1).abs().max())
    assert torch.allclose(m0, m1, atol=2e-4, rtol=1e-5)
    assert float((p0 - p1).abs().max()) < 1e-5


@pytest.mark.parametrize("n,world", [(14858, 4), (300, 1), (70001, 8)])
def test_multi_rank_clip_adam_kernel_matches_torch(n, world):
    """pc_clip_adam_advanced (the multi-rank step after the all-reduce: bucket = sum over ranks, step counter already advanced)
    against clip_grad_norm_ + torch.optim.Adam on the averaged gradient (train.py:260-261), several steps, big and small norms."""
    from ppo_car_amd._capi import check, lib
    g = torch.Generator().manual_seed(n)
    p0 = torch.randn(n, generator=g)
    ref_p = torch.nn.Parameter(p0.clone().cuda())
    opt = torch.optim.Adam([ref_p], lr=2.5e-4, eps=1e-5)
    param, m, v = p0.clone().cuda(), torch.zeros(n).cuda(), torch.zeros(n).cuda()
    step, lr = torch.zeros(1).cuda(), torch.full((1,), 2.5e-4).cuda()
    for it in range(6):
        bucket = (torch.randn(n, generator=g) * (0.001 if it % 2 else 1.0) * world).cuda()       # the sum over `world` ranks
        before = bucket.clone()
        ref_p.grad = bucket / world
        torch.nn.utils.clip_grad_norm_([ref_p], 0.5)
        opt.step()
        step += 1                                                                               # what K11 does under apply = 2
        check(lib.pc_clip_adam_advanced(0, param.data_ptr(), bucket.data_ptr(), m.data_ptr(), v.data_ptr(), step.data_ptr(),
                                        lr.data_ptr(), n, 0.5, 1.0 / world, 0.9, 0.999, 1e-5,
                                        torch.cuda.current_stream().cuda_stream), "pc_clip_adam_advanced")
        assert torch.equal(bucket, before)                                                      # the bucket is left as delivered
    torch.cuda.synchronize()
    assert float((param - ref_p.detach()).abs().max()) < 2e-6
    st = opt.state[ref_p]
    assert torch.allclose(m, st["exp_avg"], atol=1e-7, rtol=1e-5) and torch.allclose(v, st["exp_avg_sq"], atol=1e-9, rtol=1e-5)
    assert lib.pc_clip_adam_advanced(0, None, None, None, None, None, None, n, 0.5, 1.0, 0.9, 0.999, 1e-5, None) != 0


def test_checkpoint_resume_continues_bit_for_bit(tmp_path):
    """3 epochs in one go == 2 epochs, save, fresh trainer, load, 1 more epoch (policy, optimizer state, env state,
    device Philox counters and the host index generator all restored)."""
    cfg = _cfg(n_envs=512, n_steps=48, batch_size=64, train_iters=3)
    a = Trainer(cfg, device="cuda")
    for _ in range(3):
        a.run_epoch(sync=False)
    torch.cuda.synchronize()
    b = Trainer(cfg, device="cuda")
    for _ in range(2):
        b.run_epoch(sync=False)
    torch.save(b.state_dict(), tmp_path / "t.pt")
    b.close()
    c = Trainer(cfg, device="cuda")
    c.load_state_dict(torch.load(tmp_path / "t.pt", map_location="cuda", weights_only=False))
    assert c.epoch == 2 and c.global_step_idx == 2 * 512 * 48
    c.run_epoch(sync=False)
    torch.cuda.synchronize()
    assert torch.equal(a.learner.flat_param, c.learner.flat_param)
    assert torch.equal(a.buffer.act_buf, c.buffer.act_buf) and torch.equal(a.next_obs, c.next_obs)
    sa, sc = a.envs.get_state(), c.envs.get_state()
    for k in sa:
        assert np.array_equal(sa[k], sc[k]), k
    a.close(); c.close()


def test_evaluate_entry_point(tmp_path):
    import evaluate
    torch.manual_seed(0)
    agent = pc.Agent(18, 9)
    torch.save(agent.state_dict(), tmp_path / "model.dat")
    out = evaluate.main(["--checkpoint", str(tmp_path / "model.dat"), "--track", TRACKS["big_track"], "--episodes", "4"])
    assert out["episodes"] == 4 and 1 <= out["mean_steps"] <= 1000 and len(out["returns"]) == 4
    # frame dump: PNG frames of episode 0 from the software rasteriser
    out = evaluate.main(["--checkpoint", str(tmp_path / "model.dat"), "--track", TRACKS["big_track"], "--episodes", "2",
                         "--frames", str(tmp_path / "frames"), "--frame-every", "3"])
    files = sorted((tmp_path / "frames").glob("frame_*.png"))
    assert out["frames"] == len(files) >= 1
    head = files[0].read_bytes()[:24]
    assert head[:8] == b"\x89PNG\r\n\x1a\n" and int.from_bytes(head[16:20], "big") == 640 and int.from_bytes(head[20:24], "big") == 360


def test_bootstrap_value_switch_takes_the_reference_call_in_fp32():
    """PPOConfig.bootstrap_value = "fp32": Buffer.calculate_advantages gets agent.get_value(next_obs) (train.py:200) from torch's
    fp32 Linear instead of the rollout launch's own critic pass; the two agree within the fused policy step's tolerance, and the
    advantages that come out differ by no more than that."""
    outs = {}
    for mode in ("kernel", "fp32"):
        tr = Trainer(_cfg(n_envs=1024, n_steps=32, bootstrap_value=mode, rollout_kernel="mega"), device="cuda")
        tr.rollout()
        assert tr.rollout_mode == "mega" and tr._aux_valid
        with torch.no_grad():
            ref = tr.agent.get_value(tr.next_obs).reshape(-1)
            in_kernel = tr._aux_valid and tr.cfg.bootstrap_value == "kernel"
            nv = (tr._boot_val if in_kernel else ref).reshape(1, -1).clone()
            adv, _ = tr.buffer.calculate_advantages(nv, tr.next_term.reshape(1, -1), tr.next_trunc.reshape(1, -1))
        outs[mode] = (nv.clone(), adv.clone(), ref.clone())
        tr.buffer.ptr = tr.cfg.n_steps
        tr.update()              # runs with the configured bootstrap
        tr.close()
    assert torch.equal(outs["fp32"][0].reshape(-1), outs["fp32"][2])             # the reference's call, bit for bit
    assert not torch.equal(outs["kernel"][0], outs["fp32"][0])
    assert float((outs["kernel"][0] - outs["fp32"][0]).abs().max()) <= 1e-5
    assert float((outs["kernel"][1] - outs["fp32"][1]).abs().max()) <= 1e-5
    with pytest.raises(ValueError):
        Trainer(_cfg(bootstrap_value="bf16"), device="cuda")


@pytest.mark.parametrize("num_rays,batch,graphs", [(16, 512, True), (12, 100, False), (32, 64, False)])
def test_deferred_adam_epoch_chain_is_bitwise_the_three_launch_steps(num_rays, batch, graphs):
    """pc_ppo_epoch_prepared (two launches per minibatch: the clip + Adam step of minibatch i rides in the forward / backward launch
    of minibatch i + 1, state ping-ponged between two buffers) against pc_ppo_minibatch_prepared(apply = 1) per minibatch (three
    launches): identical parameters, Adam state, step counter, logged sums and last clipped gradient over three epochs
    (train.py:223-269)."""
    res = {}
    for deferred in (False, True):
        cfg = _cfg(deferred_adam=deferred, use_graphs=graphs, n_envs=256, n_steps=64, batch_size=batch, train_iters=5,
                   num_rays=num_rays, seed=9)
        tr = Trainer(cfg, device="cuda")
        scal = [tr.run_epoch() for _ in range(3)]
        L = tr.learner
        res[deferred] = ([t.clone() for t in (L.flat_param, L.exp_avg, L.exp_avg_sq, L.step_count, L.metrics, L.flat_grad)], scal)
        assert (getattr(L, "_state2", None) is not None) == deferred
        tr.close()
    for i, (a, b) in enumerate(zip(res[False][0], res[True][0])):
        assert torch.equal(a, b), i
    assert float(res[True][0][3]) == 3 * 5 * len(range(0, 64, batch))
    for a, b in zip(res[False][1], res[True][1]):
        for k in ("losses/policy_loss", "losses/value_loss", "losses/entropy", "losses/total_loss", "charts/avg_reward"):
            assert a[k] == b[k], k


def test_lazy_scalars_are_the_synchronous_scalars_one_epoch_late():
    """Trainer.run_epoch(sync="lazy") (train.py --lazy-logging): the epoch is only enqueued, the call hands back the scalars of the
    epoch before -- the same numbers the synchronous call returns, the learning rate and the policy-range word riding along -- and
    flush_scalars() the last one."""
    keys = ("losses/policy_loss", "losses/value_loss", "losses/entropy", "losses/total_loss", "charts/avg_reward", "charts/learning_rate", "global_step")
    a, b = Trainer(_cfg(n_envs=512, n_steps=64), device="cuda"), Trainer(_cfg(n_envs=512, n_steps=64), device="cuda")
    want = [a.run_epoch(sync=True) for _ in range(6)]
    got = [b.run_epoch(sync="lazy") for _ in range(6)]
    assert got[0] is None
    got = got[1:] + [b.flush_scalars()]
    assert b.flush_scalars() is None
    for w, g_ in zip(want, got):
        for k in keys:
            assert w[k] == pytest.approx(g_[k], rel=1e-6, abs=1e-9), k
    # a synchronous call after lazy ones waits for the pending epoch instead of dropping it
    b.run_epoch(sync="lazy")
    s7 = b.run_epoch(sync=True)
    assert s7["global_step"] == 8 * 512 * 64 and b.flush_scalars() is None
    a.close()
    b.close()
