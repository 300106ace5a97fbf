"""GPU tests of the GAE(lambda) scan kernel (Buffer.calculate_advantages) and the fused categorical
sampling kernel (Agent.act)."""
import numpy as np
import pytest
import torch

import oracle
import ppo_car_amd as pc
from conftest import GOLDEN, TRACKS

pytestmark = pytest.mark.gpu


def _fill(buf, rew, val, term, trunc):
    T, N = rew.shape
    D = buf.obs_buf.shape[-1]
    for t in range(T):
        buf.store(torch.zeros(N, D, device="cuda"), torch.zeros(N, device="cuda"), torch.from_numpy(rew[t]).cuda(),
                  torch.from_numpy(val[t]).cuda(), torch.from_numpy(term[t]).cuda(), torch.from_numpy(trunc[t]).cuda(),
                  torch.zeros(N, device="cuda"))


def test_gae_matches_reference_golden_bit_exact():
    g = np.load(f"{GOLDEN}/gae_cases.npz")
    for c in range(int(g["n_cases"])):
        rew, val = g[f"c{c}_rew"], g[f"c{c}_val"]
        T, N = rew.shape
        buf = pc.Buffer((2,), T, N, "cuda", float(g["gamma"]), float(g["lam"]))
        _fill(buf, rew, val, g[f"c{c}_term"], g[f"c{c}_trunc"])
        adv, ret = buf.calculate_advantages(torch.from_numpy(g[f"c{c}_last_val"]).cuda(),
                                            torch.from_numpy(g[f"c{c}_last_term"]).cuda(),
                                            torch.from_numpy(g[f"c{c}_last_trunc"]).cuda())
        assert adv.shape == (T, N)
        assert np.array_equal(adv.cpu().numpy(), g[f"c{c}_adv"]), c      # bit-exact float32
        assert np.array_equal(ret.cpu().numpy(), g[f"c{c}_ret"]), c


@pytest.mark.parametrize("T,N", [(128, 65536), (1024, 4096), (1000, 777), (7, 33)])
def test_gae_full_sizes_vs_oracle(T, N):
    """BASELINE sizes ([T,N] = [128,65536] and [1024,4096]) and ragged ones, against the C oracle."""
    rng = np.random.default_rng(T + N)
    rew = (rng.standard_normal((T, N)) * 0.3).astype(np.float32)
    val = rng.standard_normal((T, N)).astype(np.float32)
    term = (rng.random((T, N)) < 0.015).astype(np.float32)
    trunc = ((rng.random((T, N)) < 0.002) * (1 - term)).astype(np.float32)
    lv = rng.standard_normal(N).astype(np.float32)
    lt = (rng.random(N) < 0.3).astype(np.float32)
    ltr = ((rng.random(N) < 0.3) * (1 - lt)).astype(np.float32)
    A, Rt = oracle.gae(rew, val, term, trunc, lv, lt, ltr, 0.99, 0.95)
    buf = pc.Buffer((1,), T, N, "cuda", 0.99, 0.95)
    buf.rew_buf.copy_(torch.from_numpy(rew)); buf.val_buf.copy_(torch.from_numpy(val))
    buf.term_buf.copy_(torch.from_numpy(term)); buf.trunc_buf.copy_(torch.from_numpy(trunc))
    buf.ptr = T
    adv, ret = buf.calculate_advantages(torch.from_numpy(lv).cuda().reshape(1, -1), torch.from_numpy(lt).cuda().reshape(1, -1),
                                        torch.from_numpy(ltr).cuda().reshape(1, -1))
    assert np.array_equal(adv.cpu().numpy(), A) and np.array_equal(ret.cpu().numpy(), Rt)
    # property: ret - adv == val exactly as float32 adds (buffer.py:63)
    assert torch.equal(ret, adv + buf.val_buf)


def test_gae_equals_torch_expression_on_gpu():
    """the reference's own torch loop (buffer.py:51-63) run on the GPU gives the same bits"""
    T, N = 64, 512
    g = torch.Generator(device="cuda").manual_seed(0)
    rew = torch.randn(T, N, device="cuda", generator=g)
    val = torch.randn(T, N, device="cuda", generator=g)
    term = (torch.rand(T, N, device="cuda", generator=g) < 0.05).float()
    trunc = (torch.rand(T, N, device="cuda", generator=g) < 0.02).float() * (1 - term)
    lv = torch.randn(1, N, device="cuda", generator=g)
    lt = (torch.rand(1, N, device="cuda", generator=g) < 0.3).float()
    ltr = torch.zeros(1, N, device="cuda")
    buf = pc.Buffer((1,), T, N, "cuda", 0.99, 0.95)
    buf.rew_buf.copy_(rew); buf.val_buf.copy_(val); buf.term_buf.copy_(term); buf.trunc_buf.copy_(trunc)
    buf.ptr = T
    adv, ret = buf.calculate_advantages(lv, lt, ltr)
    ref = torch.zeros_like(rew)
    last_gae = 0.0
    for t in reversed(range(T)):
        nv = lv if t == T - 1 else val[t + 1]
        tm = 1.0 - lt if t == T - 1 else 1.0 - term[t + 1]
        trm = 1.0 - ltr if t == T - 1 else 1.0 - trunc[t + 1]
        delta = rew[t] + 0.99 * nv * tm - val[t]
        last_gae = delta + 0.99 * 0.95 * tm * trm * last_gae
        ref[t] = last_gae
    assert torch.equal(adv, ref.reshape(T, N)) and torch.equal(ret, ref.reshape(T, N) + val)


def test_buffer_asserts_like_the_reference():
    buf = pc.Buffer((3,), 2, 4, "cuda")
    with pytest.raises(AssertionError, match="Buffer not full"):
        buf.calculate_advantages(torch.zeros(1, 4), torch.zeros(1, 4), torch.zeros(1, 4))


def test_sample_kernel_distribution_logprob_entropy():
    N, A = 200000, 9
    g = torch.Generator(device="cuda").manual_seed(1)
    base = torch.randn(1, A, device="cuda", generator=g) * 1.5
    logits = base.expand(N, A).contiguous()
    agent = pc.Agent(23, A).cuda()
    agent.actor = torch.nn.Identity()       # feed logits straight through
    agent.critic = torch.nn.Linear(A, 1).cuda()
    agent.rng_seed = 123
    a1, lp1, v1 = agent.act(logits)
    dist = torch.distributions.Categorical(logits=logits)
    assert a1.dtype == torch.int64 and int(a1.min()) >= 0 and int(a1.max()) < A
    assert torch.allclose(lp1, dist.log_prob(a1), atol=2e-6)
    counts = torch.bincount(a1, minlength=A).double().cpu().numpy()
    p = dist.probs[0].double().cpu().numpy()
    chi2 = ((counts - N * p) ** 2 / (N * p)).sum()
    assert chi2 < 40.0            # 8 dof: P(chi2 > 40) ~ 3e-6
    # a different call counter gives a different draw; the same (seed, offset) the same one
    a2, _, _ = agent.act(logits)
    assert (a1 != a2).float().mean() > 0.3
    agent._rng_offset = 0
    a3, lp3, _ = agent.act(logits)
    assert torch.equal(a1, a3) and torch.equal(lp1, lp3)


def test_sample_kernel_entropy_and_extreme_logits():
    from ppo_car_amd._capi import check, lib
    N, A = 4096, 9
    g = torch.Generator(device="cuda").manual_seed(2)
    logits = torch.randn(N, A, device="cuda", generator=g) * 3
    logits[0] = torch.tensor([100.0] + [-100.0] * 8)          # (near-)deterministic row
    logits[1] = 0.0                                           # uniform row
    act = torch.empty(N, dtype=torch.int64, device="cuda")
    lp = torch.empty(N, device="cuda")
    ent = torch.empty(N, device="cuda")
    check(lib.pc_sample(0, logits.data_ptr(), N, A, 9, 0, act.data_ptr(), lp.data_ptr(), ent.data_ptr(),
                        torch.cuda.current_stream().cuda_stream), "pc_sample")
    dist = torch.distributions.Categorical(logits=logits)
    assert torch.allclose(ent, dist.entropy(), atol=5e-6)
    assert torch.allclose(lp, dist.log_prob(act), atol=5e-6)
    assert int(act[0]) == 0 and abs(float(ent[1]) - np.log(9)) < 1e-6


@pytest.mark.parametrize("policy_precision", [0, 1, 2])
@pytest.mark.parametrize("D", [18, 23, 39])
@pytest.mark.parametrize("N", [1000, 65536])
def test_fused_policy_kernel_matches_torch_mlp(D, N, policy_precision):
    """pc_policy_act (0: fp32 MFMA, 1: bf16 x 3, 2: fp16 x 2 split forms; hidden layer in registers) vs torch's Linear/ReLU/Linear on the same weights:
    logits and values within 1e-5 (different fp32 summation order only), log_prob consistent with the logits it
    reports, draws distributed as the categorical."""
    torch.manual_seed(D)
    agent = pc.Agent(D, 9).cuda()
    agent.policy_precision = policy_precision       # this agent's pc_policy handle (nothing process-wide)
    with torch.no_grad():
        for p in agent.parameters():          # non-trivial biases / output weights (the init has zero biases, 0.01 gain)
            p.add_(torch.randn_like(p) * 0.1)
    g = torch.Generator(device="cuda").manual_seed(N)
    x = torch.rand(N, D, device="cuda", generator=g) * 2 - 0.5
    logits = torch.empty(N, 9, device="cuda")
    af = torch.empty(N, device="cuda")
    agent.rng_seed = 77
    a, lp, v = agent.act(x, out_logits=logits, out_action_f32=af)
    with torch.no_grad():
        ref_logits, ref_v = agent.actor(x), agent.critic(x).view(-1)
    assert agent.policy_form()[0] == policy_precision   # the split forms cover D <= 40 (two K blocks above 24)
    assert torch.allclose(logits, ref_logits, atol=1e-5, rtol=1e-5)
    assert torch.allclose(v, ref_v, atol=1e-5, rtol=1e-5)
    # against float64: all three forms are fp32-class
    with torch.no_grad():
        a1, a2 = agent.actor[0], agent.actor[2]
        ref64 = torch.relu(x.double() @ a1.weight.double().T + a1.bias.double()) @ a2.weight.double().T + a2.bias.double()
    assert float((logits.double() - ref64).abs().max()) < 4e-6
    dist = torch.distributions.Categorical(logits=logits)
    assert torch.allclose(lp, dist.log_prob(a), atol=2e-6)
    assert torch.equal(af, a.float()) and int(a.min()) >= 0 and int(a.max()) <= 8
    # same (seed, offset) through the unfused path: identical uniforms, so identical draws wherever the two
    # paths' logits agree to the last bit of the CDF comparison (almost everywhere)
    agent._rng_offset = 0
    a2, lp2, v2 = agent.act(x, fused=False)
    assert (a2 == a).float().mean() > 0.999
    if N >= 65536:
        p = torch.softmax(ref_logits.double(), -1).mean(0).cpu().numpy()
        counts = torch.bincount(a, minlength=9).double().cpu().numpy()
        assert np.abs(counts / N - p).max() < 0.01


@pytest.mark.parametrize("policy_precision", [2, 1, 0])
def test_fused_policy_kernel_on_trained_weights_and_harvested_observations(policy_precision):
    """The precision claim of the fused policy kernel on REAL data rather than random weights: tests/golden/policy_trained.npz
    holds the weights after 150 PPO epochs on big_track (tools/make_policy_fixture.py, this repository's own run) and ~3900
    observations of its last rollout -- among them 300 reset rows (velocity exactly 0: operands below fp16's normal range in
    every scaled domain).  Logits and values of every arithmetic form must stay within 4e-6 of float64 (north_star: 1e-5)."""
    import os
    from conftest import GOLDEN, TRACKS
    f = np.load(os.path.join(GOLDEN, "policy_trained.npz"))
    obs = f["obs"]
    n = len(obs)
    assert obs.shape == (n, 23) and n >= 3000 and float(f["avg_reward"]) > 0.15             # a trained policy (reference curve: 0.11 -> 0.25)
    assert (np.abs(obs[:, 2:4]).max(1) == 0).sum() > 50 and (np.abs(obs[:, 2:4]).max(1) < 1e-3).sum() > 200
    agent = pc.Agent(23, 9)
    agent.load_state_dict({k: torch.from_numpy(f[k.replace(".", "_")]) for k in agent.state_dict()})
    agent = agent.cuda()
    agent.policy_precision = policy_precision
    x = torch.from_numpy(obs).cuda()
    logits = torch.empty(n, 9, device="cuda")
    agent.rng_seed = 5
    a, lp, v = agent.act(x, out_logits=logits)
    W = {k: f[k.replace(".", "_")].astype(np.float64) for k in agent.state_dict()}
    x64 = obs.astype(np.float64)
    ref_logits = np.maximum(x64 @ W["actor.0.weight"].T + W["actor.0.bias"], 0) @ W["actor.2.weight"].T + W["actor.2.bias"]
    ref_v = (np.maximum(x64 @ W["critic.0.weight"].T + W["critic.0.bias"], 0) @ W["critic.2.weight"].T + W["critic.2.bias"]).reshape(-1)
    err_l = np.abs(logits.cpu().numpy().astype(np.float64) - ref_logits).max()
    err_v = np.abs(v.cpu().numpy().astype(np.float64) - ref_v).max()
    print(f"precision form {policy_precision}: max |logit err| {err_l:.2e}, max |value err| {err_v:.2e}, max |logit| {np.abs(ref_logits).max():.2f}, "
          f"max |value| {np.abs(ref_v).max():.2f}")
    assert err_l < 4e-6 and err_v < 4e-6
    lp_ref = ref_logits - np.log(np.exp(ref_logits - ref_logits.max(1, keepdims=True)).sum(1, keepdims=True)) - ref_logits.max(1, keepdims=True)
    assert np.abs(lp.cpu().numpy() - lp_ref[np.arange(n), a.cpu().numpy()]).max() < 4e-6


def _float64_policy(f, obs, scale):
    W = {k: f[k].astype(np.float64) for k in f.files if k != "obs" and k != "avg_reward"}
    x64 = obs.astype(np.float64)
    lg = np.maximum(x64 @ (scale * W["actor_0_weight"]).T + W["actor_0_bias"], 0) @ W["actor_2_weight"].T + W["actor_2_bias"]
    v = (np.maximum(x64 @ (scale * W["critic_0_weight"]).T + W["critic_0_bias"], 0) @ W["critic_2_weight"].T + W["critic_2_bias"]).reshape(-1)
    return lg, v


@pytest.mark.parametrize("scale,expect_mask", [(1.0, 0), (30.0, 4), (300.0, 4), (6000.0, 5)])
def test_fp16x2_domain_guard_falls_back_to_the_fp32_chain(scale, expect_mask, capfd):
    """The fp16 x 2 form's operands saturate at fp16's range in their scaled domains (policy.hpp: |W1| > 4094, |W2| > 1023.5, hidden
    activations beyond 255.87) -- silently, until this round.  pc_policy_pack_checked reports, while it packs, whether the weights can
    get there (include/ppocar.h PC_POLICY_RANGE_*); Agent.pack_policy reads the status without synchronising and switches to
    precision 0.  Here: the trained fixture's first-layer weights x 1 (inside the domain: status 0, stays fp16x2), x 30 / x 300 (hidden
    units can pass 255.87: bit 2), x 6000 (a weight beyond 4094 as well: bits 0 and 2).  After the fallback the logits are model.py:34-41
    in float64 to 4e-6 RELATIVE to their magnitude (x 300 makes the logits hundreds); WITHOUT the guard (policy_range untouched, the
    status ignored) the x 300 image is wrong by more than 1 %: the failure the guard exists for."""
    import ctypes as C
    import os
    from conftest import GOLDEN
    from ppo_car_amd._capi import check, lib
    f = np.load(os.path.join(GOLDEN, "policy_trained.npz"))
    obs = f["obs"]
    n = len(obs)
    agent = pc.Agent(23, 9)
    sd = {k: torch.from_numpy(f[k.replace(".", "_")]).clone() for k in agent.state_dict()}
    sd["actor.0.weight"] *= scale
    sd["critic.0.weight"] *= scale
    agent.load_state_dict(sd)
    agent = agent.cuda()
    agent.rng_seed = 5
    x = torch.from_numpy(obs).cuda()
    ref_l, ref_v = _float64_policy(f, obs, scale)
    mag = max(1.0, float(np.abs(ref_l).max()), float(np.abs(ref_v).max()))
    # the raw status word through the C-ABI
    assert agent.pack_policy() and agent.policy_form()[0] == 2
    assert int(agent._range_dev.item()) == expect_mask
    # what the fp16x2 image gives when nobody looks at the status
    logits = torch.empty(n, 9, device="cuda")
    a, lp, v = agent.act(x, out_logits=logits, repack=False)
    err_unguarded = float(np.abs(logits.cpu().numpy().astype(np.float64) - ref_l).max()) / mag
    # the guard: the status of that pack is looked at (here: now), the agent switches form, the next pack / act is the fp32 chain
    ok = agent.check_policy_range(sync=True)
    assert ok == (expect_mask == 0)
    a, lp, v = agent.act(x, out_logits=logits)
    form = agent.policy_form()[0]
    assert form == (2 if expect_mask == 0 else 0)
    err_l = float(np.abs(logits.cpu().numpy().astype(np.float64) - ref_l).max()) / mag
    err_v = float(np.abs(v.cpu().numpy().astype(np.float64) - ref_v).max()) / mag
    msg = capfd.readouterr().err
    print(f"W1 x {scale}: status {expect_mask}, form after the check {form}, |logit|max {np.abs(ref_l).max():.1f}; relative error unguarded {err_unguarded:.2e}, guarded {err_l:.2e} / {err_v:.2e}")
    assert err_l < 4e-6 and err_v < 4e-6
    if expect_mask:
        assert "numeric domain" in msg and "precision 0" in msg
    else:
        assert msg == "" and err_unguarded < 4e-6
    if scale == 300.0:
        assert err_unguarded > 1e-2
    # "raise" instead of the fallback
    if expect_mask:
        agent2 = pc.Agent(23, 9)
        agent2.load_state_dict(sd)
        agent2 = agent2.cuda()
        agent2.policy_range = "raise"
        assert agent2.pack_policy()
        with pytest.raises(pc.PolicyRangeError):
            agent2.check_policy_range(sync=True)


def test_trainer_switches_arithmetic_when_the_weights_leave_the_fp16x2_domain(capfd):
    """End to end: a trainer whose first-layer weights are blown up between two epochs keeps running -- the persistent rollout kernel in
    precision 0 from the next rollout on -- and says so; the rollout after the switch equals a precision-0 trainer's bit for bit."""
    from conftest import TRACKS
    from ppo_car_amd.ppo import PPOConfig, Trainer
    cfg = PPOConfig(n_envs=1024, n_steps=32, batch_size=64, train_iters=1, num_rays=16, track=TRACKS["big_track"], seed=2, learning_rate=0.0)
    tr = Trainer(cfg, device="cuda")
    tr.run_epoch()
    assert tr.agent.policy_form()[0] == 2
    with torch.no_grad():
        tr.agent.actor[0].weight.mul_(400.0)
    tr.run_epoch()                   # this rollout still ran on the saturating image; run_epoch's sync point sees the status
    assert tr.agent.policy_form()[0] == 0 and "numeric domain" in capfd.readouterr().err
    import copy
    st = copy.deepcopy(tr.state_dict())      # (state_dict hands out the live tensors: the rollout below advances them)
    tr.rollout()
    torch.cuda.synchronize()
    assert tr.rollout_mode == "mega"
    got = [t.clone() for t in (tr.buffer.obs_buf, tr.buffer.act_buf, tr.buffer.logprob_buf, tr.buffer.val_buf)]
    tr.close()
    import dataclasses
    tr0 = Trainer(dataclasses.replace(cfg, policy_precision=0), device="cuda")
    tr0.load_state_dict(st)
    tr0.rollout()
    torch.cuda.synchronize()
    for a, b in zip(got, (tr0.buffer.obs_buf, tr0.buffer.act_buf, tr0.buffer.logprob_buf, tr0.buffer.val_buf)):
        assert torch.equal(a, b)
    tr0.close()


def test_two_policy_handles_with_different_arithmetic_coexist():
    """Every launch option lives in a handle (include/ppocar.h; the library keeps no process-wide setting): two agents with
    different arithmetic forms of the fused policy step, used ALTERNATELY in one process, each produce exactly what a fresh agent of
    that form produces alone -- and the two forms differ from each other (fp16x2 vs the exact fp32 chain).  Same for pc_rollout's
    per-env options.  (model.py:34-41, train.py:173-195)"""
    from ppo_car_amd.model import Agent
    torch.manual_seed(3)
    N, D = 4096, 23
    obs = torch.randn(N, D, device="cuda") * 0.5

    def make(prec):
        torch.manual_seed(5)
        a = Agent(D, 9).cuda()
        a.rng_seed = 77
        a.policy_precision = prec
        return a

    alone = {}
    for prec in (2, 0):          # each alone
        b = make(prec)
        logits = torch.empty(N, 9, device="cuda")
        _, lp, v = b.act(obs, out_logits=logits, offset=0)
        alone[prec] = (logits.clone(), lp.clone(), v.clone())
        del b
    agents = {prec: make(prec) for prec in (2, 0)}
    default = make(-1)           # -1 = the library's default form
    assert default.policy_form()[0] == 2
    for _ in range(2):           # interleaved
        for prec in (0, 2):
            logits = torch.empty(N, 9, device="cuda")
            _, lp, v = agents[prec].act(obs, out_logits=logits, offset=0)
            assert agents[prec].policy_form()[0] == prec
            for got, want in zip((logits, lp, v), alone[prec]):
                assert torch.equal(got, want), prec
    logits = torch.empty(N, 9, device="cuda")
    default.act(obs, out_logits=logits, offset=0)
    assert torch.equal(logits, alone[2][0])
    assert not torch.equal(alone[0][0], alone[2][0])          # the two forms do differ in the last bits
    assert float((alone[0][0] - alone[2][0]).abs().max()) < 1e-5

    # pc_rollout's options: per env handle
    e1 = pc.VecCarEnv(64, TRACKS["big_track"], num_rays=16)
    e2 = pc.VecCarEnv(64, TRACKS["big_track"], num_rays=16)
    e1.set_option("rollout_form", 3)
    e1.set_option("rollout_epw", 32)
    assert (e1.get_option("rollout_form"), e1.get_option("rollout_epw")) == (3, 32)
    assert (e2.get_option("rollout_form"), e2.get_option("rollout_epw"), e2.get_option("rollout_fast")) == (-1, 0, 1)
    with pytest.raises(pc.PpoCarError):
        e1.set_option("rollout_epw", 48)
    e1.close(); e2.close()
