"""SURVEY 8(c) fixture 6: one seeded PPO minibatch recorded from the REFERENCE's own classes (tests/golden/make_golden.py ppo:
lib/model.py's `Agent` imported unmodified, train.py:233-261 restated line by line on it, torch.optim.Adam as train.py:146
configures it) -> logits, log-probs, entropy, value, the three loss terms, the total, every gradient, the pre-clip gradient
norm and the parameters after the optimizer step.  Here the host mirror (ppo_car_amd.model.Agent, ppo.ppo_loss) and -- on the
GPU, through the C-ABI -- the policy kernel K5 and the minibatch kernels K10-K12 are held to those recorded numbers."""
import os

import numpy as np
import pytest
import torch

import ppo_car_amd as pc
from ppo_car_amd.ppo import PPOConfig, PPOLearner, ppo_loss

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ppo_minibatch.npz"))
CASES = list(range(int(G["n_cases"])))
PARAMS = ["actor.0.weight", "actor.0.bias", "actor.2.weight", "actor.2.bias", "critic.0.weight", "critic.0.bias", "critic.2.weight",
          "critic.2.bias"]     # module.parameters() order = the flat buffers' layout (include/ppocar.h, pc_ppo_minibatch)


def _case(ci):
    c = {k[len(f"c{ci}_"):]: G[k] for k in G.files if k.startswith(f"c{ci}_")}
    c["D"], c["B"] = c["obs"].shape[1], len(c["idx"])
    return c


def _agent(c, device="cpu", which="w_"):
    a = pc.Agent(c["D"], 9)
    a.load_state_dict({k: torch.from_numpy(c[which + k]) for k in PARAMS})
    return a.to(device)


def _flat(c, which):
    return np.concatenate([c[which + k].reshape(-1) for k in PARAMS])


@pytest.mark.parametrize("ci", CASES)
def test_host_mirror_reproduces_the_reference_agent_and_loss(ci):
    """ppo_car_amd.model.Agent with the reference Agent's weights gives the reference's logits / log-probs / entropy / value, and
    ppo.ppo_loss the three loss terms, the total and the gradients of train.py:233-259 (same torch, CPU: last-bit slack only)."""
    c = _case(ci)
    agent = _agent(c)
    idx = torch.from_numpy(c["idx"])
    obs, act = torch.from_numpy(c["obs"])[idx], torch.from_numpy(c["act"])[idx]
    with torch.no_grad():
        assert np.allclose(agent.actor(obs).numpy(), c["logits"], rtol=0, atol=1e-6)
    _, lp, ent, val = agent.get_action_and_value(obs, act)
    assert np.allclose(lp.detach().numpy(), c["new_logprob"], rtol=0, atol=1e-6)
    assert np.allclose(ent.detach().numpy(), c["entropies"], rtol=0, atol=1e-6)
    assert np.allclose(val.detach().view(-1).numpy(), c["new_values"], rtol=0, atol=1e-6)
    loss, pl, vl, en = ppo_loss(agent, obs, act, torch.from_numpy(c["old_logprob"])[idx], torch.from_numpy(c["adv"])[idx],
                                torch.from_numpy(c["ret"])[idx], float(G["clip_ratio"]), float(G["vf_coef"]), float(G["ent_coef"]))
    for got, key in ((loss, "loss"), (pl, "policy_loss"), (vl, "value_loss"), (en, "entropy")):
        assert float(got.detach()) == pytest.approx(float(c[key]), rel=2e-6, abs=1e-7), key
    loss.backward()
    g = np.concatenate([p.grad.reshape(-1).numpy() for p in agent.parameters()])
    ref = _flat(c, "g_")
    assert np.allclose(g, ref, rtol=1e-5, atol=1e-7)
    assert float(np.sqrt((g.astype(np.float64) ** 2).sum())) == pytest.approx(float(c["grad_norm"]), rel=1e-5)


def test_fixture_covers_both_sides_of_the_clip_and_of_max_grad_norm():
    norms = [float(_case(ci)["grad_norm"]) for ci in CASES]
    assert min(norms) < float(G["max_grad_norm"]) < max(norms)          # clip_grad_norm_ acts in one case and not in another
    for ci in CASES:
        r = _case(ci)["ratios"]
        assert (r < 0.8).any() and (r > 1.2).any() and ((r > 0.8) & (r < 1.2)).any()


@pytest.mark.gpu
@pytest.mark.parametrize("precision", [2, 1, 0])
@pytest.mark.parametrize("ci", CASES)
def test_policy_kernel_k5_against_the_reference_agent(ci, precision):
    """pc_policy_act (K5: both MLPs on the matrix cores + the draw) on the reference Agent's weights and the fixture's
    observations: logits and values against what lib/model.py's Agent produced (model.py:31-41), the log-prob of the action it
    drew against log_softmax of the REFERENCE logits."""
    c = _case(ci)
    agent = _agent(c, "cuda")
    agent.policy_precision = precision
    idx = torch.from_numpy(c["idx"])
    obs = torch.from_numpy(c["obs"])[idx].cuda().contiguous()
    logits = torch.empty(c["B"], 9, device="cuda")
    action, lp, val = agent.act(obs, out_logits=logits, fused=True, offset=3)
    torch.cuda.synchronize()
    assert agent.policy_form() is not None and agent.policy_form()[0] == precision      # the fused kernel ran, in the form asked for
    assert np.abs(logits.cpu().numpy() - c["logits"]).max() <= 4e-6
    assert np.abs(val.cpu().numpy() - c["new_values"]).max() <= 4e-6
    ref_lp = torch.log_softmax(torch.from_numpy(c["logits"]).double(), -1).gather(1, action.cpu().view(-1, 1)).view(-1)
    assert float((lp.cpu().double() - ref_lp).abs().max()) <= 4e-6
    a = action.cpu().numpy()
    assert a.min() >= 0 and a.max() <= 8


@pytest.mark.gpu
@pytest.mark.parametrize("ci", CASES)
def test_minibatch_kernels_k10_to_k12_against_the_reference_step(ci):
    """pc_ppo_minibatch (K10 forward / loss / backward, K11 reduction, K12 clip + Adam: no library GEMM) on the reference Agent's
    weights and the fixture's minibatch: the metric sums = the reference's three loss terms and total (train.py:263-266), the
    gradient = autograd's (after clip_grad_norm_, train.py:260), the parameters after the step = the reference optimizer's
    (train.py:261, Adam lr 3e-4 eps 1e-5)."""
    c = _case(ci)
    agent = _agent(c, "cuda")
    cfg = PPOConfig(n_envs=8, n_steps=c["B"], batch_size=c["B"], train_iters=1, use_graphs=False, fused_update=True, custom_mlp=True,
                    clip_ratio=float(G["clip_ratio"]), vf_coef=float(G["vf_coef"]), ent_coef=float(G["ent_coef"]),
                    max_grad_norm=float(G["max_grad_norm"]), learning_rate=float(G["lr"]))
    L = PPOLearner(agent, cfg, "cuda")
    assert L.custom, "the hand-written minibatch kernels must be the path under test"
    dev = lambda k, dt=torch.float32: torch.from_numpy(c[k]).to("cuda", dt).contiguous()
    L.custom_minibatch_step(dev("idx", torch.int64), dev("obs"), dev("act"), dev("old_logprob"), dev("adv"), dev("ret"))
    torch.cuda.synchronize()
    m = L.metrics.cpu().numpy()
    for got, key in zip(m, ("policy_loss", "value_loss", "entropy", "loss")):
        assert float(got) == pytest.approx(float(c[key]), rel=1e-5, abs=1e-6), key
    gref = _flat(c, "g_").astype(np.float64)
    norm = float(c["grad_norm"])
    gref = gref * min(1.0, float(G["max_grad_norm"]) / (norm + 1e-6))                 # clip_grad_norm_'s coefficient (train.py:260)
    assert np.allclose(L.flat_grad.cpu().numpy(), gref, rtol=2e-4, atol=2e-7)
    p1 = _flat(c, "p1_")
    p0 = _flat(c, "w_")
    got = L.flat_param.cpu().numpy()
    assert np.abs(got - p1).max() <= 3e-6                                            # one Adam step moves a parameter by ~lr = 3e-4
    assert np.abs(p1 - p0).max() > 1e-4                                              # (the step is visible at that tolerance)
