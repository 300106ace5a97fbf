"""The N > 1 path on CPU: world_size 2, gloo.  Envs shard across ranks with no data-path collective;
the only exchange is one all-reduce of the flat gradient bucket per minibatch (ppo.GradExchange),
after which clip_grad_norm_ and Adam run replicated.  The env / GAE kernels are GPU-only, so the shards
here are synthetic rollouts; what is under test is the learner's distributed arithmetic."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ppo_car_amd import Agent
from ppo_car_amd.ppo import PPOConfig, PPOLearner, flatten_parameters, ppo_loss

D, A, M = 23, 9, 256


def _shard(rank):
    g = torch.Generator().manual_seed(1000 + rank)
    obs = torch.randn(M, D, generator=g)
    act = torch.randint(0, A, (M,), generator=g).float()
    lp = -torch.rand(M, generator=g) * 2.0
    adv = torch.randn(M, generator=g) * (1.0 + rank)
    ret = torch.randn(M, generator=g)
    return obs, act, lp, adv, ret


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(50 + rank)                 # DIFFERENT init per rank: the learner must broadcast rank 0's
    agent = Agent(D, A)
    cfg = PPOConfig(n_envs=4, n_steps=64, batch_size=32, train_iters=2, seed=3)
    learner = PPOLearner(agent, cfg, "cpu", rank=rank, world_size=world)
    p_init = learner.flat_param.clone()
    shard = _shard(rank)
    for k in range(3):
        sl = slice(k * 32, (k + 1) * 32)
        learner.minibatch_step(*[t[sl] for t in shard])
    # a full update() call as well (per-rank index draws, scheduler step)
    learner.update(*shard)
    torch.save({"init": p_init, "after3": None, "final": learner.flat_param.clone(), "lr": learner.optimizer.param_groups[0]["lr"],
                "metrics": learner.metrics.clone()}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


def _worker_steps_only(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(50 + rank)
    agent = Agent(D, A)
    cfg = PPOConfig(n_envs=4, n_steps=64, batch_size=32, train_iters=2, seed=3)
    learner = PPOLearner(agent, cfg, "cpu", rank=rank, world_size=world)
    shard = _shard(rank)
    for k in range(3):
        sl = slice(k * 32, (k + 1) * 32)
        learner.minibatch_step(*[t[sl] for t in shard])
    torch.save(learner.flat_param.clone(), os.path.join(out_dir, f"steps_rank{rank}.pt"))
    dist.destroy_process_group()


def test_two_rank_update_matches_manual_gradient_average(tmp_path):
    world = 2
    mp.spawn(_worker_steps_only, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    p0, p1 = (torch.load(tmp_path / f"steps_rank{r}.pt") for r in range(world))
    assert torch.equal(p0, p1)                   # replicas stay bit-identical
    # single-process restatement: rank 0's init, per-shard losses, averaged gradients, clip, Adam
    torch.manual_seed(50)
    agent = Agent(D, A)
    cfg = PPOConfig()
    flat, flat_grad = flatten_parameters(agent)
    opt = torch.optim.Adam(agent.parameters(), lr=cfg.learning_rate, eps=1e-5)
    shards = [_shard(r) for r in range(world)]
    for k in range(3):
        sl = slice(k * 32, (k + 1) * 32)
        grads = []
        for sh in shards:
            flat_grad.zero_()
            loss, *_ = ppo_loss(agent, *[t[sl] for t in sh], cfg.clip_ratio, cfg.vf_coef, cfg.ent_coef)
            loss.backward()
            grads.append(flat_grad.clone())
        flat_grad.copy_((grads[0] + grads[1]) / world)
        torch.nn.utils.clip_grad_norm_(agent.parameters(), cfg.max_grad_norm)
        opt.step()
    assert torch.allclose(flat, p0, rtol=0, atol=1e-7)


def test_two_rank_full_update_keeps_replicas_identical(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (torch.load(tmp_path / f"rank{r}.pt") for r in range(world))
    assert torch.equal(r0["init"], r1["init"])       # broadcast of rank 0's parameters
    assert torch.equal(r0["final"], r1["final"])     # after 3 + 2*2 minibatch steps with different shards / index draws
    assert not torch.equal(r0["init"], r0["final"])
    assert r0["lr"] == r1["lr"] == pytest.approx(3e-4 * 0.99)
    assert not torch.equal(r0["metrics"], r1["metrics"])   # losses are per-shard quantities


def test_eight_rank_full_update_keeps_replicas_identical(tmp_path):
    """The world size of BASELINE configs[3] / [4] (8 ranks, gloo on the CPU): rank 0's parameters broadcast to all, eight different
    shards and index streams, one all-reduce of the flat bucket per minibatch (train.py:259-260) -- every replica ends on the same bits."""
    world = 8
    torch.set_num_threads(1)
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    rs = [torch.load(tmp_path / f"rank{r}.pt") for r in range(world)]
    for r in rs[1:]:
        assert torch.equal(rs[0]["init"], r["init"]) and torch.equal(rs[0]["final"], r["final"])
        assert r["lr"] == rs[0]["lr"]
    assert not torch.equal(rs[0]["init"], rs[0]["final"])
    assert len({tuple(r["metrics"].tolist()) for r in rs}) == world      # losses are per-shard quantities


def test_single_rank_learner_follows_reference_loop_bounds():
    """train.py:228 quirk: ceil(n_steps / batch_size) minibatches per train iter, whatever n_envs is."""
    torch.manual_seed(0)
    agent = Agent(D, A)
    cfg = PPOConfig(n_envs=8, n_steps=64, batch_size=16, train_iters=3, seed=1)
    learner = PPOLearner(agent, cfg, "cpu")
    assert learner.n_minibatches == 4
    calls = []
    orig = learner.minibatch_step
    learner.minibatch_step = lambda *a: (calls.append(a[0].shape[0]), orig(*a))[1]
    g = torch.Generator().manual_seed(0)
    Mx = 8 * 64
    learner.update(torch.randn(Mx, D, generator=g), torch.randint(0, A, (Mx,), generator=g).float(), -torch.rand(Mx, generator=g),
                   torch.randn(Mx, generator=g), torch.randn(Mx, generator=g))
    assert calls == [16] * 12
    idx = learner.draw_indices(Mx)
    assert idx.shape == (3, 64) and all(len(set(r.tolist())) == 64 for r in idx)     # without replacement
    assert int(idx.max()) < Mx and int(idx.min()) >= 0
    assert learner.optimizer.param_groups[0]["lr"] == pytest.approx(3e-4 * 0.99)


def test_ppo_loss_matches_restated_reference_expression():
    """train.py:233-255 restated term by term (train.py itself cannot be imported: tkinter, cv2, gymnasium)."""
    torch.manual_seed(2)
    agent = Agent(D, A)
    obs, act, lp, adv, ret = _shard(0)
    loss, pl, vl, ent = ppo_loss(agent, obs, act, lp, adv, ret, 0.2, 0.5, 0.001)
    logits = agent.actor(obs)
    dist_ = torch.distributions.Categorical(logits=logits)
    new_lp, entropy, v = dist_.log_prob(act), dist_.entropy(), agent.critic(obs).view(-1)
    ratios = torch.exp(new_lp - lp)
    a = (adv - adv.mean()) / max(float(adv.std()), 1e-5)
    pl_ref = torch.max(-a * ratios, -a * torch.clamp(ratios, 0.8, 1.2)).mean()
    vl_ref = 0.5 * ((v - ret) ** 2).mean()
    assert torch.allclose(pl, pl_ref, atol=1e-6) and torch.allclose(vl, vl_ref, atol=1e-6)
    assert torch.allclose(ent, entropy.mean()) and torch.allclose(loss, pl_ref + 0.5 * vl_ref - 0.001 * entropy.mean(), atol=1e-6)
    assert sum(p.numel() for p in agent.parameters()) == 14858          # D = 23
    assert sum(p.numel() for p in Agent(18, 9).parameters()) == 12298
