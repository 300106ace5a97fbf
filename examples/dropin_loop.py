#!/usr/bin/env python3
"""The reference's training loop (train.py:136-296), written against the reference-facing surface only -- the "minimal
swap" of INTEGRATION.md route A: `VecCarEnv` in place of `gym.vector.AsyncVectorEnv([...CarEnv...])`, and this package's
`Buffer` / `Agent` (same constructors and methods as lib/buffer.py / lib/model.py).  Everything else is plain torch,
as in the reference: Adam(eps=1e-5), StepLR(0.99), the clipped-PPO loss, clip_grad_norm_.  No fused kernels beyond what
those three objects do by themselves (env step, GAE scan); `ppo_car_amd.ppo.Trainer` is the fast form of the same loop.

    python examples/dropin_loop.py --track tracks/big_track.json --n-envs 256 --n-epochs 20
"""
import argparse
import os
import sys
import time

import numpy as np
import torch
import torch.nn as nn
import torch.optim as optim

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppo_car_amd import Agent, Buffer, VecCarEnv   # noqa: E402  (instead of lib.model / lib.buffer / gymnasium)


def run(track, n_envs=256, n_steps=256, n_epochs=10, batch_size=512, train_iters=40, num_rays=12, gamma=0.99, gae_lambda=0.95,
        clip_ratio=0.2, vf_coef=0.5, ent_coef=0.001, lr=3e-4, lr_decay=0.99, max_grad_norm=1.0, reward_scaling=0.1, seed=0,
        log=print):
    device = torch.device("cuda")
    torch.manual_seed(seed)
    envs = VecCarEnv(n_envs, track, num_rays=num_rays, reward_scaling=reward_scaling, device=device)   # train.py:137-139
    obs_dim, act_dim = envs.single_observation_space_shape, envs.single_action_space_n                 # :141-142
    agent = Agent(obs_dim[0], act_dim).to(device)                                                      # :145
    opt = optim.Adam(agent.parameters(), lr=lr, eps=1e-5)                                              # :146
    sched = optim.lr_scheduler.StepLR(opt, step_size=1, gamma=lr_decay)                                # :147
    buf = Buffer(obs_dim, n_steps, n_envs, device, gamma, gae_lambda)                                  # :152
    nxt_obs, _ = envs.reset(options={"track_path": track})                                             # :159 (a CUDA tensor already)
    nxt_term = torch.zeros(n_envs, device=device)                                                      # :165-166
    nxt_trunc = torch.zeros(n_envs, device=device)
    history, t0, steps_done = [], time.time(), 0
    for epoch in range(1, n_epochs + 1):
        rew_sum = torch.zeros((), device=device)
        for _ in range(n_steps):                                                                        # :173
            steps_done += n_envs
            o, te, tr = nxt_obs, nxt_term, nxt_trunc
            with torch.no_grad():
                a, lp, _, v = agent.get_action_and_value(o)                                             # :181
            nxt_obs, r, nxt_term, nxt_trunc, _ = envs.step(a)                                           # :185 (device in, device out)
            rew_sum += r.sum()
            buf.store(o, a, r, v.view(-1), te, tr, lp)                                                  # :195
        with torch.no_grad():
            last_v = agent.get_value(nxt_obs).reshape(1, -1)                                            # :200
            adv, ret = buf.calculate_advantages(last_v, nxt_term.reshape(1, -1), nxt_trunc.reshape(1, -1))   # :203
        t_obs, t_act, _t_val, t_lp = buf.get()                                                          # :206
        t_obs, t_act, t_lp = t_obs.view(-1, *obs_dim), t_act.view(-1), t_lp.view(-1)
        adv, ret = adv.view(-1), ret.view(-1)
        M = n_steps * n_envs
        sums = np.zeros(3)
        for _ in range(train_iters):                                                                    # :223
            perm = torch.randperm(M, device=device)                                                     # :225
            for start in range(0, n_steps, batch_size):                                                 # :228 (the reference's bound)
                idx = perm[start:start + batch_size]
                _, new_lp, ent, new_v = agent.get_action_and_value(t_obs[idx], t_act[idx])
                ratio = torch.exp(new_lp - t_lp[idx])                                                   # :235
                b_adv = adv[idx]
                b_adv = (b_adv - b_adv.mean()) / torch.max(b_adv.std(), torch.tensor(1e-5, device=device))   # :238-240
                pl = torch.max(-b_adv * ratio, -b_adv * torch.clamp(ratio, 1 - clip_ratio, 1 + clip_ratio)).mean()   # :243-245
                vl = 0.5 * ((new_v.view(-1) - ret[idx]) ** 2).mean()                                    # :249
                loss = pl + vf_coef * vl - ent_coef * ent.mean()                                        # :255
                opt.zero_grad()
                loss.backward()
                nn.utils.clip_grad_norm_(agent.parameters(), max_grad_norm)                             # :260
                opt.step()
                sums += np.array([float(pl.detach()), float(vl.detach()), float(ent.mean().detach())])
        sched.step()                                                                                    # :269
        avg_reward = float(rew_sum) / (n_steps * n_envs) / reward_scaling                               # :272-274
        history.append({"epoch": epoch, "avg_reward": avg_reward, "policy_loss": sums[0] / train_iters,
                        "value_loss": sums[1] / train_iters, "entropy": sums[2] / train_iters,
                        "sps": steps_done / (time.time() - t0)})
        log(f"Epoch {epoch} done in {time.time() - t0:.2f}s. Avg reward: {avg_reward:.4f}.")
    envs.close()                                                                                        # :296
    return agent, history


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--track", default="tracks/big_track.json")
    ap.add_argument("--n-envs", type=int, default=256)
    ap.add_argument("--n-steps", type=int, default=256)
    ap.add_argument("--n-epochs", type=int, default=10)
    ap.add_argument("--num-rays", type=int, default=12)
    a = ap.parse_args()
    run(a.track, n_envs=a.n_envs, n_steps=a.n_steps, n_epochs=a.n_epochs, num_rays=a.num_rays)
