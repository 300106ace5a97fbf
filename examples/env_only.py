#!/usr/bin/env python3
"""The env alone under pre-generated actions (SURVEY 8(d) level (i)): `for t in range(T): envs.step(actions[t])` step by step and as ONE
call, `envs.step_many(actions)` -- the same rows, bit for bit (pc_env_step / pc_env_step_many, include/ppocar.h).

    python examples/env_only.py --n-envs 65536 --n-steps 256 --num-rays 16
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppo_car_amd import VecCarEnv   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--track", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tracks", "big_track.json"))
ap.add_argument("--n-envs", type=int, default=65536)
ap.add_argument("--n-steps", type=int, default=256)
ap.add_argument("--num-rays", type=int, default=16)
ap.add_argument("--dtype", default="f32")
a = ap.parse_args()
envs = VecCarEnv(a.n_envs, a.track, num_rays=a.num_rays, reward_scaling=0.1, dtype=a.dtype)
twin = VecCarEnv(a.n_envs, a.track, num_rays=a.num_rays, reward_scaling=0.1, dtype=a.dtype)
actions = torch.randint(0, 9, (a.n_steps, a.n_envs), device="cuda")          # Discrete(9), car_env.py:525
envs.reset()
twin.reset()
torch.cuda.synchronize()
t0 = time.perf_counter()
rows = [envs.step(actions[t])[:4] for t in range(a.n_steps)]                  # train.py:185, T times
torch.cuda.synchronize()
t1 = time.perf_counter()
obs, rew, term, trunc = twin.step_many(actions)                              # ... and as one call
torch.cuda.synchronize()
t2 = time.perf_counter()
same = all(torch.equal(torch.stack([r[i] for r in rows]), x) for i, x in enumerate((obs, rew, term, trunc)))
n = a.n_envs * a.n_steps
print(f"step by step ({envs.last_step_kernel()}): {n / (t1 - t0) / 1e6:.0f} M env-steps/s; one call ({twin.last_step_kernel()}): "
      f"{n / (t2 - t1) / 1e6:.0f} M env-steps/s; identical rows: {same}; episodes ended: {int(term.sum() + trunc.sum())}")
