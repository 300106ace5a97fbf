#!/usr/bin/env python3
"""Developer soak (GPU box): pc_env_step_many (K1f, one launch for T steps) against T launches of the generic kernel K1, two env
batches in lockstep from the same state, every output row and the env state compared after every launch.

    python tools/soak_steps.py [--launches 40 --n-envs 65536 --n-steps 256 --rays 16 --dtype f32 --out profiles/r6_soak_shipped.jsonl]

Actions: a forward-biased random stream (cars reach gates, laps and walls; ~1 % of the entries outside 0..8).  One JSON line."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ppo_car_amd as pc  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--launches", type=int, default=40)
ap.add_argument("--n-envs", type=int, default=65536)
ap.add_argument("--n-steps", type=int, default=256)
ap.add_argument("--rays", type=int, default=16)
ap.add_argument("--dtype", default="f32")
ap.add_argument("--track", default="big_track")
ap.add_argument("--out", default=None)
a = ap.parse_args()
N, T = a.n_envs, a.n_steps
track = os.path.join(ROOT, "tracks", a.track + ".json")
A, B = (pc.VecCarEnv(N, track, num_rays=a.rays, reward_scaling=0.1, dtype=a.dtype) for _ in range(2))
A.set_option("step_form", 1)
B.set_option("step_form", 0)
A.reset()
B.reset()
g = torch.Generator(device="cuda").manual_seed(11)
D = A.obs_dim
rows = (torch.empty(T, N, D, device="cuda"), torch.empty(T, N, device="cuda"), torch.empty(T, N, device="cuda"), torch.empty(T, N, device="cuda"))
many = tuple(torch.empty_like(r) for r in rows)
events, entries, done, laps = 0, 0, 0, 0
for it in range(a.launches):
    acts = torch.randint(0, 9, (T, N), generator=g, device="cuda")
    acts = torch.where(torch.rand(T, N, generator=g, device="cuda") < 0.45, torch.zeros_like(acts), acts)
    acts = torch.where(torch.rand(T, N, generator=g, device="cuda") < 0.01, torch.randint(-5, 300, (T, N), generator=g, device="cuda"), acts)
    for t in range(T):
        A.step(acts[t], out=tuple(r[t] for r in rows))
    B.step_many(acts, out=many)
    for x, y in zip(rows, many):
        events += int((x != y).sum())
        entries += x.numel()
    sa, sb = A.get_state(), B.get_state()
    events += sum(int((sa[k] != sb[k]).sum()) for k in sa)
    done += int(rows[2].sum() + rows[3].sum())
    laps += int((rows[1] > 1.05).sum())
    if it % 10 == 9:
        print(f"launch {it + 1}: {entries:.2e} entries, {events} differences, {done} episodes ended, {laps} laps", flush=True)
rec = {"summary": True, "kernel": B.last_step_kernel() + " (pc_env_step_many) against " + A.last_step_kernel() + " step by step", "launches": a.launches, "n_envs": N,
       "n_steps": T, "rays": a.rays, "dtype": a.dtype, "entries_compared": entries, "events": events, "episodes_ended": done, "laps": laps}
print(json.dumps(rec))
if a.out:
    with open(a.out, "a") as f:
        f.write(json.dumps(rec) + "\n")
sys.exit(1 if events else 0)
