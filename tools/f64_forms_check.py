#!/usr/bin/env python3
"""Developer probe: the two persistent kernels of the float64 dtype against each other at scale.  Two trainers of one seed, one on the
SELECTOR form (default: K9 with the reference's literal arithmetic behind the float32 sweep), one on the FILTER form
(PC_OPT_ROLLOUT_FAST = 0: every ray x wall pair in float64), run whole epochs in lockstep (rollout + GAE + update: as long as every
buffer is bit-identical the policies stay identical, and the cars get further round the track as they learn); after every epoch
every rollout buffer and the float64 env state are compared bit for bit.
The small form's variants (small16: 4096 envs, small32: 8192) and 33 rays are compared with the per-step kernels instead.
usage: python tools/f64_forms_check.py [epochs] [shape ...]     shapes: target (65536 x 1024, big_track), mixed (32768 x 512, both tracks), rays12, rays33, small16, small32"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ppo_car_amd.ppo import PPOConfig, Trainer  # noqa: E402

BIG, SMALL = f"{ROOT}/tracks/big_track.json", f"{ROOT}/tracks/track.json"
SHAPES = {"target": dict(n_envs=65536, n_steps=1024, num_rays=16, track=BIG), "mixed": dict(n_envs=32768, n_steps=512, num_rays=16, track=[SMALL, BIG]),
          "rays12": dict(n_envs=32768, n_steps=512, num_rays=12, track=BIG), "rays33": dict(n_envs=32768, n_steps=128, num_rays=32, track=BIG),
          # the small form's two variants have no filter-form twin (K9d is a big-form kernel with the unsplit policy arithmetic): their
          # partner is the per-step path (policy_kernel<SPLIT>; env_step_kernel<double>)
          "small16": dict(n_envs=4096, n_steps=1024, num_rays=16, track=BIG), "small32": dict(n_envs=8192, n_steps=512, num_rays=16, track=BIG)}


def snap(t):
    b = t.buffer
    return [x.clone() for x in (b.obs_buf, b.act_buf, b.rew_buf, b.val_buf, b.logprob_buf, b.term_buf, b.trunc_buf, t.next_obs, t.next_term, t.next_trunc)]


def against_per_step(name, epochs):
    """ONE trainer (use_graphs off): every epoch's rollout is launched twice from the same env state, observation, flags and Philox
    counters -- as the persistent launch, then through the per-step kernels -- compared, and the update runs once.  (Two trainers in
    lockstep would drift: the persistent launch hands the update its own bootstrap value, the per-step path evaluates the critic
    through torch.)"""
    t = Trainer(PPOConfig(seed=11, rollout_kernel="mega", env_dtype="f64", use_graphs=False, **SHAPES[name]), device="cuda")
    diff, casts, kernels = 0, 0, None
    for ep in range(epochs):
        st = t.envs.get_state()
        keep = [x.clone() for x in (t.next_obs, t.next_term, t.next_trunc, t.rng_base)]
        t.cfg.rollout_kernel = "mega"
        t.rollout()
        torch.cuda.synchronize()
        kernels = [t.envs.last_rollout_kernel()]
        a, sa = snap(t), t.envs.get_state()
        t.envs.set_state(**st)
        for dst, src in zip((t.next_obs, t.next_term, t.next_trunc, t.rng_base), keep):
            dst.copy_(src)
        t.buffer.ptr = 0
        t.cfg.rollout_kernel = "steps"
        t.rollout()
        torch.cuda.synchronize()
        kernels.append(t.rollout_mode)
        b, sb = snap(t), t.envs.get_state()
        diff += sum(int((x != y).sum()) for x, y in zip(a, b)) + sum(int((sa[k] != sb[k]).sum()) for k in sa)
        casts += t.cfg.n_envs * t.cfg.n_steps * (t.obs_dim[0] - 6)
        t.update()
        del a, b
    print(json.dumps({"shape": name, "kernels": kernels, "epochs": epochs, "ray_casts_compared": casts, "entries_differing": diff,
                      "mean_gates_passed_last_epoch": float(np.mean(t.envs.get_state()["passed"]))}), flush=True)
    t.close()
    torch.cuda.empty_cache()


def main():
    epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    for name in sys.argv[2:] or ["target"]:
        if name.startswith("small") or name == "rays33":      # (33 rays: the filter form is not built either)
            against_per_step(name, epochs)
            continue
        trs = [Trainer(PPOConfig(seed=11, rollout_kernel="mega", env_dtype="f64", rollout_fast=f, **SHAPES[name]), device="cuda") for f in (1, 0)]
        diff, casts, kernels = 0, 0, None
        for ep in range(epochs):
            for t in trs:
                t.run_epoch(sync=False)
            torch.cuda.synchronize()
            kernels = [t.envs.last_rollout_kernel() if t.rollout_mode == "mega" else t.rollout_mode for t in trs]
            a, b = (t.buffer for t in trs)
            for x, y in zip((a.obs_buf, a.act_buf, a.rew_buf, a.val_buf, a.logprob_buf, a.term_buf, a.trunc_buf, trs[0].next_obs),
                            (b.obs_buf, b.act_buf, b.rew_buf, b.val_buf, b.logprob_buf, b.term_buf, b.trunc_buf, trs[1].next_obs)):
                diff += int((x != y).sum())
            sa, sb = (t.envs.get_state() for t in trs)
            diff += sum(int((sa[k] != sb[k]).sum()) for k in sa)
            c = trs[0].cfg
            casts += c.n_envs * c.n_steps * (trs[0].obs_dim[0] - 6)
        print(json.dumps({"shape": name, "kernels": kernels, "epochs": epochs, "ray_casts_compared": casts, "entries_differing": diff,
                          "mean_gates_passed_last_epoch": float(np.mean(trs[0].envs.get_state()["passed"]))}), flush=True)
        for t in trs:
            t.close()
        del trs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
