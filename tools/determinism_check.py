#!/usr/bin/env python3
"""Developer probe: is pc_rollout deterministic launch to launch?  For each shape the same rollout (same env state, same Philox
counters, same weights) is launched REPS times; every launch's buffers must be bit-identical to the first's.  A run-to-run
difference is a race or a hazard in the kernel (round 5: a 33-ray chain-packed variant differed in ONE of 3e8 observation entries in
one run of two and was not shipped).   usage: python tools/determinism_check.py [reps] [shape ...]   shapes: target cfg1 cfg2 cfg4 f64"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ppo_car_amd.ppo import PPOConfig, Trainer  # noqa: E402

SHAPES = {"target": dict(n_envs=65536, n_steps=1024, num_rays=16), "cfg1": dict(n_envs=4096, n_steps=1024, num_rays=16),
          "cfg2": dict(n_envs=65536, n_steps=128, num_rays=32), "cfg4": dict(n_envs=32768, n_steps=1024, num_rays=16, mixed=True),
          "f64": dict(n_envs=65536, n_steps=128, num_rays=16, env_dtype="f64")}


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    names = sys.argv[2:] or ["cfg1", "cfg2", "target"]
    for name in names:
        kw = dict(SHAPES[name])
        mixed = kw.pop("mixed", False)
        track = [f"{ROOT}/tracks/track.json", f"{ROOT}/tracks/big_track.json"] if mixed else f"{ROOT}/tracks/big_track.json"
        tr = Trainer(PPOConfig(track=track, seed=7, rollout_kernel="mega", **kw), device="cuda")
        for _ in range(2):      # a trained-ish policy and envs spread over the track
            tr.run_epoch(sync=False)
        torch.cuda.synchronize()
        st = tr.envs.get_state()
        keep = [t.clone() for t in (tr.next_obs, tr.next_term, tr.next_trunc, tr.rng_base)]
        first, bad, worst = None, 0, 0
        for r in range(reps):
            tr.envs.set_state(**st)
            for dst, src in zip((tr.next_obs, tr.next_term, tr.next_trunc, tr.rng_base), keep):
                dst.copy_(src)
            tr.rollout()
            torch.cuda.synchronize()
            b = tr.buffer
            cur = [x.clone() for x in (b.obs_buf, b.act_buf, b.rew_buf, b.val_buf, b.logprob_buf, b.term_buf, b.trunc_buf, tr.next_obs)]
            if first is None:
                first = cur
            else:
                n = sum(int((a != c).sum()) for a, c in zip(first, cur))
                bad += n > 0
                worst = max(worst, n)
            del cur
        print(json.dumps({"shape": name, "kernel": tr.rollout_mode, "launches": reps, "launches_differing_from_the_first": bad,
                          "most_differing_entries": worst, "entries_per_launch": sum(x.numel() for x in first)}), flush=True)
        tr.close()
        del first, tr
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
