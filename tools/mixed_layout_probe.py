"""Developer probe (GPU box): the rollout of configs[4]'s shard (32768 envs, track.json + big_track.json) per env layout and kernel path --
halves through the fast modes (K9m), halves through the generic mode (rollout_fast = 0), interleaved (track_id = i & 1) de-interleaved by
wave (mode 7: one pass per wave), through the two-track form's per-track passes (mode 6: both tracks' tables in LDS, the env step once per
track of a wave), through the generic mode's per-wave track waterfall, and through the per-step kernels (graph replay).  us per vector step, median of N launches."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ppo_car_amd.ppo import PPOConfig, Trainer  # noqa: E402

N, T = int(sys.argv[1]) if len(sys.argv) > 1 else 32768, 256
tracks = [f"{ROOT}/tracks/track.json", f"{ROOT}/tracks/big_track.json"]
for name, kw in (("halves, fast modes", dict()), ("halves, generic mode", dict(rollout_fast=0)),
                 ("interleaved, de-interleaved by wave (mode 7)", dict(track_interleave=True)),
                 ("interleaved, per-track passes (mode 6)", dict(track_interleave=True, rollout_fast=3)),
                 ("interleaved, generic mode (waterfall)", dict(track_interleave=True, rollout_fast=2)),
                 ("interleaved, per-step kernels (graph)", dict(track_interleave=True, rollout_kernel="steps")),
                 ("single track big_track, generic mode", dict(rollout_fast=0, _single=True))):
    single = kw.pop("_single", False)
    for dtype in ("f32", "f64"):
        tr = Trainer(PPOConfig(n_envs=N, n_steps=T, num_rays=16, track=tracks[1] if single else tracks, seed=3, env_dtype=dtype, **kw), device="cuda")
        for _ in range(3):
            tr.run_epoch(sync=False)
        ts = []
        for _ in range(12):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); tr.rollout(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / T)
            tr.buffer.ptr = 0
        ts.sort()
        k = tr.envs.last_rollout_kernel() if tr.rollout_mode == "mega" else tr.rollout_mode
        print(f"{name:44s} {dtype}  {ts[len(ts) // 2]:8.2f} us per vector step of {N} envs  ({k})", flush=True)
        tr.close()
