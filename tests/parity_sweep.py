"""Wider sampling of the K9 / K9s-vs-oracle check of test_rollout_baseline_gpu.py (not collected by pytest; run by hand on a GPU box):
    python tests/parity_sweep.py [n_seeds]
For each seed and each of the three BASELINE shapes: one default-dispatch pc_rollout, its first 512 envs replayed through the
oracle for all T steps.  Prints the largest observation error, the number of envs that left the oracle's trajectory (each one
asserted to sit within 1e-3 px of a threshold) and the episodes that ended inside the compared window."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402
from conftest import TRACKS  # noqa: E402
from ppo_car_amd.ppo import PPOConfig, Trainer  # noqa: E402
from test_rollout_baseline_gpu import _oracle_replay_check, _snap  # noqa: E402

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for seed in range(101, 101 + n_seeds):
    for n_envs, num_rays, n_steps in [(65536, 16, 1024), (65536, 32, 128), (4096, 16, 1024)]:
        cfg = PPOConfig(n_envs=n_envs, n_steps=n_steps, num_rays=num_rays, track=TRACKS["big_track"], rollout_kernel="mega",
                        use_graphs=False, seed=seed)
        tr = Trainer(cfg, device="cuda")
        first = tr.next_obs.clone()
        tr.rollout()
        torch.cuda.synchronize()
        assert tr.rollout_mode == "mega"
        snaps = _snap(tr)
        tr.close()
        del tr
        worst, ties, alive = _oracle_replay_check(cfg, snaps, first, f"seed {seed} N={n_envs} rays={num_rays} T={n_steps}")
        print(f"seed {seed}  N={n_envs:6d} rays={num_rays:2d} T={n_steps:4d}: obs max err {worst:.2e}, near-tie departures {ties}, "
              f"on the oracle's trajectory {alive:.4f}", flush=True)
        del snaps
        torch.cuda.empty_cache()
