"""Developer tool (GPU): train the policy for 150 epochs on big_track (the run DESIGN.md section 8 describes) and write
tests/golden/policy_trained.npz -- the trained weights plus ~3900 observations harvested from its last rollout (incl. the rows
right after a reset: velocity exactly 0) -- for tests/test_gae_sample_gpu.py's precision test of
the fused policy kernel on REAL weights and inputs.  Output goes to gpurun_out/ on the GPU box; copy it to tests/golden/."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ppo_car_amd.ppo import PPOConfig, Trainer  # noqa: E402

cfg = PPOConfig(n_envs=256, n_steps=1024, batch_size=512, train_iters=40, num_rays=16, track=f"{ROOT}/tracks/big_track.json", seed=0)
tr = Trainer(cfg, device="cuda")
for ep in range(150):
    s = tr.run_epoch()
    if ep % 25 == 0 or ep == 149:
        print(ep, s["charts/avg_reward"], flush=True)
obs = tr.buffer.obs_buf.reshape(-1, tr.obs_dim[0])
g = torch.Generator(device="cuda").manual_seed(1)
pick = torch.randperm(obs.shape[0], device="cuda", generator=g)[:3584]
after_reset = (tr.buffer.term_buf.reshape(-1) != 0).nonzero().flatten()[:256]          # reset observations: zero velocity
slow = (obs[:, 2:4].abs().max(1).values < 1e-3).nonzero().flatten()[:256]               # tiny velocities
rows = torch.cat([obs[pick], obs[after_reset], obs[slow]])[:4096].cpu().numpy()
sd = {k: v.detach().cpu().numpy() for k, v in tr.agent.state_dict().items()}
out = os.path.join(ROOT, "gpurun_out", "policy_trained.npz")
os.makedirs(os.path.dirname(out), exist_ok=True)
np.savez_compressed(out, obs=rows.astype(np.float32), avg_reward=np.float32(s["charts/avg_reward"]), **{k.replace(".", "_"): v for k, v in sd.items()})
print("wrote", out, rows.shape, "final avg reward", s["charts/avg_reward"], "min |vel|", float(np.abs(rows[:, 2:4]).min()),
      "rays at 1.0:", int((rows[:, 6:] >= 1.0).sum()))
