"""Probe (developer tool): mixed-track batch, 32 rays, forced lanes-per-env so that the env-step kernel variant with the
largest register footprint runs (env_step_kernel<float, 33 / 17, MIXED>)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ppo_car_amd as pc
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = 1000
tid = (np.arange(N) % 2).astype(np.uint8)
env = pc.VecCarEnv(N, [f"{ROOT}/tracks/track.json", f"{ROOT}/tracks/big_track.json"], num_rays=32, track_id=tid)
env.set_lanes_per_env(lanes)
print(env.launch_info(), flush=True)
obs, _ = env.reset()
a = torch.zeros(N, dtype=torch.int64, device="cuda")
for t in range(20):
    obs, r, te, tr, _ = env.step(a)
torch.cuda.synchronize()
print("ok", float(obs.sum()), flush=True)
