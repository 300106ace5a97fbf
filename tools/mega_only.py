"""Run only the persistent rollout kernel (for rocprofv3): N envs, 16 rays, T steps, a few launches.
   python tools/mega_only.py [N] [T] [mega|steps] [f32|f64] [rollout_fast: 1 | 0 (f64: 0 = the filter form)]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ppo_car_amd.ppo import PPOConfig, Trainer
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T = int(sys.argv[2]) if len(sys.argv) > 2 else 256
mode = sys.argv[3] if len(sys.argv) > 3 else "mega"
dtype = sys.argv[4] if len(sys.argv) > 4 else "f32"      # "f64": the bit-exact dtype's persistent kernel (K9d)
fast = int(sys.argv[5]) if len(sys.argv) > 5 else 1
cfg = PPOConfig(n_envs=N, n_steps=T, num_rays=16, track=f"{ROOT}/tracks/big_track.json", rollout_kernel=mode, use_graphs=False, env_dtype=dtype, rollout_fast=fast)
tr = Trainer(cfg, device="cuda")
for _ in range(3):
    tr.rollout(); tr.buffer.ptr = 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); tr.rollout(); e1.record(); torch.cuda.synchronize()
print("mode", tr.rollout_mode, "kernel", tr.envs.last_rollout_kernel() if tr.rollout_mode == "mega" else "-", "rollout ms", e0.elapsed_time(e1), "us/step", e0.elapsed_time(e1) * 1e3 / T)
