"""Run only the env-step kernel (for rocprofv3 --pmc passes): N envs, 16 rays, big_track, 40 launches."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ppo_car_amd as pc
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dtype = sys.argv[3] if len(sys.argv) > 3 else "f32"
env = pc.VecCarEnv(N, f"{ROOT}/tracks/big_track.json", num_rays=n, reward_scaling=0.1, dtype=dtype)
obs, _ = env.reset()
g = torch.Generator(device="cuda").manual_seed(0)
acts = torch.randint(0, 9, (40, N), device="cuda", generator=g)
out = (obs, torch.empty(N, device="cuda"), torch.empty(N, device="cuda"), torch.empty(N, device="cuda"))
for t in range(40):
    env.step(acts[t], out=out)
torch.cuda.synchronize()
print("done", env.launch_info())
