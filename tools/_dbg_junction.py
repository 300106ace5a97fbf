import sys, os, torch, numpy as np
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ppo_car_amd.ppo import PPOConfig, Trainer
from test_env_gpu import _junction_track_json
from test_rollout_baseline_gpu import _snap
path = _junction_track_json("/tmp/junction.json")
import ppo_car_amd as pc
e = pc.VecCarEnv(4, path, num_rays=16); print("track info", e.track_info)
for n_envs, epw in ((2048, 0), (2048, 32), (6000, 0)):
    res = {}
    for mode in ("mega", "steps"):
        cfg = PPOConfig(n_envs=n_envs, n_steps=200, num_rays=16, track=path, rollout_kernel=mode, use_graphs=False, seed=19, rollout_epw=epw)
        tr = Trainer(cfg, device="cuda"); tr.rollout(); torch.cuda.synchronize()
        print(mode, tr.envs.last_rollout_kernel())
        res[mode] = _snap(tr); tr.close()
    a, b = res["mega"][0], res["steps"][0]
    d = (a != b).nonzero()
    print(n_envs, epw, "obs diffs", d.shape[0])
    for r in d[:4].tolist():
        t, e_, c = r
        print(" t", t, "env", e_, "col", c, "mega", float(a[t, e_, c]), "steps", float(b[t, e_, c]))
