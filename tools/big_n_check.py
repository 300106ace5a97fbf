"""Developer check: persistent rollout kernel vs the per-step kernels, bit for bit, at batch sizes beyond the test suite's (262144 envs; a size that is not a multiple of 256; track.json)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppo_car_amd.ppo import PPOConfig, Trainer
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for N, T, trk, rays in ((262144, 24, "big_track", 16), (100000, 32, "track", 12), (65536 + 17, 16, "big_track", 32)):
    res = {}
    for mode in ("steps", "mega"):
        cfg = PPOConfig(n_envs=N, n_steps=T, num_rays=rays, track=f"{ROOT}/tracks/{trk}.json", rollout_kernel=mode, use_graphs=False, seed=2)
        tr = Trainer(cfg, device="cuda")
        tr.rollout(); torch.cuda.synchronize()
        b = tr.buffer
        res[mode] = (tr.rollout_mode, [x.clone() for x in (b.obs_buf, b.act_buf, b.rew_buf, b.term_buf, b.logprob_buf, tr.next_obs)])
        tr.close(); del tr; torch.cuda.empty_cache()
    ok = all(torch.equal(x, y) for x, y in zip(res["steps"][1], res["mega"][1]))
    print(N, T, trk, rays, res["steps"][0], res["mega"][0], "bitwise equal" if ok else "MISMATCH", "terms", int(res["mega"][1][3].sum()))
