import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from ppo_car_amd.ppo import PPOConfig, Trainer
root = os.getcwd()
res = {}
for name, kw in (("steps", dict(rollout_kernel="steps")), ("m4", dict(rollout_kernel="mega")), ("m1", dict(rollout_kernel="mega", rollout_fast=2))):
    cfg = PPOConfig(n_envs=int(os.environ.get("DBG_N", "65536")), n_steps=int(os.environ.get("DBG_T", "128")), num_rays=32, track=f"{root}/tracks/big_track.json", seed=3, use_graphs=False, policy_split=0, **kw)
    tr = Trainer(cfg, device="cuda")
    tr.rollout(); torch.cuda.synchronize()
    res[name] = tr.buffer.obs_buf.clone().cpu().numpy()
    print(name, tr.rollout_mode)
    tr.close()
for k in ("m4", "m1"):
    d = res[k] != res["steps"]
    print(k, "differing entries", int(d.sum()), "of", d.size)
    if d.any():
        t, e, c = np.argwhere(d)[0]
        print(" first at t,e,col", t, e, c, res[k][t, e, c], res["steps"][t, e, c])
        cols = np.unique(np.argwhere(d)[:, 2]); print(" columns", cols[:40])
        ts = np.unique(np.argwhere(d)[:, 0]); print(" first steps", ts[:10])
        es = np.unique(np.argwhere(d)[:, 1]); print(" envs", len(es), es[:20], "env%256", np.unique(es % 256)[:40], "wave", np.unique((es % 256) // 32))
        tt, ee, cc = np.argwhere(d)[0]
        print(" row steps/mega:", res["steps"][tt, ee], res[k][tt, ee])
# arbitrate with the oracle: replay the differing env from the reset state with the stored actions
import oracle
cfg = PPOConfig(n_envs=int(os.environ.get("DBG_N", "65536")), n_steps=int(os.environ.get("DBG_T", "128")), num_rays=32, track=f"{root}/tracks/big_track.json", seed=3, use_graphs=False, policy_split=0, rollout_kernel="steps")
tr = Trainer(cfg, device="cuda"); tr.rollout(); torch.cuda.synchronize()
acts = tr.buffer.act_buf.cpu().numpy().astype(np.int64)
d = res["m4"] != res["steps"]
if d.any():
    tt, ee, cc = np.argwhere(d)[0]
    ora = oracle.OracleVecEnv(oracle.Track(cfg.track), 1, num_rays=32, reward_scaling=0.1)
    ora.reset()
    for t in range(tt):
        o, r, te, trn = ora.step(acts[t, ee:ee + 1])
    print("oracle obs at that entry:", o[0, cc], "steps", res["steps"][tt, ee, cc], "m4", res["m4"][tt, ee, cc])
    print("oracle state px py rot", ora.px, ora.py, ora.rot, "vx vy", ora.vx, ora.vy)
