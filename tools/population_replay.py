"""Developer / evidence tool (run on a GPU box): the persistent rollout kernel in its default dispatch against the float64 CPU
oracle over a POPULATION of the launch -- envs strided so that every 32-env wave (hence every workgroup) is represented --
for all T steps:  python tools/population_replay.py [target|cfg2|cfg1 ...] [--per-wave K] [--dtype f32|f64] [--out FILE]
Writes the observation-error histogram, the bit-equal fraction and every departure from the oracle's trajectory (with its
threshold margin) as JSON lines.  train.py:173-195 is the loop being replayed."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from conftest import TRACKS  # noqa: E402
from ppo_car_amd.ppo import PPOConfig, Trainer  # noqa: E402
from test_rollout_baseline_gpu import _oracle_replay_check, _snap, strided_population  # noqa: E402

SHAPES = {"target": (65536, 16, 1024), "cfg2": (65536, 32, 128), "cfg1": (4096, 16, 1024)}
ap = argparse.ArgumentParser()
ap.add_argument("shapes", nargs="*", default=["target"])
ap.add_argument("--per-wave", type=int, default=4)
ap.add_argument("--seed", type=int, default=11)
ap.add_argument("--epochs", type=int, default=1, help="rollouts before the one that is replayed (later rollouts start mid-episode)")
ap.add_argument("--dtype", default="f32", help='env dtype: "f32" (default) or "f64" (the bit-exact dtype: every entry must be bit-equal)')
ap.add_argument("--out", default=None)
args = ap.parse_args()
out = open(args.out, "a") if args.out else None
for name in args.shapes:
    n_envs, num_rays, n_steps = SHAPES[name]
    cfg = PPOConfig(n_envs=n_envs, n_steps=n_steps, num_rays=num_rays, track=TRACKS["big_track"], rollout_kernel="mega",
                    use_graphs=False, seed=args.seed, env_dtype=args.dtype)
    tr = Trainer(cfg, device="cuda")
    first = tr.next_obs.clone()
    tr.rollout()
    torch.cuda.synchronize()
    assert tr.rollout_mode == "mega"
    kernel = tr.envs.last_rollout_kernel()
    snaps = _snap(tr)
    tr.close()
    del tr
    sel = strided_population(n_envs, per_wave=args.per_wave)
    stats = {}
    t0 = time.time()
    worst, ties, alive = _oracle_replay_check(cfg, snaps, first, name, sel=sel, stats=stats)
    rec = {"shape": name, "dtype": args.dtype, "kernel": kernel, "n_envs": n_envs, "num_rays": num_rays, "n_steps": n_steps, "seed": args.seed,
           "waves_covered": int(len(set((sel // 32).tolist()))), "waves_total": (n_envs + 31) // 32,
           "oracle_seconds": round(time.time() - t0, 1), **stats}
    rec["above_1e-6"] = int(sum(h for e, h in zip(rec["hist_edges"], rec["hist"]) if e >= 1e-6))
    rec["above_1e-5"] = int(sum(h for e, h in zip(rec["hist_edges"], rec["hist"]) if e >= 1e-5))
    line = json.dumps(rec)
    print(line, flush=True)
    if out:
        out.write(line + "\n")
        out.flush()
    del snaps
    torch.cuda.empty_cache()
