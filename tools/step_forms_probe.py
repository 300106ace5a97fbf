#!/usr/bin/env python3
"""Developer probe (GPU box): pc_env_step's two kernels and pc_env_step_many over a ladder of batch sizes.

    python tools/step_forms_probe.py [rays] [track]       -> us per vector step: K1 (generic), K1f (table-driven, one launch per step),
                                                             K1f inside pc_env_step_many (T = 256: one launch, 1/den table staged)
How PC_STEP_FAST_MIN_ENVS was chosen and what DESIGN.md section 4 quotes for the env alone."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ppo_car_amd as pc  # noqa: E402

rays = int(sys.argv[1]) if len(sys.argv) > 1 else 16
track = os.path.join(ROOT, "tracks", (sys.argv[2] if len(sys.argv) > 2 else "big_track") + ".json")
REPS, T = 200, 256


def timed(fn, reps):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for dtype in ("f32", "f64"):
    for N in [int(x) for x in os.environ.get("PROBE_N", "2048,4096,8192,16384,32768,65536,131072,262144").split(",")]:
        g = torch.Generator().manual_seed(N)
        acts = torch.randint(0, 9, (T, N), generator=g).cuda()
        res = {}
        for form, name in ((1, "K1"), (2, "K1f")):
            env = pc.VecCarEnv(N, track, num_rays=rays, reward_scaling=0.1, dtype=dtype)
            env.set_option("step_form", form)
            env.reset()
            out = (torch.empty(N, env.obs_dim, device="cuda"), torch.empty(N, device="cuda"), torch.empty(N, device="cuda"), torch.empty(N, device="cuda"))
            for t in range(40):                      # envs spread over the track, some episodes over
                env.step(acts[t], out=out)
            rows = [acts[t] for t in range(T)]
            it = iter(range(10 ** 9))
            best = min(timed(lambda: env.step(rows[next(it) % T], out=out), REPS) for _ in range(3))
            res[name] = (best, env.last_step_kernel())
            if form == 2:
                outm = (torch.empty(T, N, env.obs_dim, device="cuda"), torch.empty(T, N, device="cuda"), torch.empty(T, N, device="cuda"),
                        torch.empty(T, N, device="cuda"))
                env.step_many(acts, out=outm)
                best = min(timed(lambda: env.step_many(acts, out=outm), 4) for _ in range(3)) / T
                res["many"] = (best, env.last_step_kernel())
                del outm
            env.close()
        print(f"{dtype} rays {rays} N {N:7d}: " + "  ".join(f"{k} {v[0]:7.2f} us ({v[1]})" for k, v in res.items()) +
              f"  | env-only {N / res['many'][0]:.0f} M env-steps/s in one launch, {N / min(res['K1'][0], res['K1f'][0]):.0f} M step by step", flush=True)
