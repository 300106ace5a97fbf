"""Quick GPU sanity + timing probe (developer tool, not part of the product or the tests)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402  (checker only)
import ppo_car_amd as pc  # noqa: E402

G = f"{ROOT}/tests/golden"
STATE = ("px", "py", "vx", "vy", "rot", "time_step", "next_gate", "passed")


def teacher(track, n, dtype):
    g = np.load(f"{G}/env_{track}_n{n}.npz")
    for grp in ("long", "short"):
        T, N = g[f"{grp}_action"].shape
        M = T * N
        env = pc.VecCarEnv(M, f"{ROOT}/tracks/{track}.json", num_rays=n, reward_scaling=0.1, dtype=dtype)
        env.reset()
        env.set_state(**{k: g[f"{grp}_pre_{k}"].reshape(-1) for k in STATE})
        fin = torch.empty(M, env.obs_dim, device="cuda")
        gp = torch.empty(M, dtype=torch.int32, device="cuda")
        obs, rew, term, trunc, _ = env.step(torch.from_numpy(g[f"{grp}_action"].reshape(-1)).cuda(), final_obs=fin, gates_passed=gp)
        torch.cuda.synchronize()
        fin = fin.cpu().numpy()
        ref = g[f"{grp}_step_obs"].reshape(M, -1)
        term_ref = g[f"{grp}_terminated"].reshape(-1)
        trunc_ref = g[f"{grp}_truncated"].reshape(-1)
        rew_ref = g[f"{grp}_reward_scaled"].reshape(-1).astype(np.float32)
        err = np.abs(fin - ref)
        mm_term = (term.cpu().numpy() != term_ref)
        mm_trunc = (trunc.cpu().numpy() != trunc_ref)
        mm_rew = rew.cpu().numpy() != rew_ref
        wm, gm = g[f"{grp}_wall_margin"].reshape(-1), g[f"{grp}_gate_margin"].reshape(-1)
        mm_pass = gp.cpu().numpy() != g[f"{grp}_post_passed"].reshape(-1)
        print(f"{track} n={n} {dtype} {grp}: M={M} obs max err {err.max():.3e} (n>1e-5: {(err > 1e-5).sum()}, n!=0: {(err != 0).sum()}) "
              f"term mism {mm_term.sum()} (margin>1e-3: {(mm_term & (wm > 1e-3)).sum()}) trunc mism {mm_trunc.sum()} "
              f"gate mism {mm_pass.sum()} (margin>1e-3: {(mm_pass & (gm > 1e-3)).sum()}) rew mism {mm_rew.sum()}", flush=True)
        if err.max() > 1e-5:
            i, j = np.unravel_index(err.argmax(), err.shape)
            print("   worst", i, j, fin[i, j], ref[i, j])
        env.close()


def timing(N, n, dtype, lanes_list, steps=50):
    env = pc.VecCarEnv(N, f"{ROOT}/tracks/big_track.json", num_rays=n, reward_scaling=0.1, dtype=dtype)
    obs, _ = env.reset()
    g = torch.Generator(device="cuda").manual_seed(0)
    acts = torch.randint(0, 9, (steps, N), device="cuda", generator=g)
    out = (obs, torch.empty(N, device="cuda"), torch.empty(N, device="cuda"), torch.empty(N, device="cuda"))
    for lanes in lanes_list:
        try:
            env.set_lanes_per_env(lanes)
        except pc.PpoCarError as e:
            print("  lanes", lanes, "->", e)
            continue
        for t in range(5):
            env.step(acts[t], out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for t in range(steps):
            env.step(acts[t], out=out)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / steps
        print(f"  N={N} n={n} {dtype} lanes={lanes} {env.launch_info()} : {us:.2f} us/step -> {N / us:.1f} M env-steps/s", flush=True)
    env.close()


if __name__ == "__main__":
    print(torch.cuda.get_device_name(0), flush=True)
    for dtype in ("f64", "f32"):
        for track, n in (("big_track", 16), ("big_track", 12), ("track", 32)):
            teacher(track, n, dtype)
    for N, lanes in ((65536, (0, 1, 2, 4, 8, 16)), (4096, (0, 4, 8, 16, 32)), (524288, (0, 1, 2, 4))):
        timing(N, 16, "f32", lanes)
    timing(65536, 16, "f64", (0, 2, 4, 8))
    timing(65536, 32, "f32", (0, 1, 2, 4, 8))
    timing(65536, 12, "f32", (0, 1, 2, 4))
