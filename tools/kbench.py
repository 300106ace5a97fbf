"""Per-kernel timing probe (developer tool): back-to-back launches of one kernel, HIP events."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ppo_car_amd as pc  # noqa: E402


def timeit(fn, n=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def k5(N, D=23):
    agent = pc.Agent(D, 9).cuda()
    x = torch.rand(N, D, device="cuda")
    a = torch.empty(N, dtype=torch.int64, device="cuda")
    lp, v, af = (torch.empty(N, device="cuda") for _ in range(3))
    agent.pack_policy()
    us = timeit(lambda: agent.act(x, out_action=a, out_logprob=lp, out_value=v, out_action_f32=af, repack=False))
    flops = 2 * (D * 512 + 256 * 10) * N
    print(f"K5 policy N={N} D={D}: {us:.2f} us  ({flops / us / 1e6:.1f} TFLOP/s useful)", flush=True)


def k1(N, n=16, dtype="f32", lanes=0):
    env = pc.VecCarEnv(N, f"{ROOT}/tracks/big_track.json", num_rays=n, reward_scaling=0.1, dtype=dtype)
    obs, _ = env.reset()
    if lanes:
        env.set_lanes_per_env(lanes)
    g = torch.Generator(device="cuda").manual_seed(0)
    acts = torch.randint(0, 9, (64, N), device="cuda", generator=g)
    out = (obs, torch.empty(N, device="cuda"), torch.empty(N, device="cuda"), torch.empty(N, device="cuda"))
    i = [0]

    def f():
        env.step(acts[i[0] & 63], out=out)
        i[0] += 1
    us = timeit(f)
    print(f"K1 env N={N} n={n} {dtype} {env.launch_info()}: {us:.2f} us -> {N / us / 1e3:.2f} G env-steps/s", flush=True)
    env.close()


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "k5"):
        for N in (4096, 65536, 262144):
            k5(N)
        k5(65536, 18)
        k5(65536, 39)
    if which in ("all", "k1"):
        for N in (4096, 65536, 524288):
            k1(N)
        for lanes in (1, 2, 4, 8):
            k1(65536, lanes=lanes)
        k1(65536, n=32)
        k1(65536, n=12)
        k1(65536, dtype="f64")
