#!/usr/bin/env python3
"""Golden-vector generator for the CarEnv hot path.  NOT shipped to the GPU box as code
that runs there: it reads /root/reference at run time and only its OUTPUT (the .npz
fixtures next to it) is used by the tests.

It imports the reference environment *unmodified* (lib/car_env.py, lib/buffer.py) and
records inputs / outputs.  gymnasium and pygame are absent from this image, so three
inert stand-in modules are injected into sys.modules before the import.  They carry
NO arithmetic of the path: `gym.Env` (empty base class whose reset() is a no-op; the
reference calls super().reset(seed=seed) at car_env.py:617 and never uses the RNG),
`spaces.Box/Discrete` (plain containers, car_env.py:522-525), `gym.register`
(car_env.py:816, no-op) and `pygame.image.load/transform.scale/get_rect`
(car_env.py:250-255, sprite only).

gymnasium's AsyncVectorEnv / TransformReward (train.py:53-69,138-140,185) are not in
/root/reference; their two semantics on this path (same-step auto-reset, reward *
reward_scaling) are restated in `run_group` below from the gymnasium 0.29.1 behaviour
(requirements.txt:5) -- that part of the fixtures is "parity unpinned" by the reference.

Run:  python tests/golden/make_golden.py          (≈ 3 min, writes tests/golden/*.npz)
      python tests/golden/make_golden.py ppo      (seconds: tests/golden/ppo_minibatch.npz only -- lib/model.py's Agent)
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _inject_stubs():
    gym = types.ModuleType("gymnasium")

    class Env:
        def reset(self, seed=None, options=None):
            return None

    gym.Env = Env
    gym.register = lambda **k: None
    spaces = types.ModuleType("gymnasium.spaces")

    class Box:
        def __init__(self, low, high, dtype=None):
            self.low, self.high, self.shape, self.dtype = low, high, low.shape, dtype

    class Discrete:
        def __init__(self, n):
            self.n = n

    spaces.Box, spaces.Discrete = Box, Discrete
    gym.spaces = spaces
    pg = types.ModuleType("pygame")

    class Surface:
        pass

    class _Img:
        def get_rect(self, **k):
            return None

    pg.Surface = Surface
    pg.image = types.SimpleNamespace(load=lambda p: _Img())
    pg.transform = types.SimpleNamespace(scale=lambda im, sz: im)
    sys.modules.update({"gymnasium": gym, "gymnasium.spaces": spaces, "pygame": pg})


_inject_stubs()
sys.path.insert(0, REF)
from lib.car_env import Boundary, Car, CarEnv, Ray  # noqa: E402

REPO = os.path.dirname(os.path.dirname(OUT))
# the reference's two tracks, plus a generated 128-wall circuit of this repository (python -m ppo_car_amd.track_tool make-oval
# tracks/oval64.json --points 64 --gates 40 --wobble 0.01 --seed 0): the reference runs it like any other track file
TRACKS = {"big_track": f"{REF}/tracks/big_track.json", "track": f"{REF}/tracks/track.json", "oval64": f"{REPO}/tracks/oval64.json"}
REWARD_SCALING = 0.1  # train.py:90 default


# --------------------------------------------------------------------------------------
# access to the reference's private state (name-mangled attributes)
# --------------------------------------------------------------------------------------
def make_env(track, n):
    env = CarEnv()
    if n != 12:
        env._CarEnv__car = Car(0, 0, num_rays=n)  # the only way to get n != 12 (car_env.py:505-509)
    env.reset(options={"track_path": TRACKS[track]})
    return env


def get_state(env):
    car = env._CarEnv__car
    p, v = car._Car__pos, car._Car__velocity
    return (float(p[0]), float(p[1]), float(v[0]), float(v[1]), float(car._Car__rotation),
            int(env._CarEnv__time_step), int(env._CarEnv__next_gate_index),
            int(env._CarEnv__passed_reward_gates))


def collision_ray_dists(env, boundary):
    car = env._CarEnv__car
    n = car._Car__num_rays
    return [car._Car__rays[r].get_distance(boundary) for r in range(0, n, n // 4)]


def chase_action(env, rng, eps, vtarget):
    """A crude gate-chasing driver, only there to reach gates, laps and the 1000-step
    truncation with a moving car.  Test tooling, not reference code."""
    if rng.random() < eps:
        return int(rng.integers(0, 9))
    st = get_state(env)
    gates = env._CarEnv__reward_gates
    a, b = gates[st[6]].get_points()
    a2, b2 = gates[(st[6] + 1) % len(gates)].get_points()
    tgt = 0.25 * (a + b) + 0.25 * (a2 + b2)
    des = np.degrees(np.arctan2(tgt[1] - st[1], tgt[0] - st[0]))
    diff = (des - st[4] + 180.0) % 360.0 - 180.0
    thrust = np.hypot(st[2], st[3]) < vtarget
    if diff > 2.5:
        return 5 if thrust else 3
    if diff < -2.5:
        return 4 if thrust else 2
    return 0 if thrust else 8


STATE_KEYS = ("px", "py", "vx", "vy", "rot", "time_step", "next_gate", "passed")


def run_group(track, n, policies, T, seed):
    """Free-running reference envs, one per policy, with gymnasium-0.29.1-style same-step
    auto-reset (restated).  Records every transition teacher-forcing needs."""
    N = len(policies)
    envs = [make_env(track, n) for _ in range(N)]
    D = len(envs[0]._get_obs())
    rngs = [np.random.default_rng(seed * 1000 + i) for i in range(N)]
    rec = {f"pre_{k}": np.zeros((T, N), np.float64 if i < 5 else np.int64) for i, k in enumerate(STATE_KEYS)}
    rec.update({f"post_{k}": np.zeros((T, N), np.float64 if i < 5 else np.int64) for i, k in enumerate(STATE_KEYS)})
    rec["action"] = np.zeros((T, N), np.int64)
    rec["step_obs"] = np.zeros((T, N, D), np.float32)   # obs returned by CarEnv.step itself
    rec["ret_obs"] = np.zeros((T, N, D), np.float32)    # what the vector env hands back (reset obs if done)
    rec["reward"] = np.zeros((T, N), np.float64)        # CarEnv.step reward, unscaled
    rec["reward_scaled"] = np.zeros((T, N), np.float64)  # TransformReward: r * 0.1
    rec["terminated"] = np.zeros((T, N), np.bool_)
    rec["truncated"] = np.zeros((T, N), np.bool_)
    rec["gate_margin"] = np.zeros((T, N), np.float64)   # min_r |d_r(gate[next]) - 10| at the pre-step pose
    rec["wall_margin"] = np.zeros((T, N), np.float64)   # min_r |d_r(walls) - 10| at the post-step pose
    for t in range(T):
        for i, (env, pol) in enumerate(zip(envs, policies)):
            pre = get_state(env)
            gates = env._CarEnv__reward_gates
            gd = collision_ray_dists(env, gates[pre[6]])
            kind = pol[0]
            if kind == "const":
                a = pol[1]
            elif kind == "random":
                a = int(rngs[i].integers(0, 9))
            else:
                a = chase_action(env, rngs[i], pol[1], pol[2])
            obs, r, term, trunc, _ = env.step(np.int64(a))
            post = get_state(env)
            wd = collision_ray_dists(env, env._CarEnv__boundaries)
            for k, vpre, vpost in zip(STATE_KEYS, pre, post):
                rec[f"pre_{k}"][t, i] = vpre
                rec[f"post_{k}"][t, i] = vpost
            rec["action"][t, i] = a
            rec["step_obs"][t, i] = obs
            rec["reward"][t, i] = r
            rec["reward_scaled"][t, i] = r * REWARD_SCALING
            rec["terminated"][t, i] = term
            rec["truncated"][t, i] = trunc
            rec["gate_margin"][t, i] = min(abs(d - 10.0) for d in gd)
            rec["wall_margin"][t, i] = min(abs(d - 10.0) for d in wd)
            if term or trunc:  # gymnasium 0.29.1 AsyncVectorEnv worker: reset in the same call
                obs, _ = env.reset()
            rec["ret_obs"][t, i] = obs
    return rec


def ray_unit_cases(seed=7, m=1500):
    """Ray.get_distance on single segments: random + degenerate (parallel, endpoint, behind)."""
    rng = np.random.default_rng(seed)
    cases = []

    def add(px, py, ang, x1, y1, x2, y2):
        d = Ray(px, py, ang).get_distance(Boundary(x1, y1, x2, y2))
        cases.append((px, py, ang, x1, y1, x2, y2, d))

    for _ in range(m):
        px, py = rng.uniform(0, 1280), rng.uniform(0, 720)
        ang = rng.uniform(-4000, 4000)
        x1, y1, x2, y2 = rng.uniform(0, 1280), rng.uniform(0, 720), rng.uniform(0, 1280), rng.uniform(0, 720)
        add(px, py, ang, x1, y1, x2, y2)
    # parallel (den == 0 exactly): horizontal ray, horizontal segment
    add(100.0, 100.0, 0.0, 200.0, 150.0, 400.0, 150.0)
    add(100.0, 100.0, 0.0, 200.0, 100.0, 400.0, 100.0)      # collinear
    add(100.0, 100.0, 180.0, 200.0, 150.0, 400.0, 150.0)
    # ray through an endpoint: t == 0 / t == 1 exactly (strict test rejects)
    add(100.0, 100.0, 0.0, 300.0, 100.0, 300.0, 200.0)      # hits p1 exactly (t == 0)
    add(100.0, 100.0, 0.0, 300.0, 0.0, 300.0, 100.0)        # hits p2 exactly (t == 1)
    add(100.0, 100.0, 0.0, 300.0, 50.0, 300.0, 150.0)       # mid hit, d = 200
    add(100.0, 100.0, 180.0, 300.0, 50.0, 300.0, 150.0)     # behind the ray (u < 0)
    add(100.0, 100.0, 90.0, 50.0, 1300.0, 150.0, 1300.0)    # farther than 1000 -> 1000
    add(100.0, 100.0, 90.0, 50.0, 1099.5, 150.0, 1099.5)    # 999.5
    add(100.0, 100.0, 45.0, 100.0, 100.0, 200.0, 100.0)     # ray origin on the segment end (u == 0)
    add(100.0, 100.0, 360.0 * 11 + 90.0, 0.0, 110.0, 200.0, 110.0)   # d = 10 (collision threshold), unreduced angle
    add(100.0, 100.0, 90.0, 0.0, 110.0, 200.0, 110.0)
    a = np.array(cases, np.float64)
    return {"px": a[:, 0], "py": a[:, 1], "angle": a[:, 2], "x1": a[:, 3], "y1": a[:, 4],
            "x2": a[:, 5], "y2": a[:, 6], "dist": a[:, 7]}


def gae_cases():
    """Buffer.calculate_advantages (buffer.py:36-64) on seeded random inputs, CPU torch."""
    import torch
    from lib.buffer import Buffer
    out = {}
    for ci, (T, N, p_term, p_trunc, seed) in enumerate([(64, 24, 0.02, 0.01, 0), (1024, 24, 0.015, 0.002, 1),
                                                        (17, 5, 0.3, 0.2, 2), (1, 3, 0.5, 0.5, 3)]):
        g = torch.Generator().manual_seed(seed)
        buf = Buffer((4,), T, N, torch.device("cpu"), 0.99, 0.95)
        rew = torch.randn(T, N, generator=g) * 0.3
        val = torch.randn(T, N, generator=g)
        term = (torch.rand(T, N, generator=g) < p_term).float()
        trunc = ((torch.rand(T, N, generator=g) < p_trunc).float()) * (1 - term)
        if T > 1:
            term[-1, 0] = 1.0
            trunc[-1, 1] = 1.0
        for t in range(T):
            buf.store(torch.zeros(N, 4), torch.zeros(N), rew[t], val[t], term[t], trunc[t], torch.zeros(N))
        last_val = torch.randn(1, N, generator=g)
        last_term = (torch.rand(1, N, generator=g) < 0.3).float()
        last_trunc = (torch.rand(1, N, generator=g) < 0.3).float() * (1 - last_term)
        adv, ret = buf.calculate_advantages(last_val, last_term, last_trunc)
        for k, v in dict(rew=rew, val=val, term=term, trunc=trunc, last_val=last_val, last_term=last_term,
                         last_trunc=last_trunc, adv=adv, ret=ret).items():
            out[f"c{ci}_{k}"] = v.numpy()
    out["n_cases"] = np.int64(4)
    out["gamma"] = np.float64(0.99)
    out["lam"] = np.float64(0.95)
    return out


def ppo_cases():
    """SURVEY 8(c) fixture 6: ONE seeded PPO minibatch through the reference's own `Agent` (lib/model.py:12-41, imported
    unmodified; `layer_init` :6-9 runs inside its constructor) -> logits, log-probs, entropy, value, the three loss terms, the
    total, every gradient, the pre-clip gradient norm, and the parameters after the one optimizer step of train.py:257-261.
    train.py itself cannot be imported (tkinter / cv2 / tensorboardX / gymnasium at module level, a script body under
    __main__): its minibatch expression train.py:233-261 is restated below LINE BY LINE with the reference's own names, on
    the reference's Agent / torch.optim.Adam(lr=3e-4, eps=1e-5) (train.py:146).  The WEIGHTS are stored in the fixture, so the
    consumer does not depend on this torch version's orthogonal_ RNG stream."""
    import torch
    import torch.nn as nn
    from lib.model import Agent
    out = {}
    clip_ratio, vf_coef, ent_coef, max_grad_norm, lr = 0.2, 0.5, 0.001, 1.0, 3e-4      # train.py:84-89 defaults
    # (track, nominal rays, minibatch size, seed, scale of the returns: case 1's gradient norm exceeds max_grad_norm, so the clip acts)
    cases = [("big_track", 16, 512, 0, 1.0), ("big_track", 12, 512, 1, 20.0), ("track", 32, 64, 2, 1.0)]
    for ci, (track, n, B, seed, ret_scale) in enumerate(cases):
        torch.manual_seed(1234 + seed)
        g = torch.Generator().manual_seed(seed)
        rec = np.load(f"{OUT}/env_{track}_n{n}.npz")
        pool = torch.from_numpy(rec["short_ret_obs"].reshape(-1, rec["short_ret_obs"].shape[-1]))   # observations the reference env produced
        pool = pool[torch.randperm(pool.shape[0], generator=g)[:2048]].contiguous()
        D = pool.shape[1]
        agent = Agent(D, 9)                                         # train.py:145
        # a few SGD-free perturbations so that the output layers are not the near-zero init (std 0.01) only
        with torch.no_grad():
            for p_ in agent.parameters():
                p_.add_(0.05 * torch.randn(p_.shape, generator=g))
        optimizer = torch.optim.Adam(agent.parameters(), lr=lr, eps=1e-5)   # train.py:146
        weights = {k: v.detach().clone().numpy() for k, v in agent.state_dict().items()}
        M = pool.shape[0]
        batch_indices = torch.randperm(M, generator=g)[:B]
        traj_obs = pool
        traj_act = torch.randint(0, 9, (M,), generator=g).float()                  # actions are stored as float32 (buffer.py:13)
        with torch.no_grad():
            _, lp_now, _, _ = agent.get_action_and_value(traj_obs, traj_act)
        traj_logprob = lp_now + 0.25 * torch.randn(M, generator=g)                # ratios on both sides of the clip range
        traj_adv = torch.randn(M, generator=g) * 2.0 + 0.3
        traj_ret = torch.randn(M, generator=g) * ret_scale
        device = torch.device("cpu")
        # ---- train.py:233-261, verbatim semantics
        _, new_logprobs, entropies, new_values = agent.get_action_and_value(traj_obs[batch_indices], traj_act[batch_indices])
        ratios = torch.exp(new_logprobs - traj_logprob[batch_indices])
        batch_adv = traj_adv[batch_indices]
        batch_adv = (batch_adv - batch_adv.mean()) / torch.max(batch_adv.std(), torch.tensor(1e-5, device=device))
        policy_loss1 = -batch_adv * ratios
        policy_loss2 = -batch_adv * torch.clamp(ratios, 1.0 - clip_ratio, 1.0 + clip_ratio)
        policy_loss = torch.max(policy_loss1, policy_loss2).mean()
        new_values = new_values.view(-1)
        value_loss = 0.5 * ((new_values - traj_ret[batch_indices]) ** 2).mean()
        entropy = entropies.mean()
        loss = policy_loss + vf_coef * value_loss - ent_coef * entropy
        optimizer.zero_grad()
        loss.backward()
        grads = {k: p_.grad.detach().clone().numpy() for k, p_ in agent.named_parameters()}
        grad_norm = nn.utils.clip_grad_norm_(agent.parameters(), max_grad_norm)   # returns the norm BEFORE clipping
        optimizer.step()
        logits = agent.actor(traj_obs[batch_indices]).detach()
        pre = f"c{ci}_"
        out.update({pre + "w_" + k: v for k, v in weights.items()})
        out.update({pre + "g_" + k: v for k, v in grads.items()})
        out.update({pre + "p1_" + k: v.detach().numpy() for k, v in agent.state_dict().items()})
        # NB `logits` above was taken AFTER the step; the pre-step logits are recomputed from the stored weights
        ref = Agent(D, 9)
        ref.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()})
        with torch.no_grad():
            logits0 = ref.actor(traj_obs[batch_indices])
        out.update({pre + "obs": traj_obs.numpy(), pre + "act": traj_act.numpy(), pre + "old_logprob": traj_logprob.numpy(),
                    pre + "adv": traj_adv.numpy(), pre + "ret": traj_ret.numpy(), pre + "idx": batch_indices.numpy().astype(np.int64),
                    pre + "logits": logits0.numpy(), pre + "new_logprob": new_logprobs.detach().numpy(),
                    pre + "entropies": entropies.detach().numpy(), pre + "new_values": new_values.detach().numpy(),
                    pre + "ratios": ratios.detach().numpy(), pre + "batch_adv_norm": batch_adv.detach().numpy(),
                    pre + "policy_loss": np.float32(policy_loss.item()), pre + "value_loss": np.float32(value_loss.item()),
                    pre + "entropy": np.float32(entropy.item()), pre + "loss": np.float32(loss.item()),
                    pre + "grad_norm": np.float32(float(grad_norm)), pre + "num_rays_nominal": np.int64(n)})
        print(f"ppo case {ci}: D={D} B={B} loss {loss.item():.6f} = {policy_loss.item():.6f} + {vf_coef}*{value_loss.item():.6f} - "
              f"{ent_coef}*{entropy.item():.6f}; grad norm {float(grad_norm):.4f}; clipped share "
              f"{float(((ratios < 0.8) | (ratios > 1.2)).float().mean()):.2f}", flush=True)
    out.update({"n_cases": np.int64(len(cases)), "clip_ratio": np.float64(clip_ratio), "vf_coef": np.float64(vf_coef),
                "ent_coef": np.float64(ent_coef), "max_grad_norm": np.float64(max_grad_norm), "lr": np.float64(lr),
                "adam_eps": np.float64(1e-5), "torch_version": np.array(torch.__version__)})
    return out


def main():
    os.chdir(REF)  # the reference loads "lib/assets/car.png" relative to cwd (stubbed, but keep the cwd it expects)
    if sys.argv[1:] == ["ppo"]:               # `make_golden.py ppo`: only the PPO-minibatch fixture (reads the env fixtures for its observations)
        np.savez_compressed(f"{OUT}/ppo_minibatch.npz", **ppo_cases())
        return
    if not sys.argv[1:]:
        np.savez_compressed(f"{OUT}/ray_cases.npz", **ray_unit_cases())
        np.savez_compressed(f"{OUT}/gae_cases.npz", **gae_cases())
    const = [("const", a) for a in range(9)]
    only = sys.argv[1:]                       # e.g. `make_golden.py oval64` regenerates that track's fixtures only
    for track in ("big_track", "track", "oval64"):
        if only and track not in only:
            continue
        for n in ((16,) if track == "oval64" else (12, 16, 32)):
            main_cfg = (track == "big_track" and n == 16)
            env = make_env(track, n)
            reset_obs = env._get_obs()
            reset_state = np.array(get_state(env)[:5], np.float64)
            walls = np.array([np.concatenate(b.get_points()) for b in env._CarEnv__boundaries], np.float64)
            gates = np.array([np.concatenate(g.get_points()) for g in env._CarEnv__reward_gates], np.float64)
            # long group: truncation at 1000 steps (action 8, and a moving car), unreduced angle drift (action 2 / 3), a full lap
            long_pol = [("const", 8), ("const", 2), ("chase", 0.0, 5.0)] + ([("const", 3), ("chase", 0.02, 7.0)] if main_cfg else [])
            long_rec = run_group(track, n, long_pol, 1010, seed=1)
            # short group: every constant action, random policies, noisy drivers
            short_pol = const + [("random",)] * (6 if main_cfg else 3) + [("chase", e, v) for e, v in
                                                                          (((0.02, 4.0), (0.05, 6.0), (0.1, 8.0), (0.2, 10.0), (0.3, 5.0), (0.5, 7.0))
                                                                           if main_cfg else ((0.05, 6.0), (0.2, 9.0), (0.4, 5.0)))]
            short_rec = run_group(track, n, short_pol, 320 if main_cfg else 200, seed=2)
            out = {"reset_obs": reset_obs, "reset_state": reset_state, "walls": walls, "gates": gates,
                   "num_rays_nominal": np.int64(n), "reward_scaling": np.float64(REWARD_SCALING)}
            out.update({f"long_{k}": v for k, v in long_rec.items()})
            out.update({f"short_{k}": v for k, v in short_rec.items()})
            path = f"{OUT}/env_{track}_n{n}.npz"
            np.savez_compressed(path, **out)
            ng = int(long_rec["post_passed"].max()), int(short_rec["post_passed"].max())
            nterm = int(long_rec["terminated"].sum() + short_rec["terminated"].sum())
            ntrunc = int(long_rec["truncated"].sum() + short_rec["truncated"].sum())
            laps = int(((long_rec["reward"] > 10.5).sum()) + ((short_rec["reward"] > 10.5).sum()))
            print(f"{track} n={n}: D={len(reset_obs)} max gates passed {ng} term {nterm} trunc {ntrunc} "
                  f"laps {laps} -> {os.path.getsize(path) / 1e6:.2f} MB", flush=True)


if __name__ == "__main__":
    main()
