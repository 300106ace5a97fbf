#!/usr/bin/env python3
"""evaluate.py -- headless evaluation of a trained policy: the episode loop of the reference's
log_video (train.py:23-50) without the renderer: one env, actions from the agent, until the car is destroyed
(`done = terminated`, train.py:45 -- a truncation does not end the loop there; here the episode limit does).

    python evaluate.py --checkpoint checkpoints/<run>/model.dat --track tracks/big_track.json [--num-rays 12] [--episodes 5]
                       [--frames DIR [--frame-every 5]]      # PNG frames of episode 0 (software rasteriser, no pygame)
"""
import argparse
import json

import torch


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--checkpoint", required=True, help="agent state_dict as train.py saves it (checkpoint_<n>.dat / model.dat)")
    ap.add_argument("--track", default="tracks/big_track.json")
    ap.add_argument("--num-rays", type=int, default=12)
    ap.add_argument("--episodes", type=int, default=5)
    ap.add_argument("--greedy", action="store_true", help="argmax instead of sampling (the reference samples)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--frames", default=None, help="directory for PNG frames of episode 0 (the role of log_video's frames)")
    ap.add_argument("--frame-every", type=int, default=5)
    args = ap.parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("evaluate.py needs the GPU: the env has no CPU path")
    import ppo_car_amd as pc
    torch.manual_seed(args.seed)
    env = pc.VecCarEnv(args.episodes, args.track, num_rays=args.num_rays, reward_scaling=1.0, device="cuda")
    agent = pc.Agent(env.obs_dim, env.act_dim).cuda()
    agent.load_state_dict(torch.load(args.checkpoint, map_location="cuda"))
    obs, _ = env.reset()
    N = args.episodes
    alive = torch.ones(N, dtype=torch.bool, device="cuda")
    ret = torch.zeros(N, device="cuda")
    steps = torch.zeros(N, dtype=torch.int64, device="cuda")
    gates = torch.zeros(N, dtype=torch.int32, device="cuda")
    gp = torch.empty(N, dtype=torch.int32, device="cuda")
    frames = 0
    if args.frames:
        import os
        from ppo_car_amd.env import Track
        from ppo_car_amd.render import rasterise, write_png
        os.makedirs(args.frames, exist_ok=True)
        walls, gate_segs = Track(args.track).geometry()
    with torch.no_grad():
        for t in range(1000):               # CarEnv's own time limit (car_env.py:491)
            if args.frames and bool(alive[0]) and t % args.frame_every == 0:
                st = env.get_state()        # (synchronous test hook: fine for a viewer)
                rot = float(st["rot"][0])
                img = rasterise(walls, gate_segs, float(st["px"][0]), float(st["py"][0]), rot, obs[0, 6:].cpu().numpy(),
                                next_gate=int(st["next_gate"][0]), num_rays_nominal=args.num_rays)
                write_png(os.path.join(args.frames, f"frame_{frames:05d}.png"), img)
                frames += 1
            if args.greedy:
                action = agent.actor(obs).argmax(-1)
            else:
                action, _, _, _ = agent.get_action_and_value(obs)
            obs, r, term, trunc, _ = env.step(action, gates_passed=gp)
            ret += torch.where(alive, r, torch.zeros_like(r))
            steps += alive.long()
            gates = torch.where(alive, gp, gates)
            alive &= ~((term + trunc) > 0)
            if not bool(alive.any()):
                break
    out = {"episodes": N, "mean_return": float(ret.mean()), "mean_steps": float(steps.float().mean()),
           "mean_gates_passed": float(gates.float().mean()), "returns": ret.tolist(), "gates_passed": gates.tolist()}
    if args.frames:
        out["frames"] = frames
    print(json.dumps(out))
    return out


if __name__ == "__main__":
    main()
