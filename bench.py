#!/usr/bin/env python3
"""bench.py -- headline benchmark: env steps/sec of the whole PPO loop (rollout + GAE + update, the
`charts/SPS` definition of the reference, train.py:174,292) on big_track.json, "16 rays" (17 actual).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload target|cfg1|cfg2]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

A "step" is ONE EPOCH of the hot path on one batch of synthetic input: n_steps vector-env steps of
n_envs envs per GPU (policy forward + sample + env-step kernel + store), the bootstrap value, the GAE
scan and train_iters x ceil(n_steps/batch_size) clipped-PPO minibatch updates (with the gradient
all-reduce when N > 1).  Nothing is skipped inside the timed region.  value = all ranks' env steps / the
slowest rank's time (weak scaling: n_envs per GPU is fixed).

Extra objects on the JSON line:
  roofline     -- the env-step kernel (K1): algorithmic bytes per launch (176 B per env step
                  at 16 rays) / its mean duration measured with HIP events on the launch stream inside the
                  timed epochs; peak = 8 TB/s HBM.  K1 is VALU-bound, so a second, informative
                  `valu` entry prices the same duration against the fp32 vector peak.
  cpu_baseline -- the CPU oracle (oracle/, the checker -- never the product) driving the same rollout on the
                  host cores of this box for a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # the size the BASELINE.json target is quoted on (>= 10 M env steps/s at n_envs = 65536, 16 rays) and the
    # per-GPU shard of configs[3] (524288 envs on 8 GPUs); n_steps / batch / iters = the reference defaults
    "target": dict(n_envs=65536, n_steps=1024, num_rays=16, batch_size=512, train_iters=40),
    # BASELINE.json configs[1]
    "cfg1": dict(n_envs=4096, n_steps=1024, num_rays=16, batch_size=512, train_iters=40),
    # BASELINE.json configs[2] (ray-kernel stress: 32 -> 33 rays)
    "cfg2": dict(n_envs=65536, n_steps=128, num_rays=32, batch_size=512, train_iters=40),
    # the per-GPU shard of BASELINE.json configs[4] (262144 envs on 8 GPUs, track.json and big_track.json in one batch)
    "cfg4": dict(n_envs=32768, n_steps=1024, num_rays=16, batch_size=512, train_iters=40, mixed=True),
}
ALGO_BYTES = {12: 156, 16: 176, 32: 240}      # SURVEY 8(d) / BASELINE.md section 4, per env step
ALGO_FLOPS = {12: 8200, 16: 11600, 32: 22400}  # ditto, big_track (24 wall segments)
HBM_PEAK_GBS = 8000.0                          # MI355X_MICROARCH.md: 8 TB/s
VALU_PEAK_TFLOPS = 157.3                       # fp32 vector peak


def cpu_baseline(cfg, budget_s=12.0):
    """Rollout of the same workload on the host: torch-CPU policy forward + the C oracle env on all host
    cores, bounded sample.  The PPO update is not included (it favours the CPU figure)."""
    import oracle
    from ppo_car_amd.model import Agent
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    n_envs = 4096
    env = oracle.OracleVecEnv(oracle.Track(os.path.join(ROOT, "tracks", "big_track.json")), n_envs,
                              num_rays=cfg["num_rays"], reward_scaling=0.1, threads=cores)
    obs = torch.from_numpy(env.reset())
    agent = Agent(env.D, 9)
    steps, t0 = 0, time.time()
    with torch.no_grad():
        while True:
            a, lp, _, v = agent.get_action_and_value(obs)
            o, r, te, tr = env.step(a.numpy())
            obs = torch.from_numpy(o)
            steps += 1
            if time.time() - t0 > budget_s and steps >= 4:
                break
    dt = time.time() - t0
    return {"value": n_envs * steps / dt, "unit": "env steps/s", "cores": cores, "kind": "port",
            "sample": f"rollout only (torch-CPU policy + C oracle env, {cores} threads), n_envs={n_envs}, {steps} steps, "
                      f"{cfg['num_rays']} rays, big_track; PPO update not included"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10, help="timed epochs")
    ap.add_argument("--warmup", type=int, default=2, help="untimed epochs")
    ap.add_argument("--workload", default="target", choices=sorted(WORKLOADS))
    ap.add_argument("--n-envs", type=int, default=None, help="override envs per GPU")
    ap.add_argument("--n-steps", type=int, default=None)
    ap.add_argument("--env-dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--policy", default="fused", choices=["fused", "sample", "torch"], help="rollout policy-step implementation")
    ap.add_argument("--policy-arith", default="fp16x2", choices=["fp16x2", "bf16x3", "fp32"], help="arithmetic of the fused policy step's GEMMs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend (nccl = RCCL)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal on a 1-GPU box: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--eager-rollout", action="store_true", help="no rollout graph; bracket env-step launches with events instead")
    ap.add_argument("--rollout-kernel", default="auto", choices=["auto", "mega", "steps"], help="persistent rollout kernel or 2 kernels/step")
    ap.add_argument("--rollout-form", type=int, default=-1, choices=[-1, 0, 1, 2, 3], help="pc_rollout_set_form: -1 auto; 0/1 force the 32-env-wave / split form; 2/3 the same without the 1/den table in LDS (A/B knob)")
    ap.add_argument("--no-graphs", action="store_true", help="eager update and rollout")
    ap.add_argument("--torch-mlp", action="store_true", help="torch autograd GEMMs for the MLPs inside the minibatch step (fused loss/Adam kernels only)")
    ap.add_argument("--torch-update", action="store_true", help="reference torch ops for the whole minibatch step (no fused loss/Adam kernels)")
    ap.add_argument("--event-stride", type=int, default=8, help="with --eager-rollout: bracket every k-th env-step launch")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs `python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py ...`")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU path for the hot path")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    from ppo_car_amd.ppo import PPOConfig, Trainer
    from ppo_car_amd._capi import lib as _lib
    _lib.pc_policy_set_precision({"fp16x2": 2, "bf16x3": 1, "fp32": 0}[args.policy_arith])
    _lib.pc_rollout_set_form(args.rollout_form)
    POLICY_ARITH = {"fp16x2": "fp16x2 split (v = h + 2^-11 l), 3 products, two fp32 accumulators on the fp16 matrix cores (fp32-class; DESIGN.md section 5)",
                    "bf16x3": "bf16x3 split, 6 products, fp32 accumulate on the bf16 matrix cores (fp32-equivalent; DESIGN.md section 5)",
                    "fp32": "fp32-input MFMA (exact fp32 fmaf chain)"}[args.policy_arith]
    wl = dict(WORKLOADS[args.workload])
    if args.n_envs:
        wl["n_envs"] = args.n_envs
    if args.n_steps:
        wl["n_steps"] = args.n_steps
    mixed = wl.pop("mixed", False)
    track = ([os.path.join(ROOT, "tracks", "track.json"), os.path.join(ROOT, "tracks", "big_track.json")] if mixed
             else os.path.join(ROOT, "tracks", "big_track.json"))
    cfg = PPOConfig(track=track, env_dtype=args.env_dtype, seed=0, policy=args.policy, use_graphs=not args.no_graphs, fused_update=not args.torch_update, custom_mlp=not args.torch_mlp,
                    rollout_kernel=args.rollout_kernel, **wl)
    tr = Trainer(cfg, device=dev, rank=rank, world_size=world)
    tr.profile_stride = 0
    # K1 probe: a second env batch of the same size and launch geometry.  When the rollout runs as a replayed HIP
    # graph its kernels cannot be bracketed one by one, so inside the timed region (same stream) this twin is
    # stepped PROBE times back to back between two events with the trainer's latest actions.
    from ppo_car_amd.env import VecCarEnv
    PROBE = 32
    probe = VecCarEnv(cfg.n_envs, os.path.join(ROOT, "tracks", "big_track.json"), num_rays=cfg.num_rays, reward_scaling=cfg.reward_scaling, device=dev, dtype=cfg.env_dtype)
    p_obs, _ = probe.reset()
    p_out = (p_obs, torch.empty(cfg.n_envs, device=dev), torch.empty(cfg.n_envs, device=dev), torch.empty(cfg.n_envs, device=dev))
    probe_events = []

    def run_probe():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(PROBE):
            probe.step(tr.actions, out=p_out)
        e1.record()
        probe_events.append((e0, e1))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 2 if cfg.use_graphs else 0)):   # graphs: 1 eager epoch, then capture, then replay
        tr.run_epoch(sync=False)
    run_probe()
    probe_events.clear()
    if args.eager_rollout:
        tr.profile_stride = args.event_stride
    tr.k1_events = []
    tr.phase_events = []
    tr.mega_events = []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tr.run_epoch(sync=False)
    barrier()
    dt = time.perf_counter() - t0
    for _ in range(args.steps):          # the stand-alone K1 probe, outside the timed region (it is not part of the path)
        run_probe()
    torch.cuda.synchronize()
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    k1_us = float(np.mean([a.elapsed_time(b) for a, b in probe_events]) * 1e3 / PROBE)
    k1_bracketed_us = float(np.mean([a.elapsed_time(b) for a, b in tr.k1_events]) * 1e3) if tr.k1_events else None
    probe.close()
    info = tr.envs.launch_info()
    split = {"rollout_ms": float(np.mean([e[0].elapsed_time(e[1]) for e in tr.phase_events])),
             "gae_update_ms": float(np.mean([e[1].elapsed_time(e[2]) for e in tr.phase_events]))}
    tr.close()

    if rank == 0:
        env_steps = cfg.n_envs * cfg.n_steps * args.steps * world
        nr = cfg.num_rays
        per_step_bytes = ALGO_BYTES.get(nr, 4 * (tr.obs_dim[0]) + 84)     # algorithmic bytes per env step
        per_step_flops = ALGO_FLOPS.get(nr, 0)
        k1 = {"kernel": "env_step_kernel (K1), stand-alone", "launch_us": k1_us, "achieved": per_step_bytes * cfg.n_envs / (k1_us * 1e-6) / 1e9,
              "unit": "GB/s", "frac": per_step_bytes * cfg.n_envs / (k1_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
              "launch_us_method": f"{PROBE} back-to-back launches on an identical env batch between two HIP events on the launch stream, "
                                  "after the timed epochs", "launch_us_bracketed_in_rollout": k1_bracketed_us}
        if tr.mega_events:
            # the timed region's dominant kernel is the persistent rollout kernel: ONE launch does n_steps env steps (+ policy steps)
            # for every env; algorithmic bytes = SURVEY's per-env-step figure x n_envs x n_steps
            dom_us = float(np.mean([a.elapsed_time(b) for a, b in tr.mega_events]) * 1e3)
            algo_bytes = per_step_bytes * cfg.n_envs * cfg.n_steps
            dom_name = "rollout_kernel (K9: policy step + env step + Buffer.store for all n_steps, one persistent launch)"
            dom_method = "HIP events on the launch stream around each pc_rollout launch inside the timed epochs"
            units = cfg.n_envs * cfg.n_steps
        else:
            dom_us, algo_bytes, dom_name, dom_method, units = k1_us, per_step_bytes * cfg.n_envs, k1["kernel"], k1["launch_us_method"], cfg.n_envs
        achieved = algo_bytes / (dom_us * 1e-6) / 1e9
        out = {
            "metric": "env steps/sec (whole node) on big_track.json, 16 rays",
            "value": env_steps / dt, "unit": "env steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.env_dtype == "f32" else "f64", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {'track.json + big_track.json (halves)' if mixed else 'big_track.json'}, num_rays={nr} ({tr.obs_dim[0] - 6} actual), "
                                   f"n_envs={cfg.n_envs}/GPU, n_steps={cfg.n_steps}, batch_size={cfg.batch_size}, "
                                   f"train_iters={cfg.train_iters}; one step = one PPO epoch (rollout + GAE + update)",
                       "n_envs_total": cfg.n_envs * world, "parallelism": f"env-sharded dp{world}, 1 flat grad all-reduce/minibatch",
                       "env_kernel": info, "policy_step": args.policy, "rollout": tr.rollout_mode,
                       "policy_gemm_arithmetic": POLICY_ARITH, "hip_graphs": bool(cfg.use_graphs),
                       "fused_update": bool(cfg.fused_update), "custom_mlp_update": bool(tr.learner.custom), "epoch_split": split,
                       "numerics": "float64 kinematic state, float32 ray geometry" if args.env_dtype == "f32"
                       else "float64 throughout (reference operation order)"},
            "roofline": {"kernel": dom_name, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "launch_us": dom_us, "launch_us_method": dom_method, "algorithmic_bytes_per_launch": algo_bytes,
                         "env_steps_per_launch": units,
                         "note": "the path is bound by the SIMDs' vector-issue port (fp32 VALU ray geometry + operand splitting), "
                                 "not by HBM: DESIGN.md section 4; `valu` prices the same launch against the fp32 vector peak",
                         "valu": {"achieved": per_step_flops * units / (dom_us * 1e-6) / 1e12, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": per_step_flops * units / (dom_us * 1e-6) / 1e12 / VALU_PEAK_TFLOPS,
                                  "counts": "SURVEY 8(d) env-step flops only (policy MFMA work not included)"},
                         "k1_standalone": k1},
        }
        traffic_file = os.path.join(ROOT, "profiles", "k1_traffic.json")
        if os.path.exists(traffic_file):
            try:
                tf = json.load(open(traffic_file))
                key = (f"rollout_{args.env_dtype}_n{nr}_N{cfg.n_envs}_T{cfg.n_steps}" if tr.mega_events
                       else f"{args.env_dtype}_n{nr}_N{cfg.n_envs}")
                if key in tf:
                    out["roofline"]["traffic"] = tf[key]["hbm_bytes_per_launch"]
                    out["roofline"]["traffic_source"] = tf[key].get("source", "profiles/k1_traffic.json")
            except Exception:
                pass
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
