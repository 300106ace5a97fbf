#!/usr/bin/env python3
"""train.py -- the reference's training CLI (reference train.py:72-92,114-301) on the MI355X-native
hot path.  Same flags and defaults; the Tk file dialog (train.py:95-111) is replaced by --track, and
the pieces outside the hot path (video, TensorBoard) are not reproduced: scalars go to a JSONL file
with the reference's tag names (train.py:286-292), checkpoints to the reference's file names
(train.py:280-283,301).

Single GPU:   python train.py --run-name demo --cuda --track tracks/big_track.json --n-envs 4096
Multi GPU:    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
                  train.py --run-name demo --cuda --n-envs 65536        (n-envs is per GPU)
"""
import argparse
import datetime
import json
import os
import time

import torch


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--run-name", required=True, help="Name of the run")
    p.add_argument("--cuda", default=False, action="store_true", help="Enable the GPU (required: the env has no CPU path)")
    p.add_argument("--env", default="CarEnv-v0", help="Environment to use (only CarEnv-v0)")
    p.add_argument("--n-envs", type=int, default=16, help="Number of environments (per GPU)")
    p.add_argument("--n-epochs", type=int, default=200, help="Number of epochs to run")
    p.add_argument("--n-steps", type=int, default=1024, help="Number of steps per epoch per environment")
    p.add_argument("--batch-size", type=int, default=512, help="Batch size")
    p.add_argument("--train-iters", type=int, default=40, help="Number of training iterations")
    p.add_argument("--gamma", type=float, default=0.99, help="Discount factor")
    p.add_argument("--gae-lambda", type=float, default=0.95, help="Lambda for GAE")
    p.add_argument("--clip-ratio", type=float, default=0.2, help="PPO clip ratio")
    p.add_argument("--ent-coef", type=float, default=0.001, help="Entropy coefficient")
    p.add_argument("--vf-coef", type=float, default=0.5, help="Value function coefficient")
    p.add_argument("--learning-rate", type=float, default=3e-4, help="Learning rate")
    p.add_argument("--learning-rate-decay", type=float, default=0.99, help="Multiply with lr every epoch")
    p.add_argument("--max-grad-norm", type=float, default=1.0, help="Maximum gradient norm")
    p.add_argument("--reward-scaling", type=float, default=0.1,
                   help="Scaling factor for the rewards for stable value function training")
    # additions (replaces the file dialog; exposes Car(num_rays))
    p.add_argument("--track", default="tracks/big_track.json", help="Track JSON (replaces the reference's file dialog)")
    p.add_argument("--num-rays", type=int, default=12, help="Car num_rays (reference hard-wires 12; 16 -> 17 rays, 32 -> 33)")
    p.add_argument("--env-dtype", default="f32", choices=["f32", "f64"], help="ray-geometry precision of the env kernel")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--full-sweep", action="store_true", help="use every sample each train iter instead of train.py:228's loop bound")
    p.add_argument("--policy-arith", default="fp16x2", choices=["fp16x2", "bf16x3", "fp32"],
                   help="arithmetic of the rollout policy step's GEMMs on the matrix cores: fp16x2 / bf16x3 operand splits (fp32-grade) or the exact fp32 chain")
    p.add_argument("--policy-range", default="fallback", choices=["fallback", "raise"],
                   help="weights outside the fp16x2 policy arithmetic's numeric domain (checked at every pack): switch to the exact fp32 chain, or raise")
    p.add_argument("--bootstrap-value", default="kernel", choices=["kernel", "fp32"],
                   help="agent.get_value(next_obs) for GAE (train.py:200): from inside the rollout launch, or torch's fp32 Linear")
    p.add_argument("--out-dir", default=".", help="where checkpoints/ and logs/ are created")
    p.add_argument("--lazy-logging", action="store_true", help="one rank: print and log one epoch behind the device (Trainer.run_epoch(sync=\"lazy\")) instead of "
                   "fetching every epoch's scalars before the next epoch is launched (the reference's order, the default: it costs ~1 %% at 65536 envs)")
    p.add_argument("--resume", default=None, help="trainer_<epoch>.pt written by an earlier run: continue it exactly")
    return p.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    if args.env != "CarEnv-v0":
        raise SystemExit("only CarEnv-v0 is implemented")
    if not (args.cuda and torch.cuda.is_available()):
        raise SystemExit("train.py: the CarEnv hot path runs on the GPU only -- pass --cuda on a machine with an AMD GPU")
    from ppo_car_amd.ppo import PPOConfig, Trainer

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    folder = f"{datetime.datetime.now().strftime('%Y-%m-%d_%H-%M-%S')}_{args.run_name}"       # train.py:118-123
    if world > 1:   # ONE folder name for the job: the ranks' clocks may straddle a second boundary
        import torch.distributed as dist
        box = [folder]
        dist.broadcast_object_list(box, src=0)
        folder = box[0]
    ckpt_dir = os.path.join(args.out_dir, "checkpoints", folder)
    log_dir = os.path.join(args.out_dir, "logs", folder)
    if rank == 0:
        os.makedirs(ckpt_dir, exist_ok=True)
        os.makedirs(log_dir, exist_ok=True)
        with open(os.path.join(log_dir, "hyperparameters.md"), "w") as f:      # train.py:132-135
            f.write("|param|value|\n|-|-|\n" + "\n".join(f"|{k}|{v}|" for k, v in vars(args).items()) + "\n")
        log = open(os.path.join(log_dir, "scalars.jsonl"), "w")

    cfg = PPOConfig(n_envs=args.n_envs, n_epochs=args.n_epochs, n_steps=args.n_steps, batch_size=args.batch_size,
                    train_iters=args.train_iters, gamma=args.gamma, gae_lambda=args.gae_lambda, clip_ratio=args.clip_ratio,
                    ent_coef=args.ent_coef, vf_coef=args.vf_coef, learning_rate=args.learning_rate,
                    learning_rate_decay=args.learning_rate_decay, max_grad_norm=args.max_grad_norm,
                    reward_scaling=args.reward_scaling, track=args.track, num_rays=args.num_rays, env_dtype=args.env_dtype,
                    seed=args.seed, full_sweep=args.full_sweep, bootstrap_value=args.bootstrap_value,
                    policy_precision={"fp16x2": 2, "bf16x3": 1, "fp32": 0}[args.policy_arith], policy_range=args.policy_range)
    trainer = Trainer(cfg, device=torch.device("cuda", local_rank), rank=rank, world_size=world)
    first_epoch = 1
    if args.resume:
        path = args.resume if world == 1 else args.resume.replace(".pt", f".rank{rank}.pt")
        trainer.load_state_dict(torch.load(path, map_location=trainer.device, weights_only=False))
        first_epoch = trainer.epoch + 1             # (charts/SPS keeps counting from the checkpoint's elapsed time)
    if rank == 0:
        print(trainer.agent.actor)      # train.py:148-149
        print(trainer.agent.critic)
    start = time.time()
    try:
        def report(ep, scalars):
            print(f"Epoch {ep} done in {time.time() - start:.2f}s. Avg reward: {scalars['charts/avg_reward']:.4f}. ", flush=True)   # train.py:275-276
            log.write(json.dumps(scalars) + "\n")
            log.flush()
        # --lazy-logging (one rank): the host runs one epoch AHEAD of what it prints (Trainer.run_epoch(sync="lazy")) -- the device never idles
        # while an epoch's scalars are fetched, printed and logged.  Measured at 65536 envs: 17.16 against 17.34 ms per epoch: off by default
        lazy = world == 1 and args.lazy_logging
        for epoch in range(first_epoch, args.n_epochs + 1):
            scalars = trainer.run_epoch(sync="lazy" if lazy else True)
            if rank == 0:
                if scalars is not None:
                    report(epoch - 1 if lazy else epoch, scalars)
                if epoch % 10 == 0:                                                                # train.py:280-283
                    torch.save(trainer.agent.state_dict(), os.path.join(ckpt_dir, f"checkpoint_{epoch}.dat"))
            if epoch % 10 == 0:   # full resumable state next to the reference-format file (every rank: env shards differ)
                os.makedirs(ckpt_dir, exist_ok=True)
                name = f"trainer_{epoch}.pt" if world == 1 else f"trainer_{epoch}.rank{rank}.pt"
                torch.save(trainer.state_dict(), os.path.join(ckpt_dir, name))
        if lazy and rank == 0:
            last = trainer.flush_scalars()
            if last is not None:
                report(args.n_epochs, last)
    finally:
        trainer.close()                                                                            # train.py:296
        if rank == 0:
            log.close()
            torch.save(trainer.agent.state_dict(), os.path.join(ckpt_dir, "model.dat"))            # train.py:301
        if world > 1:
            import torch.distributed as dist
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
