#!/usr/bin/env python3
"""bench.py -- headline benchmark: env steps/sec of the whole PPO loop (rollout + GAE + update, the
`charts/SPS` definition of the reference, train.py:174,292) on big_track.json, "16 rays" (17 actual).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload target|cfg1|cfg2|cfg4|cfg4i]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself: the parent
process -- before it has touched the GPU in any way -- runs `python -m torch.distributed.run` as a CHILD process (never
an exec) and exits with its code; rank 0 of the children prints the JSON line.

A "step" is ONE EPOCH of the hot path on one batch of synthetic input: n_steps vector-env steps of
n_envs envs per GPU (policy forward + sample + env-step kernel + store), the bootstrap value, the GAE
scan and train_iters x ceil(n_steps/batch_size) clipped-PPO minibatch updates (with the gradient
all-reduce when N > 1).  Nothing is skipped inside the timed region.  value = all ranks' env steps / the
slowest rank's time (weak scaling: n_envs per GPU is fixed).

Extra objects on the JSON line:
  roofline     -- the dominant kernel of the timed region (the persistent rollout kernel K9 / K9s when pc_rollout runs
                  the rollout, else the env-step kernel K1).  `achieved` / `peak` / `frac` keep the contract's figure:
                  algorithmic bytes per launch (176 B per env step at 16 rays) / the launch's mean duration by
                  HIP events on the launch stream inside the timed epochs, against the 8 TB/s HBM peak.  `bound` names
                  the roof that actually binds -- "valu": the env step is branchy fp32 geometry, 66 flop/B -- and `valu`,
                  `mfma`, `hbm` price the same launch against each roof; `gae` is the GAE scan's own HBM fraction.
  cpu_baseline -- the CPU oracle (oracle/, the checker -- never the product) driving the same rollout on ALL usable host cores of
                  this box for a bounded sample (`host_cores`, `usable_cores`, `threads` stated; `env_only_value` = CarEnv.step alone,
                  `value` = env + torch-CPU policy; `env_only_one_thread_value` beside the reference's Python figure per core).
  exact_f64_value -- the same workload with the env in float64 throughout (dtype "f64": bit-exact against the reference, state included),
                  3 epochs after the timed region: the throughput of the configuration that is exact BY CONSTRUCTION (`kernel` names the
                  persistent kernel that ran: "K9-literal" = float32 selection + the reference's literal arithmetic).
  other_workloads -- every other single-GPU BASELINE configuration (cfg1 = configs[1], cfg2 = configs[2], cfg4 = the per-GPU shard of
                  configs[4]; `target` when another workload is the headline), 5 epochs each after the timed region, with its own
                  ms_per_step, rollout-launch duration by HIP events and roofline fraction -- and `exact_f64`: the same
                  configuration in the bit-exact dtype (8 epochs).
  parity_check -- one more pc_rollout launch of the same trainer AFTER the timed region: one env of every 32-env wave x 64 steps
                  replayed through the CPU oracle (observations within one float32 ulp; rewards / flags exact; an env may leave the
                  oracle's trajectory only at a step whose threshold margin is below 1e-9 px: 0 departures above that margin).
                  `rare_branches`: the same check on a fresh trainer of the same workload started from INJECTED states (and, at 17 rays,
                  the trained policy fixture): laps, time limits, terminated-at-the-limit, +-990 turns of heading -- `events_replayed`.
  strict_fp32_value -- the same metric with the policy GEMMs as exact-fp32 MFMAs (3 epochs, outside the headline timing).
  strict_f64_fp32_value -- the strictest cell: dtype f64 (bit-exact env) AND the exact-fp32 policy chain, 3 epochs: every number of the rollout in
                  the reference's own arithmetic, as one persistent launch.
  fp32_grade_bf16x3_value -- the same with the bf16 x 3 split (six piece products: fp32-grade error, not the fp32 chain's bits).
"""
import argparse
import json
import os
import subprocess
import sys
import time

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
    # ... and its INTERLEAVED variant (`track_id = i & 1`, every wave holds both tracks -- what car_env.py:621-628 makes
    # legal per env): the persistent kernel's generic mode, the env step once per distinct track of a wave
    "cfg4i": dict(n_envs=32768, n_steps=1024, num_rays=16, batch_size=512, train_iters=40, mixed=True, interleave=True),
}
# SURVEY 8(d) / BASELINE.md section 4, per env step: state 64 B + action 8 B + observation 4 (6 + R) B + reward / flags 12 B
# = 156 / 176 / 240 B at 12 / 17 / 33 actual rays; 28 flop per ray-segment test x (R S + 4 gate tests) + R sincos pairs + ~30 physics
# = 8.2 / 11.6 / 22.4 kflop on big_track (S = 24 walls).  As FORMULAS of the actual ray count R and the track's wall count S, so
# that every ray count and every track prices its launch (a mixed batch: the mean over its envs' tracks).
def step_bytes(R):
    return 4 * (6 + R) + 84


def step_flops(R, walls):
    """walls: the wall count of every track of the batch, weighted equally (ppo.Trainer deals the envs to the tracks in equal blocks)"""
    return sum(28 * (R * S + 4) + 2 * R + 30 for S in walls) / len(walls)


HBM_PEAK_GBS = 8000.0                          # MI355X_MICROARCH.md: 8 TB/s
VALU_PEAK_TFLOPS = 157.3                       # fp32 vector peak
MFMA_F16_PEAK_TFLOPS = 2500.0                  # dense fp16 / bf16 matrix peak
MFMA_F32_PEAK_TFLOPS = 157.3                   # fp32-input matrix peak (v_mfma_f32_16x16x4_f32: the fp32 vector rate; MI355X_MICROARCH.md)
GAE_BYTES = 24                                 # per transition: 4 reads + 2 writes of float32
# the reference Python CarEnv timed in the survey container (it cannot travel to the GPU box)
PY_REFERENCE = {12: 320.0, 16: 227.0, 32: 154.0}   # env steps/s on one core, big_track


def parse_args():
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
    ap.add_argument("--no-extras", action="store_true", help="skip the parity check and the strict-fp32 bracket after the timed region")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend (nccl = RCCL)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal on a 1-GPU box: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--eager-rollout", action="store_true", help="no rollout graph; bracket env-step launches with events instead")
    ap.add_argument("--rollout-kernel", default="auto", choices=["auto", "mega", "steps"], help="persistent rollout kernel or 2 kernels/step")
    ap.add_argument("--rollout-form", type=int, default=-1, choices=[-1, 0, 1, 2, 3], help="PC_OPT_ROLLOUT_FORM of the trainer's env handle: -1 auto; 0/1 force the 32-env-wave / split form; 2/3 the same without the 1/den table in LDS (A/B knob)")
    ap.add_argument("--no-graphs", action="store_true", help="eager update and rollout")
    ap.add_argument("--torch-mlp", action="store_true", help="torch autograd GEMMs for the MLPs inside the minibatch step (fused loss/Adam kernels only)")
    ap.add_argument("--torch-update", action="store_true", help="reference torch ops for the whole minibatch step (no fused loss/Adam kernels)")
    ap.add_argument("--event-stride", type=int, default=8, help="with --eager-rollout: bracket every k-th env-step launch")
    ap.add_argument("--master-port", type=int, default=None, help="rendezvous port when bench.py starts the ranks itself")
    ap.add_argument("--force-collective", action="store_true", help="1 GPU: take the MULTI-RANK update path (K10, K11, RCCL all-reduce on a 1-rank "
                    "communicator, clip+Adam) -- the multi-GPU update's cost minus the xGMI transport")
    ap.add_argument("--exchange", default="rccl", choices=["rccl", "p2p"], help="multi-rank gradient exchange per minibatch: torch.distributed all_reduce (RCCL) "
                    "or the library's one-shot all-reduce over peer-mapped buffers (pc_xchg_*)")
    ap.add_argument("--capture-collectives", action="store_true", help="multi-rank, backend nccl or --exchange p2p: capture the per-minibatch all-reduce into the epoch's update graph "
                    "(off by default: the minibatch steps are enqueued eagerly around an eager all-reduce)")
    return ap.parse_args()


def visible_gpus():
    """How many devices this process can see, WITHOUT initialising HIP (torch.cuda.device_count() only enumerates on this image; the
    parent must stay GPU-free so that it may start the ranks as child processes)."""
    try:
        import torch
        return int(torch.cuda.device_count())
    except Exception:      # noqa: BLE001 -- no torch / no driver: the ranks will say so themselves
        return -1


def spawn_ranks(args):
    """--gpus N > 1 without a launcher: start N fresh rank processes (torch.distributed.run, one per GPU) as a child of this
    process, which has not initialised HIP (nothing here imports the extension or calls torch.cuda), and pass its exit code on."""
    have = visible_gpus()
    if not args.same_device and 0 <= have < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but only {have} device(s) are visible to this process (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES?); "
              "use --same-device --backend gloo for a one-device rehearsal", file=sys.stderr)
        return 2
    port = args.master_port or (29500 + os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    # (HSA_ENABLE_IPC_MODE_LEGACY=0 -- dmabuf IPC, which this pool's host driver needs for RCCL / cross-process device memory -- is
    # exported by the image itself and inherited here; nothing is set that the environment did not ask for)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    return subprocess.call(cmd, env=env)


def usable_cpus():
    """(host_cores, usable): os.cpu_count() and what this process may actually run on (its affinity mask, cut by a cgroup CPU quota
    when one is set)."""
    host = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = host
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            elif float(txt[0]) > 0:
                quota = float(txt[0]) / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except (OSError, ValueError, IndexError):
            continue
    if quota:
        usable = max(1, min(usable, int(quota + 0.999)))
    return host, usable, quota


def cpu_baseline(cfg, torch, budget_s=5.0):
    """The same path on the host, by the CPU oracle (oracle/: the checker -- `kind: "port"`, the C restatement of car_env.py; the
    Python reference cannot travel to the GPU box, its own figure is quoted from BASELINE.md).  Bounded legs on big_track, ALL usable
    host cores (no cap; the count is stated):
      env only, 1 thread            -- the per-core figure beside the reference's Python 227 steps/s/core;
      env only, all usable cores    -- `env_only_value`: CarEnv.step alone, uniform random actions; the envs dealt to POSIX threads in
                                       static ranges inside ONE C call per 64 steps (oc_vec_rollout_mt: envs never interact, so no
                                       per-step barrier; best 64-step block of the leg) -- `env_only_scaling` = that / (threads x the one-thread figure);
      env + torch-CPU policy        -- `value`: the rollout at the workload's ray count (policy forward + sample + env step);
      configs[0] as BASELINE.json writes it -- `configs0_whole_loop`: big_track, n_envs = 24, n_steps = 1024, 16 rays, the WHOLE loop of
                                       train.py:171-266 (rollout with the torch policy, GAE, 40 x 2 clipped-PPO minibatches on torch-CPU)."""
    import numpy as np

    import oracle
    from ppo_car_amd.model import Agent
    from ppo_car_amd.ppo import PPOConfig, PPOLearner
    host, usable, quota = usable_cpus()
    threads = usable
    track = oracle.Track(os.path.join(ROOT, "tracks", "big_track.json"))

    def env_only(n_threads, n_envs, budget):
        env = oracle.OracleVecEnv(track, n_envs, num_rays=cfg["num_rays"], reward_scaling=0.1, threads=1)
        env.reset()
        rng = np.random.default_rng(0)
        acts = rng.integers(0, 9, size=(64, n_envs)).astype(np.int64)
        env.run_steps(acts[:2], threads=n_threads)
        steps, t0, best = 0, time.time(), 0.0
        while True:
            ta = time.time()
            env.run_steps(acts, threads=n_threads)
            best = max(best, n_envs * 64 / (time.time() - ta))      # the CPU's best 64-step block (a baseline is owed the benefit of the doubt:
            steps += 64                                              # the job's cores are a cgroup quota on a shared host)
            if time.time() - t0 > budget:
                break
        return best, steps

    n_envs = max(8192, 512 * threads)
    one, steps1 = env_only(1, 1024, budget_s * 0.4)
    allc, steps_all = env_only(threads, n_envs, budget_s * 0.6)
    torch.set_num_threads(threads)
    env = oracle.OracleVecEnv(track, n_envs, num_rays=cfg["num_rays"], reward_scaling=0.1, threads=threads)
    obs = torch.from_numpy(env.reset())
    agent = Agent(env.D, 9)
    steps, t0 = 0, time.time()
    with torch.no_grad():
        while True:
            a, lp, _, v = agent.get_action_and_value(obs)
            o, r, te, tr = env.step(a.numpy())
            obs = torch.from_numpy(o)
            steps += 1
            if time.time() - t0 > budget_s * 0.6 and steps >= 4:
                break
    dt = time.time() - t0

    # ---- configs[0]: "CPU reference: train.py big_track.json n_envs=24 n_steps=1024 16 rays" -- the whole loop on the host
    def configs0(budget):
        N0, T0, n0 = 24, 1024, 16
        th = max(1, min(threads, 8))
        torch.set_num_threads(th)
        c0 = PPOConfig(n_envs=N0, n_steps=T0, batch_size=512, train_iters=40, num_rays=n0, seed=0, use_graphs=False, fused_update=False)
        e0 = oracle.OracleVecEnv(track, N0, num_rays=n0, reward_scaling=c0.reward_scaling, threads=1)
        ag = Agent(e0.D, 9)
        learner = PPOLearner(ag, c0, "cpu")
        nobs = torch.from_numpy(e0.reset())
        nterm = ntrunc = torch.zeros(N0)
        obs_b, act_b, rew_b = torch.zeros(T0, N0, e0.D), torch.zeros(T0, N0), torch.zeros(T0, N0)
        val_b, lp_b, te_b, tr_b = torch.zeros(T0, N0), torch.zeros(T0, N0), torch.zeros(T0, N0), torch.zeros(T0, N0)
        epochs, t_roll, t_upd, t0_ = 0, 0.0, 0.0, time.time()
        while True:
            ta = time.time()
            with torch.no_grad():
                for t in range(T0):                                   # train.py:173-195
                    obs_b[t], te_b[t], tr_b[t] = nobs, nterm, ntrunc
                    a_, lp_, _, v_ = ag.get_action_and_value(nobs)
                    o_, r_, te_, tr_ = e0.step(a_.numpy())
                    act_b[t], lp_b[t], val_b[t] = a_, lp_, v_.view(-1)
                    rew_b[t] = torch.from_numpy(r_.astype(np.float32))
                    nobs, nterm, ntrunc = torch.from_numpy(o_), torch.from_numpy(te_.astype(np.float32)), torch.from_numpy(tr_.astype(np.float32))
                last_v = ag.get_value(nobs).view(-1)                  # :200
            tb = time.time()
            adv, ret = oracle.gae(rew_b.numpy(), val_b.numpy(), te_b.numpy(), tr_b.numpy(), last_v.numpy(), nterm.numpy(), ntrunc.numpy(),
                                  c0.gamma, c0.gae_lambda)            # buffer.py:36-64
            learner.update(obs_b.view(-1, e0.D), act_b.view(-1), lp_b.view(-1), torch.from_numpy(adv).view(-1), torch.from_numpy(ret).view(-1))
            tc = time.time()
            t_roll += tb - ta
            t_upd += tc - tb
            epochs += 1
            if tc - t0_ > budget:
                break
        tot = time.time() - t0_
        return {"value": N0 * T0 * epochs / tot, "unit": "env steps/s", "epochs": epochs, "seconds": tot, "rollout_s_per_epoch": t_roll / epochs,
                "gae_update_s_per_epoch": t_upd / epochs, "torch_threads": th, "env_threads": 1,
                "workload": "BASELINE.json configs[0]: big_track.json, n_envs=24, n_steps=1024, 16 -> 17 rays, batch 512, 40 iters: rollout (torch-CPU policy + C "
                            "oracle env) + GAE (C restatement of buffer.py:36-64) + 80 clipped-PPO minibatches (torch-CPU autograd, train.py:223-266)",
                "reference_published": {"value": 2300.0, "unit": "env steps/s", "source": "BASELINE.md: the reference README's whole-loop figure (Python env, 16 processes)"}}

    try:
        c0 = configs0(budget_s)
    except Exception as ex:      # noqa: BLE001
        c0 = {"error": repr(ex)}
    py = PY_REFERENCE.get(cfg["num_rays"])
    return {"value": n_envs * steps / dt, "unit": "env steps/s", "cores": threads, "kind": "port",
            "host_cores": host, "usable_cores": usable, "cgroup_cpu_quota": quota, "threads": threads,
            "env_only_value": allc, "env_only_one_thread_value": one, "env_only_scaling": allc / one / threads,
            "configs0_whole_loop": c0,
            "sample": f"big_track, {cfg['num_rays']} rays: (a) env only, 1 thread, 1024 envs x {steps1} steps; (b) env only, {threads} POSIX threads over static env "
                      f"ranges, {n_envs} envs x {steps_all} steps -> env_only_value; (c) rollout = torch-CPU policy ({threads} threads) + C oracle env "
                      f"({threads} threads), {n_envs} envs x {steps} steps -> value; (d) configs0_whole_loop: 24 envs x 1024 steps x its epochs; every leg bounded "
                      f"to ~{budget_s * 0.5:.0f}-{budget_s:.0f} s",
            "reference_python": {"value": py, "unit": "env steps/s", "cores": 1,
                                 "source": "BASELINE.md section 2: the reference's Python CarEnv on one Xeon 2.1 GHz core of the survey "
                                           "container (env only, big_track); it cannot travel to the GPU box, so it is quoted, not re-timed"}}


def parity_check(tr, cfg, torch, np, envs=1024, steps=64):
    """One more pc_rollout launch of this trainer (same shape, same kernel, after the timed region): `steps` steps of a STRIDED
    sample of its envs -- one or more out of every 32-env wave of the launch, so that every workgroup is represented -- replayed
    through the CPU oracle from the env state the launch started from.  An env may leave the oracle's trajectory only at a
    step whose threshold margin |d - 10 px| (collision rays against the walls, or against gate[next] at the pre-step pose) is
    below 1e-9 px -- checked for every departure; observations must agree within one float32 ulp before that.  A mixed-track
    batch (track list, in blocks: ppo.Trainer's layout) is checked on a strided sample per track."""
    import oracle
    OBS_TOL, MARGIN_PX = 1.2e-7, 1e-9
    st = tr.envs.get_state()
    tracks = list(cfg.track) if isinstance(cfg.track, (list, tuple)) else [cfg.track]
    nt = len(tracks)
    T = min(steps, cfg.n_steps - 1)            # rows 1..T of the buffer hold the observations after steps 0..T-1
    first_all = tr.next_obs.clone()
    tr.rollout()
    torch.cuda.synchronize()
    b = tr.buffer
    i = np.arange(cfg.n_envs)
    tid = i % nt if cfg.track_interleave else np.minimum((i // 32 * 32) * nt // cfg.n_envs, nt - 1)      # ppo.Trainer's two layouts
    n, step_deg = cfg.num_rays, 360 // cfg.num_rays
    col = list(range(0, n, n // 4))            # Car.check_collision's rays (car_env.py:389)
    worst, flips, checked, n_cmp, n_eq, worst_margin, waves = 0.0, 0, 0, 0, 0, 0.0, set()
    events = dict(gates=0, laps=0, truncations=0, terminated_at_time_limit=0, episodes_ended=0)    # what the replay went through (car_env.py:726-750)
    for k, path in enumerate(tracks):
        mine = np.nonzero(tid == k)[0]
        wave0 = np.unique(mine // 32) * 32                                   # first env of every 32-env wave on this track
        per_wave = min(32, max(1, (envs // nt) // len(wave0)))               # at least one env of EVERY wave, at most all 32
        stride = max(1, 32 // per_wave)
        offs = (np.arange(per_wave)[None, :] * stride + (wave0[:, None] // 32) % stride)
        sel = (wave0[:, None] + offs).reshape(-1)
        sel = np.unique(sel[(sel < cfg.n_envs) & (tid[np.minimum(sel, cfg.n_envs - 1)] == k)])
        P = len(sel)
        waves |= set((sel // 32).tolist())
        idx = torch.as_tensor(sel, device=b.obs_buf.device)
        acts = b.act_buf[:T][:, idx].cpu().numpy().astype(np.int64)
        trk = oracle.Track(path)
        ora = oracle.OracleVecEnv(trk, P, num_rays=n, reward_scaling=cfg.reward_scaling, threads=4)
        ora.reset()
        ora.set_state(**{f: st[f][sel] for f in ("px", "py", "vx", "vy", "rot", "time_step", "next_gate", "passed")})
        OB, RW = b.obs_buf[:T + 1][:, idx].cpu().numpy(), b.rew_buf[:T][:, idx].cpu().numpy()
        TE, TR = b.term_buf[:T + 1][:, idx].cpu().numpy() != 0, b.trunc_buf[:T + 1][:, idx].cpu().numpy() != 0
        alive = np.ones(P, bool)
        worst = max(worst, float(np.abs(OB[0] - first_all[idx].cpu().numpy()).max()))
        for t in range(T):
            pre = {f: getattr(ora, f).copy() for f in ("px", "py", "rot", "next_gate", "time_step")}
            o, r, te, trn, fin = ora.step(acts[t], want_final_obs=True)
            x = r.astype(np.float64) / cfg.reward_scaling        # raw reward: 0.01 forward + 1 gate + 10 lap - 3 crash
            lap = x > 5.0
            events["laps"] += int((lap & alive).sum())
            events["gates"] += int((((x - 10.0 * lap + 3.0 * te) > 0.5) & alive).sum())
            events["truncations"] += int((trn & alive).sum())
            events["terminated_at_time_limit"] += int((te & (pre["time_step"] >= 999) & alive).sum())   # `elif`: terminated wins (car_env.py:746-750)
            events["episodes_ended"] += int(((te | trn) & alive).sum())
            bad = (TE[t + 1] != te) | (TR[t + 1] != trn) | (RW[t] != r.astype(np.float32))
            for e in np.nonzero(bad & alive)[0]:     # a departure: how close to a threshold was the reference itself?
                walls = [abs(float(fin[e, 6 + c]) * 1000.0 - 10.0) for c in col]
                gate = trk.gates[int(pre["next_gate"][e])]
                gates = [abs(oracle.ray_distance(pre["px"][e], pre["py"][e], pre["rot"][e] + c * step_deg, gate) - 10.0) for c in col]
                worst_margin = max(worst_margin, min(walls + gates))
                flips += 1
            alive &= ~bad
            if alive.any():
                err = np.abs(OB[t + 1][alive].astype(np.float64) - o[alive])
                worst = max(worst, float(err.max()))
                n_cmp += err.size
                n_eq += int((err == 0).sum())
        checked += P
    return {"kernel": tr.rollout_mode, "envs": checked, "waves_sampled": len(waves), "waves_total": (cfg.n_envs + 31) // 32, "tracks": nt, "steps": T,
            "obs_max_abs_err": worst, "obs_tolerance": OBS_TOL, "obs_entries_bit_equal": n_eq / max(1, n_cmp),
            "envs_left_oracle_trajectory": flips, "largest_threshold_margin_px_of_a_departure": worst_margin, "margin_tolerance_px": MARGIN_PX,
            "rewards_and_flags": "exact on every env still on the oracle's trajectory", "events_replayed": events,
            "ok": bool(worst <= OBS_TOL and worst_margin <= MARGIN_PX),     # i.e. 0 departures above the margin, however many envs
            "checker": "oracle/carenv_oracle.c (float64 restatement of car_env.py:693-760), teacher-forced by the stored actions"}


def parity_check_rare(make_trainer, wl, track, torch, np, steps=64):
    """The second parity leg: the SAME kernel (same shape, tracks, dtype and arithmetic: a fresh trainer of the benchmarked workload with a
    short buffer) driven into the bookkeeping branches a young policy never reaches -- the lap wrap (car_env.py:730-737), the time limit
    and its `elif` (:746-750), heading drift of up to 990 turns -- by injected states (oracle/scenarios.py: cars on the approach to the last
    gate with next_gate = G - 1, time steps 997 .. 999, cars about to hit a wall as the limit falls due) and, at 16 -> 17 rays, by the
    trained policy fixture (tests/golden/policy_trained.npz); then parity_check's replay from those states.  `ok` also requires that the
    replay actually went through laps, truncations and a terminated-at-the-time-limit step."""
    from oracle.scenarios import injected_state, load_trained_policy
    wl_ = dict(wl, n_steps=steps + 1)
    cfg_, t_ = make_trainer(wl_=wl_, track_=track)
    try:
        trained = t_.obs_dim[0] == 23
        if trained:
            load_trained_policy(t_.agent)
        tracks = list(track) if isinstance(track, (list, tuple)) else [track]
        i = np.arange(cfg_.n_envs)
        tid = i % len(tracks) if cfg_.track_interleave else np.minimum((i // 32 * 32) * len(tracks) // cfg_.n_envs, len(tracks) - 1)
        full = None
        for k, path in enumerate(tracks):
            mine = np.nonzero(tid == k)[0]
            part = injected_state(path, mine)
            if full is None:
                full = {f: np.zeros(cfg_.n_envs, v.dtype) for f, v in part.items()}
            for f, v in part.items():
                full[f][mine] = v
        t_.envs.set_state(**full)
        res = parity_check(t_, cfg_, torch, np, envs=2560, steps=steps)
        ev = res["events_replayed"]
        res["policy"] = "tests/golden/policy_trained.npz" if trained else "freshly initialised (the fixture is a 17-ray policy)"
        res["states"] = "oracle/scenarios.py: injected_state (approach to the last gate, time steps 995 .. 999, +-80 .. 990 turns of heading, cars about to crash at the time limit)"
        res["ok"] = bool(res["ok"] and ev["laps"] > 0 and ev["truncations"] > 0 and ev["terminated_at_time_limit"] > 0)
        return res
    finally:
        t_.close()


def main():
    args = parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        sys.exit(spawn_ranks(args))

    # Only the JSON line may reach stdout: RCCL prints a version banner there when a communicator is created.  Everything else
    # this process (and the libraries it loads) writes to fd 1 goes to stderr; the line itself is written to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(world_env or "1")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    knobs = {k: v for k, v in os.environ.items() if k.startswith("PPOCAR_")}
    if knobs:
        raise SystemExit(f"bench.py: PPOCAR_* environment variables are set ({knobs}); the product reads none -- unset them")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU path for the hot path")
    if args.same_device:
        local_rank = 0
    elif torch.cuda.device_count() < world:      # (a launcher started the ranks: the same one-line reason, non-zero exit on every rank)
        raise SystemExit(f"bench.py: --gpus {world} but only {torch.cuda.device_count()} device(s) are visible to rank {rank}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or args.force_collective:
        import torch.distributed as dist
        if world == 1:       # --force-collective: a one-rank communicator
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(args.master_port or (29500 + os.getpid() % 2000)))
            dist.init_process_group(args.backend, rank=0, world_size=1, **({"device_id": dev} if args.backend == "nccl" else {}))
        elif args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    from ppo_car_amd.ppo import PPOConfig, Trainer
    from ppo_car_amd._capi import lib as _lib
    if _lib.pc_build_ablate() != 0:
        raise SystemExit("bench.py: libppocar.so is a timing-ablation build (PC_ABLATE != 0)")
    PREC = {"fp16x2": 2, "bf16x3": 1, "fp32": 0}
    POLICY_ARITH = {"fp16x2": "fp16x2 split (v = h + l), 3 products, fp32 accumulate on the fp16 matrix cores (fp32-class; DESIGN.md section 5)",
                    "bf16x3": "bf16x3 split, 6 products, fp32 accumulate on the bf16 matrix cores (fp32-equivalent; DESIGN.md section 5)",
                    "fp32": "fp32-input MFMA (exact fp32 fmaf chain)"}[args.policy_arith]
    wl = dict(WORKLOADS[args.workload])
    if args.n_envs:
        wl["n_envs"] = args.n_envs
    if args.n_steps:
        wl["n_steps"] = args.n_steps
    mixed = wl.pop("mixed", False)
    interleave = wl.pop("interleave", False)
    if interleave:
        wl["track_interleave"] = True
    track = ([os.path.join(ROOT, "tracks", "track.json"), os.path.join(ROOT, "tracks", "big_track.json")] if mixed
             else os.path.join(ROOT, "tracks", "big_track.json"))

    def tracks_of(mixed_):
        return ([os.path.join(ROOT, "tracks", "track.json"), os.path.join(ROOT, "tracks", "big_track.json")] if mixed_
                else os.path.join(ROOT, "tracks", "big_track.json"))

    def make_trainer(policy_precision=PREC[args.policy_arith], wl_=None, env_dtype=None, track_=None):
        # every launch option lives in this trainer's handles (pc_policy, pc_env): nothing process-wide is touched
        cfg_ = PPOConfig(track=track if track_ is None else track_, env_dtype=env_dtype or args.env_dtype, seed=0, policy=args.policy,
                         use_graphs=not args.no_graphs,
                         fused_update=not args.torch_update, custom_mlp=not args.torch_mlp, rollout_kernel=args.rollout_kernel,
                         force_collective=args.force_collective, capture_collectives=bool(args.capture_collectives),
                         policy_precision=policy_precision, exchange=args.exchange, rollout_form=args.rollout_form, **(wl if wl_ is None else wl_))
        t_ = Trainer(cfg_, device=dev, rank=rank, world_size=world)
        return cfg_, t_

    def wall_counts(track_):
        from ppo_car_amd.env import Track
        return [Track(t).n_walls for t in (track_ if isinstance(track_, (list, tuple)) else [track_])]

    def side_measurement(name, wl_in, env_dtype="f32", epochs=5, policy_precision=None):
        """Another workload (or the same one in another env dtype) AFTER the headline's timed region, same process, same box:
        2 warm-up epochs (graphs: one eager, then capture), then `epochs` timed epochs bracketed like the headline; the rollout
        launch by HIP events on the launch stream inside those epochs; the launch priced against the same roofs."""
        wl_ = dict(wl_in)
        mixed_ = wl_.pop("mixed", False)
        inter_ = wl_.pop("interleave", False) or wl_.get("track_interleave", False)
        if inter_:
            wl_["track_interleave"] = True
        trk = tracks_of(mixed_)
        c_, t_ = (make_trainer(wl_=wl_, env_dtype=env_dtype, track_=trk) if policy_precision is None
                  else make_trainer(policy_precision=policy_precision, wl_=wl_, env_dtype=env_dtype, track_=trk))
        try:
            for _ in range(2):
                t_.run_epoch(sync=False)
            t_.mega_events, t_.phase_events = [], []
            d_ = timed(t_, epochs)
            torch.cuda.synchronize()
            R_ = t_.obs_dim[0] - 6
            fl, by = step_flops(R_, wall_counts(trk)), step_bytes(R_)
            units_ = c_.n_envs * c_.n_steps
            r_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in t_.phase_events]))
            mega = float(np.mean([a.elapsed_time(b) for a, b in t_.mega_events]) * 1e3) if t_.mega_events else None
            sec_ = (mega if mega is not None else r_ms * 1e3) * 1e-6     # per-step kernels (f64): the rollout phase of the epoch
            res = {"workload": f"{name}: {('track.json + big_track.json (' + ('interleaved env by env' if inter_ else 'halves') + ')') if mixed_ else 'big_track.json'}, num_rays={c_.num_rays} ({R_} actual), "
                               f"n_envs={c_.n_envs}, n_steps={c_.n_steps}, batch_size={c_.batch_size}, train_iters={c_.train_iters}",
                   "value": units_ * epochs / d_, "unit": "env steps/s", "epochs": epochs, "ms_per_step": d_ / epochs * 1e3,
                   "dtype": env_dtype, "rollout": t_.rollout_mode,
                   "kernel": t_.envs.last_rollout_kernel() if t_.rollout_mode == "mega" else "per-step kernels",
                   "epoch_split": {"rollout_ms": r_ms, "gae_update_ms": float(np.mean([e[1].elapsed_time(e[2]) for e in t_.phase_events]))},
                   "roofline": {"bound": "valu", "launch_us": sec_ * 1e6,
                                "launch_us_method": ("HIP events on the launch stream around each pc_rollout launch inside the timed epochs" if mega is not None
                                                     else "HIP events around the rollout phase (per-step kernels replayed as one HIP graph) inside the timed epochs"),
                                "achieved": fl * units_ / sec_ / 1e12, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": fl * units_ / sec_ / 1e12 / VALU_PEAK_TFLOPS, "flops_per_env_step": fl,
                                "hbm_frac": by * units_ / sec_ / 1e9 / HBM_PEAK_GBS, "bytes_per_env_step": by}}
            return res
        finally:
            t_.mega_events = t_.phase_events = None
            t_.close()

    cfg, tr = make_trainer()
    tr.profile_stride = 0
    # Env-step probe (SURVEY 8(d)'s level (i): the env alone), after the timed epochs and outside them: a second env batch of the same
    # size is given the trainer's env STATE as the timed epochs left it (cars spread over the track by the current policy -- a batch
    # fresh from reset has every car at the start line, where no ray takes the rare paths, and flatters a single step by 2x) and the
    # ACTIONS the last rollout stored; timed between HIP events on the launch stream: PROBE launches of pc_env_step as a user of the
    # drop-in boundary gets them (the table-driven kernel K1f from 8192 envs on), the same with the generic kernel K1 forced, and
    # pc_env_step_many over PROBE_T action rows (one launch).
    from ppo_car_amd.env import VecCarEnv
    PROBE, PROBE_T = 32, 256
    probe = VecCarEnv(cfg.n_envs, os.path.join(ROOT, "tracks", "big_track.json"), num_rays=cfg.num_rays, reward_scaling=cfg.reward_scaling, device=dev, dtype=cfg.env_dtype)
    p_obs, _ = probe.reset()
    p_out = (p_obs, torch.empty(cfg.n_envs, device=dev), torch.empty(cfg.n_envs, device=dev), torch.empty(cfg.n_envs, device=dev))

    def run_probe():
        """-> {"step_us", "step_kernel", "generic_us", "many_us_per_step", "many_kernel", "state"}"""
        res = {"state": "reset (mixed-track workload: the probe batch is big_track only)"}
        Tn = min(PROBE_T, cfg.n_steps)
        acts = tr.buffer.act_buf[:Tn].to(torch.int64).contiguous()          # what the policy drew in the last rollout
        state = None
        if not mixed:
            torch.cuda.synchronize()
            state = tr.envs.get_state()
            res["state"] = "the trainer's env state after the timed epochs"

        def fresh():
            probe.reset()
            if state is not None:
                probe.set_state(**state)

        def span(fn, reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(reps):
                fn(i)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / reps
        for key, form in (("generic_us", 1), ("step_us", 0)):
            probe.set_option("step_form", form)
            best = None
            for _ in range(3):
                fresh()
                us = span(lambda i: probe.step(acts[i % Tn], out=p_out), PROBE)
                best = us if best is None else min(best, us)
            res[key] = best
            if form == 0:
                res["step_kernel"] = probe.last_step_kernel()
        outm = (torch.empty(Tn, cfg.n_envs, obs_dim_probe, device=dev), torch.empty(Tn, cfg.n_envs, device=dev),
                torch.empty(Tn, cfg.n_envs, device=dev), torch.empty(Tn, cfg.n_envs, device=dev))
        best = None
        for _ in range(3):
            fresh()
            us = span(lambda i: probe.step_many(acts, out=outm), 1) / Tn
            best = us if best is None else min(best, us)
        res["many_us_per_step"], res["many_kernel"], res["many_T"] = best, probe.last_step_kernel(), Tn
        return res
    obs_dim_probe = probe.obs_dim

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(trn, n_epochs):
        barrier()
        t0 = time.perf_counter()
        for _ in range(n_epochs):
            trn.run_epoch(sync=False)
        barrier()
        dt_ = time.perf_counter() - t0
        trn.check_exchange()        # --exchange p2p: a timed-out exchange invalidates the run (raises, non-zero exit)
        return dt_

    for _ in range(max(args.warmup, 2 if cfg.use_graphs else 0)):   # graphs: 1 eager epoch, then capture, then replay
        tr.run_epoch(sync=False)
    if args.eager_rollout:
        tr.profile_stride = args.event_stride
    tr.k1_events = []
    tr.phase_events = []
    tr.mega_events = []
    if tr.learner.collective and not tr.learner._can_capture_update():
        tr.learner.exchange_events = []      # HIP events around every eagerly enqueued exchange of the timed epochs (a captured one cannot be bracketed)
    dt = timed(tr, args.steps)
    try:                                 # the stand-alone env-step probe, outside the timed region (it is not part of the path)
        probe_res = run_probe()
    except Exception as ex:              # (the probe must never take the benchmark line down with it)
        probe_res = {"error": repr(ex)}
    # the GAE scan alone (HBM-bound: 24 B per transition), outside the timed region
    gae_events = []
    tr.buffer.ptr = tr.buffer.capacity          # (Buffer.get() rewinds it; the rows of the last epoch are still there)
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        tr.buffer.calculate_advantages(tr.next_term.reshape(1, -1), tr.next_term.reshape(1, -1), tr.next_trunc.reshape(1, -1))
        e1.record()
        gae_events.append((e0, e1))
    torch.cuda.synchronize()
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    # the exchange step of the timed epochs, by HIP events on the stream: from the end of K11 (this rank's gradient is complete) to the end of
    # the all-reduce -- the wait for the slowest peer included.  THE multi-GPU number: the message is 59 KB, the step is latency-bound.
    exchange_us, rollout_ranks_ms, rccl_live = None, None, 0
    if tr.learner.exchange_events:
        ex = np.sort(np.array([a.elapsed_time(b) * 1e3 for a, b in tr.learner.exchange_events]))
        exchange_us = {"mean": float(ex.mean()), "p50": float(ex[len(ex) // 2]), "p90": float(ex[int(len(ex) * 0.9)]), "max": float(ex[-1]), "n": int(len(ex)),
                       "method": "HIP events on the launch stream around every exchange of the timed epochs, rank 0 (includes the wait for the slowest rank's gradient)"}
    elif tr.learner.collective:
        exchange_us = {"mean": None, "note": "the exchanges are captured inside the epoch's update graph (--capture-collectives): not bracketed; run without it"}
    tr.learner.exchange_events = None
    if dist is not None and world > 1:
        mine_ms = torch.tensor([float(np.mean([e[0].elapsed_time(e[1]) for e in tr.phase_events])),
                                float(np.mean([e[1].elapsed_time(e[2]) for e in tr.phase_events]))], dtype=torch.float64,
                               device=dev if dist.get_backend() == "nccl" else "cpu")
        allr = [torch.empty_like(mine_ms) for _ in range(world)]
        dist.all_gather(allr, mine_ms)
        ro, up = [float(t[0]) for t in allr], [float(t[1]) for t in allr]
        rollout_ranks_ms = {"min": min(ro), "max": max(ro), "gae_update_min": min(up), "gae_update_max": max(up), "per_rank": ro}
    if dist is not None and dist.get_backend() == "nccl":
        one = torch.ones(1, device=dev)
        dist.all_reduce(one)                 # the LIVE communicator's size: how many ranks actually took part in an RCCL collective
        rccl_live = int(one.item())
    replicas_equal = None
    if dist is not None and world > 1:
        # the replicas after the timed epochs: every rank's flat parameter buffer against rank 0's, bit for bit (clip + Adam are
        # replicated on the all-reduced gradient: train.py:259-261 on every rank)
        mine = tr.learner.flat_param.detach().clone()
        if dist.get_backend() != "nccl":
            mine = mine.cpu()
        every = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        replicas_equal = bool(all(torch.equal(every[0], t) for t in every[1:]))
    k1_us = probe_res.get("step_us")
    k1_bracketed_us = float(np.mean([a.elapsed_time(b) for a, b in tr.k1_events]) * 1e3) if tr.k1_events else None
    gae_us = float(np.median([a.elapsed_time(b) for a, b in gae_events]) * 1e3)
    probe.close()
    info = tr.envs.launch_info()
    split = {"rollout_ms": float(np.mean([e[0].elapsed_time(e[1]) for e in tr.phase_events])),
             "gae_update_ms": float(np.mean([e[1].elapsed_time(e[2]) for e in tr.phase_events]))}
    mega_us = float(np.mean([a.elapsed_time(b) for a, b in tr.mega_events]) * 1e3) if tr.mega_events else None
    rollout_mode, obs_dim, custom = tr.rollout_mode, tr.obs_dim[0], bool(tr.learner.custom)
    rollout_kernel_name = tr.envs.last_rollout_kernel() if rollout_mode == "mega" else "per-step kernels"
    captured = tr.learner._epoch_graph is not None
    tr.mega_events = None
    tr.phase_events = None

    extras = {}
    if rank == 0 and world == 1 and not args.no_extras and args.env_dtype == "f32":
        try:
            extras["parity_check"] = parity_check(tr, cfg, torch, np)
        except Exception as ex:                      # the check must never take the benchmark line down with it
            extras["parity_check"] = {"ok": False, "error": repr(ex)}
        try:
            extras["parity_check"]["rare_branches"] = parity_check_rare(make_trainer, wl, track, torch, np)
        except Exception as ex:
            extras["parity_check"]["rare_branches"] = {"ok": False, "error": repr(ex)}
    tr.close()
    del tr
    if world == 1 and not args.no_extras and args.policy_arith != "fp32" and args.policy == "fused" and not args.force_collective:
        # the price of exact-fp32 policy GEMMs (v_mfma_f32_16x16x4_f32) on the same workload: 3 epochs, outside the headline timing
        try:
            cfg2, tr2 = make_trainer(policy_precision=0)
            for _ in range(2):
                tr2.run_epoch(sync=False)
            dt2 = timed(tr2, 3)
            extras["strict_fp32_value"] = {"value": cfg2.n_envs * cfg2.n_steps * 3 / dt2, "unit": "env steps/s", "epochs": 3,
                                           "ms_per_step": dt2 / 3 * 1e3, "rollout": tr2.rollout_mode,
                                           "policy_gemm_arithmetic": "fp32-input MFMA (exact fp32 fmaf chain), --policy-arith fp32"}
            tr2.close()
            del tr2
        except Exception as ex:
            extras["strict_fp32_value"] = {"error": repr(ex)}
        if args.policy_arith != "bf16x3":
            # ... and the fp32-GRADE form in between: bf16 x 3 operand split, six piece products on the bf16 matrix cores -- 1.0e-7 against
            # float64 on the trained-policy fixture where the exact fp32 chain has 0.8e-7 (tests/test_gae_sample_gpu.py), no scaled domains
            try:
                cfg3, tr3 = make_trainer(policy_precision=1)
                for _ in range(2):
                    tr3.run_epoch(sync=False)
                dt3 = timed(tr3, 3)
                extras["fp32_grade_bf16x3_value"] = {"value": cfg3.n_envs * cfg3.n_steps * 3 / dt3, "unit": "env steps/s", "epochs": 3,
                                                     "ms_per_step": dt3 / 3 * 1e3, "rollout": tr3.rollout_mode,
                                                     "policy_gemm_arithmetic": "bf16 x 3 split, 6 piece products (fp32-grade: not the fp32 chain's bits), --policy-arith bf16x3"}
                tr3.close()
                del tr3
            except Exception as ex:
                extras["fp32_grade_bf16x3_value"] = {"error": repr(ex)}

    if world == 1 and not args.no_extras and args.env_dtype == "f32" and args.policy == "fused" and not args.force_collective:
        # the bit-exact configuration (float64 throughout, the reference's own operation order: observations, rewards, events AND
        # the float64 state equal the reference's bit for bit) on the SAME workload ...
        try:
            extras["exact_f64_value"] = side_measurement(args.workload, WORKLOADS[args.workload] if not (args.n_envs or args.n_steps) else dict(wl, mixed=mixed, interleave=interleave),
                                                         env_dtype="f64", epochs=3)
        except Exception as ex:
            extras["exact_f64_value"] = {"error": repr(ex)}
        # ... and THE STRICTEST CELL: float64 env AND the policy GEMMs as the exact fp32 chain -- every number of the rollout in the reference's
        # own arithmetic (car_env.py in float64, model.py's fp32 Linear chain), still one persistent launch (K9's literal form with the fp32 image)
        try:
            sf = side_measurement(args.workload, WORKLOADS[args.workload] if not (args.n_envs or args.n_steps) else dict(wl, mixed=mixed, interleave=interleave),
                                  env_dtype="f64", epochs=3, policy_precision=0)
            sf["policy_gemm_arithmetic"] = "fp32-input MFMA (exact fp32 fmaf chain), --policy-arith fp32"
            extras["strict_f64_fp32_value"] = sf
        except Exception as ex:
            extras["strict_f64_fp32_value"] = {"error": repr(ex)}
        # ... and every other single-GPU BASELINE configuration, each with its own ms_per_step and roofline fraction
        others = {}
        for name in ("target", "cfg1", "cfg2", "cfg4", "cfg4i"):
            if name == args.workload:
                continue
            try:
                others[name] = side_measurement(name, WORKLOADS[name], env_dtype="f32", epochs=5)
            except Exception as ex:
                others[name] = {"error": repr(ex)}
            try:      # ... and the same configuration in the bit-exact dtype (8 epochs): value, ms per epoch, the kernel that ran
                f64 = side_measurement(name, WORKLOADS[name], env_dtype="f64", epochs=8)      # (8: three epochs of configs[1] are 20 ms, one hiccup halves the figure)
                others[name]["exact_f64"] = {k: f64[k] for k in ("value", "unit", "ms_per_step", "kernel", "epoch_split")}
            except Exception as ex:
                others[name]["exact_f64"] = {"error": repr(ex)}
        if "cfg4" in others and "cfg4i" in others and "value" in others["cfg4"] and "value" in others["cfg4i"]:
            others["cfg4i"]["ratio_to_halves_layout"] = others["cfg4i"]["value"] / others["cfg4"]["value"]
        extras["other_workloads"] = others

    if rank == 0:
        env_steps = cfg.n_envs * cfg.n_steps * args.steps * world
        nr = cfg.num_rays
        per_step_bytes = step_bytes(obs_dim - 6)                           # algorithmic bytes per env step
        per_step_flops = step_flops(obs_dim - 6, wall_counts(track))       # ... and flops, for this batch's track(s)
        step_names = {"K1": "env_step_kernel (K1: the generic per-step kernel)", "K1f": "env_steps_fast_kernel (K1f: the table-driven env step, one launch per step)",
                      "K1f-table": "env_steps_fast_kernel (K1f, 1/den table staged: T steps in one launch)"}
        if k1_us:
            k1 = {"kernel": step_names.get(probe_res.get("step_kernel"), "?") + ", stand-alone (pc_env_step as the drop-in boundary launches it)", "launch_us": k1_us,
                  "achieved": per_step_bytes * cfg.n_envs / (k1_us * 1e-6) / 1e9, "unit": "GB/s", "frac": per_step_bytes * cfg.n_envs / (k1_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                  "valu_frac": per_step_flops * cfg.n_envs / (k1_us * 1e-6) / 1e12 / VALU_PEAK_TFLOPS, "env_steps_per_s": cfg.n_envs / (k1_us * 1e-6),
                  "generic_kernel_launch_us": probe_res.get("generic_us"), "state": probe_res.get("state"),
                  "launch_us_method": f"{PROBE} back-to-back launches (the last rollout's action rows) between two HIP events on the launch stream, best of 3, after the timed epochs",
                  "launch_us_bracketed_in_rollout": k1_bracketed_us}
        else:
            k1 = {"error": probe_res.get("error", "no probe")}
        env_only = None
        if probe_res.get("many_us_per_step"):
            mu = probe_res["many_us_per_step"]
            env_only = {"kernel": step_names.get(probe_res.get("many_kernel"), "?") + " -- pc_env_step_many: the env alone under pre-generated actions (SURVEY 8(d) level (i))",
                        "us_per_step": mu, "steps_per_launch": probe_res.get("many_T"), "value": cfg.n_envs / (mu * 1e-6), "unit": "env steps/s",
                        "valu_frac": per_step_flops * cfg.n_envs / (mu * 1e-6) / 1e12 / VALU_PEAK_TFLOPS,
                        "hbm_frac": per_step_bytes * cfg.n_envs / (mu * 1e-6) / 1e9 / HBM_PEAK_GBS, "state": probe_res.get("state"),
                        "method": "one launch between two HIP events on the launch stream, best of 3, after the timed epochs"}
        if mega_us is not None:
            # the timed region's dominant kernel is the persistent rollout kernel: ONE launch does n_steps env steps (+ policy steps)
            # for every env; algorithmic bytes = SURVEY's per-env-step figure x n_envs x n_steps
            dom_us = mega_us
            algo_bytes = per_step_bytes * cfg.n_envs * cfg.n_steps
            dom_name = "rollout_kernel (K9: policy step + env step + Buffer.store for all n_steps, one persistent launch)"
            dom_method = "HIP events on the launch stream around each pc_rollout launch inside the timed epochs"
            units = cfg.n_envs * cfg.n_steps
        else:
            dom_us, algo_bytes, dom_name, dom_method, units = k1_us, per_step_bytes * cfg.n_envs, k1.get("kernel"), k1.get("launch_us_method"), cfg.n_envs
        sec = dom_us * 1e-6
        achieved = algo_bytes / sec / 1e9
        D, A = obs_dim, 9
        mlp_flops = 2 * (2 * D * 256 + 256 * A + 256)                       # both MLPs, one env step (model.py:14-32)
        n_prod = {"fp16x2": 3, "bf16x3": 6, "fp32": 1}[args.policy_arith]
        track_name = ("track.json + big_track.json (" + ("interleaved env by env" if interleave else "halves") + ")") if mixed else "big_track.json"
        roof = {"kernel": dom_name, "bound": "valu",
                "achieved": per_step_flops * units / sec / 1e12, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": per_step_flops * units / sec / 1e12 / VALU_PEAK_TFLOPS, "traffic": None,
                "launch_us": dom_us, "launch_us_method": dom_method, "algorithmic_bytes_per_launch": algo_bytes,
                "env_steps_per_launch": units,
                "note": "the launch is bound by the SIMDs' vector issue port (fp32 ray geometry: 66 flop/B against a ridge of 20; PMC: "
                        "vector issue 85-89 % busy, DESIGN.md section 4.2): achieved / peak / frac price SURVEY 8(d)'s env-step flops "
                        "(" + f"{per_step_flops:.0f}" + " per env step: 28 x (rays x walls + 4) + 2 x rays + 30, the mean over the batch's tracks) against the fp32 vector peak -- most of the kernel's vector instructions are not "
                        "FMAs, so the flop fraction understates the port's occupancy; `hbm` = SURVEY 8(d)'s algorithmic bytes against the "
                        "HBM peak (the figure the task prices), `traffic` = the PMC bytes per launch; `mfma` = the policy GEMMs",
                "hbm": {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None},
                "valu": {"achieved": per_step_flops * units / sec / 1e12, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": per_step_flops * units / sec / 1e12 / VALU_PEAK_TFLOPS,
                         "counts": "SURVEY 8(d) env-step flops only"},
                "k1_standalone": k1, "env_only": env_only,
                "gae": {"kernel": "gae_kernel (K3)", "bound": "hbm", "launch_us": gae_us,
                        "achieved": GAE_BYTES * cfg.n_envs * cfg.n_steps / (gae_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": GAE_BYTES * cfg.n_envs * cfg.n_steps / (gae_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                        "algorithmic_bytes_per_launch": GAE_BYTES * cfg.n_envs * cfg.n_steps,
                        "launch_us_method": "median of 5 stand-alone launches between HIP events after the timed epochs"}}
        if mega_us is not None and args.policy == "fused":
            roof["mfma"] = {"achieved": n_prod * mlp_flops * units / sec / 1e12, "peak": MFMA_F16_PEAK_TFLOPS if n_prod > 1 else MFMA_F32_PEAK_TFLOPS,
                            "unit": "TFLOP/s", "frac": n_prod * mlp_flops * units / sec / 1e12 / (MFMA_F16_PEAK_TFLOPS if n_prod > 1 else MFMA_F32_PEAK_TFLOPS),
                            "counts": f"{n_prod} piece products x {mlp_flops} flop of the two MLPs per env step (D = {D}, unpadded), same launch"}
        out = {
            "metric": f"env steps/sec (whole node) on {track_name}, {nr} rays",
            "value": env_steps / dt, "unit": "env steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.env_dtype == "f32" else "f64", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {track_name}, num_rays={nr} ({obs_dim - 6} actual), "
                                   f"n_envs={cfg.n_envs}/GPU, n_steps={cfg.n_steps}, batch_size={cfg.batch_size}, "
                                   f"train_iters={cfg.train_iters}; one step = one PPO epoch (rollout + GAE + update)",
                       "n_envs_total": cfg.n_envs * world, "parallelism": f"env-sharded dp{world}, 1 flat grad all-reduce/minibatch",
                       "ranks": dist.get_world_size() if dist is not None else 1,
                       "rccl_ranks": rccl_live,   # ranks counted by an all-reduce of ones on the live RCCL communicator (0: no RCCL communicator exists)
                       "exchange_us": exchange_us, "rollout_ms_over_ranks": rollout_ranks_ms,
                       "visible_devices": torch.cuda.device_count(),
                       "update_path": ("multi-rank (all-reduce + clip/Adam per minibatch), " + ("captured in the epoch graph" if captured else "enqueued eagerly"))
                       if (world > 1 or args.force_collective) else "single-rank epoch graph",
                       "backend": (dist.get_backend() if dist is not None else None),
                       "gradient_exchange": (("one-shot all-reduce over peer-mapped buffers (pc_xchg)" if args.exchange == "p2p" else "torch.distributed all_reduce")
                                             if world > 1 else None),
                       "replicas_bit_identical": replicas_equal,      # (None on one rank)
                       "env_kernel": info, "policy_step": args.policy, "rollout": rollout_mode,
                       "rollout_kernel": rollout_kernel_name,
                       "policy_gemm_arithmetic": POLICY_ARITH, "hip_graphs": bool(cfg.use_graphs),
                       "fused_update": bool(cfg.fused_update), "custom_mlp_update": custom, "epoch_split": split,
                       "env_knobs": knobs, "ablate_build": int(_lib.pc_build_ablate()),
                       "numerics": "float64 kinematic state; float32 selection of each ray's wall segment, float64 refinement of the selected "
                                   "segment (observations, < 10 px tests)" if args.env_dtype == "f32"
                       else "float64 throughout: every value in the reference's operation order (glibc cos / sin by table); a float32 sweep only selects "
                            "which wall each ray's literal cast is evaluated on (ties and near-ties resolved by the literal loop over all walls)"},
            "roofline": roof,
        }
        traffic_file = os.path.join(ROOT, "profiles", "k1_traffic.json")
        if os.path.exists(traffic_file):
            try:
                tf = json.load(open(traffic_file))
                key = (f"rollout_{args.env_dtype}_n{nr}_N{cfg.n_envs}_T{cfg.n_steps}" if mega_us is not None
                       else f"{args.env_dtype}_n{nr}_N{cfg.n_envs}")
                if key in tf and not mixed:
                    out["roofline"]["traffic"] = out["roofline"]["hbm"]["traffic"] = tf[key]["hbm_bytes_per_launch"]
                    out["roofline"]["traffic_source"] = tf[key].get("source", "profiles/k1_traffic.json")
            except Exception:
                pass
        out.update(extras)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl, torch)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
