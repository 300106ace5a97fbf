"""PPO rollout / GAE / clipped-update loop of the reference's train.py (train.py:144-292), GPU-resident.

What changes against the reference is WHERE things run, not WHAT is computed:
  * the N worker processes + pipes of AsyncVectorEnv are one HIP kernel launch per step (VecCarEnv);
    actions, observations, rewards and flags never leave the GPU, and the env kernel writes straight
    into the next rows of the rollout buffer (no per-step host round trip, train.py:185-192);
  * GAE is the HIP scan kernel behind Buffer.calculate_advantages;
  * the minibatch indices of one epoch (train.py:225-230) are drawn on the host for all train_iters at
    once and shipped in one async copy; metrics are accumulated on the device and read once per epoch;
  * multi-GPU: one process per GPU, envs sharded, parameters identical on every rank, ONE flat-bucket
    all-reduce (RCCL over xGMI) of all gradients per minibatch between backward and clip_grad_norm_
    (train.py:259-260), averaged; everything after it is replicated and stays bit-identical across ranks.
Reference quirks kept on purpose: the minibatch loop runs over range(0, n_steps, batch_size)
(train.py:228), advantages are normalised per minibatch with the unbiased std (train.py:239),
the value loss is unclipped (train.py:249), logged sums are divided by train_iters (train.py:286-289).
"""
import dataclasses
import time

import numpy as np
import torch
import torch.nn as nn

from . import _capi
from ._capi import check, lib
from .buffer import Buffer
from .env import VecCarEnv
from .model import Agent


@dataclasses.dataclass
class PPOConfig:
    # train.py:72-92 defaults
    n_envs: int = 16
    n_epochs: int = 200
    n_steps: int = 1024
    batch_size: int = 512
    train_iters: int = 40
    gamma: float = 0.99
    gae_lambda: float = 0.95
    clip_ratio: float = 0.2
    ent_coef: float = 0.001
    vf_coef: float = 0.5
    learning_rate: float = 3e-4
    learning_rate_decay: float = 0.99
    max_grad_norm: float = 1.0
    reward_scaling: float = 0.1
    # additions
    track: str = "tracks/big_track.json"   # replaces the Tk file dialog (train.py:95-111,119).  A list / tuple of paths = a
                                           # mixed-track batch (BASELINE configs[4]): env i runs track (i * n_tracks) // n_envs,
                                           # rounded to blocks of 32 envs (the layout the persistent rollout kernel accepts)
    track_interleave: bool = False         # mixed tracks env by env instead (i % n_tracks: every wave holds all tracks: the persistent kernel's generic mode,
                                           # its env step once per distinct track of a wave)
    num_rays: int = 12                     # Car(num_rays=...) (car_env.py:227); 16 -> 17 rays, 32 -> 33
    env_dtype: str = "f32"
    seed: int = 0
    full_sweep: bool = False               # opt-in: iterate over all n_steps*n_envs samples per train iter
    use_graphs: bool = True                # capture the minibatch update / the rollout in HIP graphs (GPU only)
    fused_update: bool = True              # GPU only: minibatch gather, PPO loss fwd+bwd and clip+Adam as three HIP kernels
                                           # (the MLP GEMMs stay torch autograd); False = the reference's torch ops throughout
    custom_mlp: bool = True                # with fused_update: the MLP forward/backward too are HIP kernels (pc_ppo_minibatch,
                                           # no library GEMM); False = torch autograd GEMMs between the fused loss / Adam kernels
    prepared_minibatches: bool = True      # gather all minibatches of an epoch in one launch (pc_ppo_prepare) before the steps
    deferred_adam: bool = False            # single rank + prepared minibatches: the epoch's minibatch loop as pc_ppo_epoch_prepared -- the clip + Adam
                                           # step of minibatch i is taken by the forward / backward launch of minibatch i + 1 (two launches per
                                           # minibatch instead of three; the same bits).  OFF: measured SLOWER on MI355X (update 2.28 vs 1.92 ms
                                           # per epoch at the target shape, 2.07 vs 1.73 at configs[1])
    rollout_kernel: str = "auto"           # "mega": the whole rollout as one persistent launch (pc_rollout); "steps": two
                                           # kernels per step (HIP-graph replayed); "auto": mega whenever pc_rollout supports the shape
    policy: str = "fused"                  # rollout policy step: "fused" (one MFMA kernel: MLPs + draw),
                                           # "sample" (torch GEMMs + sampling kernel), "torch" (reference ops)
    capture_collectives: bool = False      # multi-rank + backend nccl (RCCL): True = the per-minibatch gradient all-reduce is captured
                                           # INSIDE the epoch's update graph (one replay per epoch, as on a single rank).  Off by default:
                                           # a captured RCCL collective has only ever run on a ONE-rank communicator here (no multi-GPU box);
                                           # False / gloo: the minibatch steps are enqueued eagerly around an eager all-reduce
    force_collective: bool = False         # test knob: take the multi-rank update path (all-reduce + pc_clip_adam) on ONE rank
    exchange: str = "rccl"                 # multi-rank gradient exchange per minibatch: "rccl" = torch.distributed all_reduce (RCCL over xGMI with
                                           # backend nccl; whatever the process group's backend is otherwise); "p2p" = the library's one-shot
                                           # all-reduce over peer-mapped buffers (pc_xchg_*: every rank writes its bucket into every peer's slot,
                                           # sums locally in rank order) -- one xGMI hop of latency instead of a ring / tree schedule, a plain
                                           # kernel, so capture_collectives can put it into the epoch graph with any backend
    bootstrap_value: str = "kernel"        # the value of the final observation for GAE (agent.get_value(next_obs), train.py:200): "kernel" = the
                                           # persistent rollout kernel's own critic pass (the fused policy step's arithmetic: fp16x2 by default,
                                           # within 4e-6 of float64 -- what val_buf's other rows hold); "fp32" = torch's fp32 Linear on next_obs,
                                           # the reference's very call
    exchange_timeout_s: float = 20.0       # exchange = "p2p": how long an exchange kernel waits for a peer's flag before it gives up; the
                                           # Trainer checks for that wherever it synchronises (run_epoch(sync=True), check_exchange(), close())
                                           # and raises ExchangeTimeout
    policy_precision: int = -1             # arithmetic of the fused policy step's GEMMs: 2 fp16x2, 1 bf16x3, 0 fp32-input MFMA;
                                           # -1 = the library's default (fp16x2).  Per Trainer (a pc_policy handle), not process-wide
    policy_split: int = -1                 # work decomposition of the fused policy step: -1 automatic by batch size, 0 never, 1 always
                                           # (hidden tiles split over a workgroup's waves; differs in fp32 summation order)
    policy_range: str = "fallback"         # weights outside the fp16x2 form's numeric domain (a weight / hidden activation that saturates in its
                                           # scaled fp16 domain; checked at every pack, include/ppocar.h pc_policy_pack_checked): "fallback" = switch
                                           # the fused policy step to precision 0 (exact fp32 chain) with a message on stderr, "raise" = PolicyRangeError
    rollout_form: int = -1                 # pc_rollout's per-handle options (include/ppocar.h PC_OPT_ROLLOUT_*; every choice gives the
    rollout_epw: int = 0                   # same bits): form -1 automatic / 0 big / 1 small / 2, 3 without the LDS 1/den table; envs per
    rollout_fast: int = 1                  # workgroup 0 automatic / 16 / 32 / 128 / 256; fast 1 / 2 / 0 (table-driven modes on / generic sweep / off)


def flatten_parameters(module):
    """Re-home all parameters (and their .grad) of `module` in two flat float32 buffers so that the
    gradient exchange is a single contiguous all-reduce and zero_grad is one memset.
    Returns (flat_param, flat_grad)."""
    params = [p for p in module.parameters()]
    total = sum(p.numel() for p in params)
    dev, dt = params[0].device, params[0].dtype
    flat = torch.empty(total, device=dev, dtype=dt)
    flat_grad = torch.zeros(total, device=dev, dtype=dt)
    off = 0
    for p in params:
        n = p.numel()
        flat[off:off + n].copy_(p.data.reshape(-1))
        p.data = flat[off:off + n].view_as(p.data)
        p.grad = flat_grad[off:off + n].view_as(p.data)
        off += n
    return flat, flat_grad


class GradExchange:
    """DDP-style gradient averaging without DDP: one all-reduce of the flat gradient bucket."""

    def __init__(self, flat_grad, world_size):
        self.flat_grad = flat_grad
        self.world_size = world_size

    def __call__(self):
        if self.world_size > 1:
            import torch.distributed as dist
            dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM)
            self.flat_grad.div_(self.world_size)


class ExchangeTimeout(RuntimeError):
    """A peer rank did not arrive at a gradient exchange within the timeout: the sums of that exchange (and every later one on
    the handle) are WRONG and this rank's replica has diverged.  The job must stop; the launcher tears the other ranks down."""


class P2PUnavailable(RuntimeError):
    """The one-shot exchange could not be set up on SOME rank (a peer's device hidden from a process, no peer access, an IPC mapping
    refused -- e.g. ranks in containers with separate IPC namespaces).  Raised on EVERY rank (the connect results are all-gathered),
    so that the job can fall back to exchange = "rccl" collectively."""


class P2PExchange:
    """pc_xchg_*: the flat gradient bucket summed over the ranks of one node by a one-shot exchange over hipIpc-mapped staging
    buffers (include/ppocar.h).  The IPC handles travel through torch.distributed (any backend); the exchange itself is one
    kernel launch on the caller's stream, in place, bit-identical on every rank."""

    def __init__(self, flat_grad, rank, world_size, device, timeout_s=20.0):
        import ctypes as C
        import socket
        import torch.distributed as dist
        self.flat_grad, self.rank, self.world = flat_grad, rank, world_size
        self.device = torch.device(device)
        self._h = None
        self.failed = False
        # hipIpc handles are meaningful inside one node only.  The node's identity is the kernel's boot id, not the hostname: two
        # containers on one node have different hostnames (and can exchange), two nodes may be given the same one.  Whether a
        # peer's DEVICE is reachable is pc_xchg_connect's check (by PCI bus id), with its own error.
        try:
            node = open("/proc/sys/kernel/random/boot_id").read().strip()
        except OSError:
            node = socket.gethostname()
        nodes = [None] * world_size
        dist.all_gather_object(nodes, node)
        if len(set(nodes)) != 1:
            raise ValueError(f"PPOConfig.exchange = 'p2p' needs all ranks on one node (got {len(set(nodes))} different boot ids): use exchange = 'rccl'")
        di = self.device.index if self.device.index is not None else torch.cuda.current_device()
        h = C.c_void_p()
        check(lib.pc_xchg_create(di, rank, world_size, flat_grad.numel(), C.byref(h)), "pc_xchg_create")
        self._h = h
        check(lib.pc_xchg_set_timeout(h, float(timeout_s)), "pc_xchg_set_timeout")
        mine = (C.c_char * _capi.PC_XCHG_HANDLE_BYTES)()
        check(lib.pc_xchg_local_handle(h, mine), "pc_xchg_local_handle")
        handles = [None] * world_size
        dist.all_gather_object(handles, bytes(mine.raw))
        blob = b"".join(handles)
        # every rank learns every rank's connect result: a connect that fails on SOME ranks only (peers in containers with their own IPC
        # namespace, a device hidden from one process) must not leave the others exchanging with a rank that is not there
        rc = lib.pc_xchg_connect(h, C.c_char_p(blob))
        why = lib.pc_last_hip_error().decode() if rc != 0 else ""
        results = [None] * world_size
        dist.all_gather_object(results, (int(rc), why))
        bad = [(r, c, w) for r, (c, w) in enumerate(results) if c != 0]
        if bad:
            lib.pc_xchg_destroy(h)
            self._h = None
            raise P2PUnavailable("pc_xchg_connect failed on rank(s) " + "; ".join(f"{r}: code {c} {w}" for r, c, w in bad))
        dist.barrier()              # every rank has mapped every peer before the first exchange

    def __call__(self):
        check(lib.pc_xchg_allreduce(self._h, self.flat_grad.data_ptr(), torch.cuda.current_stream(self.device).cuda_stream),
              "pc_xchg_allreduce")

    def status(self):
        """Synchronises the device.  Raises ExchangeTimeout if any exchange so far gave up waiting for a peer."""
        rc = lib.pc_xchg_status(self._h)
        if rc == -7:
            self.failed = True
            raise ExchangeTimeout(f"rank {self.rank}: a peer did not arrive at a gradient exchange (pc_xchg: PC_ERR_TIMEOUT); the reduced "
                                  "gradients since then are wrong -- aborting instead of training on them")
        check(rc, "pc_xchg_status")

    def close(self):
        if self._h is not None:
            import torch.distributed as dist
            torch.cuda.synchronize(self.device)
            if dist.is_initialized() and not self.failed:
                dist.barrier()      # no rank frees its staging buffer while a peer may still write into it
            # (after a timeout the peers are gone or hung: a barrier would hang this rank too; its buffer is released as it is)
            lib.pc_xchg_destroy(self._h)
            self._h = None


def ppo_loss(agent, obs, act, old_logprob, adv, ret, clip_ratio, vf_coef, ent_coef):
    """One minibatch of train.py:233-255.  `adv` is the raw advantage slice; normalisation is here."""
    _, new_logprobs, entropies, new_values = agent.get_action_and_value(obs, act)
    ratios = torch.exp(new_logprobs - old_logprob)                                            # :235
    adv = (adv - adv.mean()) / adv.std().clamp_min(1e-5)      # :238-240 torch.max(std, 1e-5), without the H2D scalar copy
    policy_loss1 = -adv * ratios                                                              # :243
    policy_loss2 = -adv * torch.clamp(ratios, 1.0 - clip_ratio, 1.0 + clip_ratio)             # :244
    policy_loss = torch.max(policy_loss1, policy_loss2).mean()                                # :245
    value_loss = 0.5 * ((new_values.view(-1) - ret) ** 2).mean()                              # :248-249
    entropy = entropies.mean()                                                                # :252
    loss = policy_loss + vf_coef * value_loss - ent_coef * entropy                            # :255
    return loss, policy_loss, value_loss, entropy


class PPOLearner:
    """The update half of train.py (train.py:216-269): optimizer, scheduler, minibatch index draws,
    clipped-PPO minibatch steps with the flat-bucket gradient exchange.  Pure torch: runs on any device
    (the multi-rank path is tested on CPU with gloo); the env and the GAE scan live in Trainer.

    On the GPU the minibatch step (gather, forward, loss, backward | clip, Adam, metric sums) is captured
    once into HIP graphs and replayed: the ~70 small kernels of one step are launch-bound when issued
    eagerly (measured 1.6 ms of host time for 0.5 ms of GPU work).  With more than one rank the capture is
    split at the gradient all-reduce, which stays an eager RCCL call between the two graphs."""

    def __init__(self, agent, cfg: PPOConfig, device, rank=0, world_size=1):
        self.agent, self.cfg, self.device, self.rank, self.world_size = agent, cfg, torch.device(device), rank, world_size
        self.flat_param, self.flat_grad = flatten_parameters(agent)
        if world_size > 1:
            import torch.distributed as dist
            dist.broadcast(self.flat_param, src=0)       # every rank starts from rank 0's parameters
        self.exchange = GradExchange(self.flat_grad, world_size)
        self.collective = world_size > 1 or bool(cfg.force_collective)     # the update has an exchange step
        if cfg.bootstrap_value not in ("kernel", "fp32"):
            raise ValueError(f"PPOConfig.bootstrap_value must be 'kernel' or 'fp32', not {cfg.bootstrap_value!r}")
        if cfg.exchange not in ("rccl", "p2p"):
            raise ValueError(f"PPOConfig.exchange must be 'rccl' or 'p2p', not {cfg.exchange!r}")
        self.p2p = None
        self.exchange_events = None      # bench.py: a list -> (start, end) HIP events around every eagerly enqueued exchange
        if cfg.exchange == "p2p" and world_size > 1 and self.device.type == "cuda":
            try:
                self.p2p = P2PExchange(self.flat_grad, rank, world_size, self.device, timeout_s=cfg.exchange_timeout_s)
                self.exchange = self._exchange_and_average      # the torch-op update paths (fused_update off) exchange through it too
            except P2PUnavailable as ex:      # raised on every rank alike: the whole job falls back together
                import sys
                print(f"[ppo_car_amd] rank {rank}: exchange = 'p2p' is not available ({ex}); falling back to exchange = 'rccl' on all ranks",
                      file=sys.stderr, flush=True)
                self.p2p = None
        self._capture_failed = False
        self.graphs = bool(cfg.use_graphs) and self.device.type == "cuda"
        self.fused = bool(cfg.fused_update) and self.device.type == "cuda" and 2 <= cfg.batch_size <= 1024
        if self.fused:    # Adam state of the fused clip+Adam kernel (pc_clip_adam): flat, on the device
            self.exp_avg = torch.zeros_like(self.flat_param)
            self.exp_avg_sq = torch.zeros_like(self.flat_param)
            self.step_count = torch.zeros(1, device=self.device)
            self.lr_dev = torch.full((1,), cfg.learning_rate, device=self.device, dtype=torch.float32)
        a1 = agent.actor[0] if isinstance(agent.actor, nn.Sequential) else None
        self.custom = (self.fused and bool(cfg.custom_mlp) and agent._std_mlp() and a1.out_features == 256
                       and lib.pc_ppo_workspace_floats(cfg.batch_size, a1.in_features, 256, agent.actor[2].out_features) > 0)
        if self.custom:
            self._ws = torch.empty(lib.pc_ppo_workspace_floats(cfg.batch_size, a1.in_features, 256, agent.actor[2].out_features),
                                   device=self.device, dtype=torch.float32)
        self._epoch_graph = None
        if self.graphs:   # capturable Adam: step count and lr live on the device, so a captured step stays valid
            lr = torch.tensor(cfg.learning_rate, device=self.device, dtype=torch.float32)
            self.optimizer = torch.optim.Adam(agent.parameters(), lr=lr, eps=1e-5, capturable=True, foreach=True)
        else:
            self.optimizer = torch.optim.Adam(agent.parameters(), lr=cfg.learning_rate, eps=1e-5)        # train.py:146
        self.scheduler = torch.optim.lr_scheduler.StepLR(self.optimizer, step_size=1, gamma=cfg.learning_rate_decay)  # :147
        self._np_rng = np.random.default_rng(cfg.seed * 7919 + rank)
        self.n_minibatches = len(range(0, cfg.n_steps, cfg.batch_size))     # train.py:228
        idx = torch.empty(cfg.train_iters, self.n_minibatches * cfg.batch_size, dtype=torch.int64)
        # two pinned staging buffers, used alternately: epochs are queued without host synchronisation, so the async
        # H2D copy of epoch k may still be pending when the host draws the indices of epoch k+1
        cuda = self.device.type == "cuda"
        self._idx_hosts = [idx.clone().pin_memory() if cuda else idx.clone() for _ in range(2)]
        self._idx_events = [torch.cuda.Event() if cuda else None for _ in range(2)]
        self._idx_turn = 0
        self._idx_dev = torch.empty_like(idx, device=self.device)
        self.metrics = torch.zeros(4, device=self.device)
        self._graph_key = None

    def current_lr(self):
        if self.fused:
            return float(self.lr_dev)
        lr = self.optimizer.param_groups[0]["lr"]
        return float(lr) if torch.is_tensor(lr) else lr

    def _dev_index(self):
        return self.device.index if self.device.index is not None else torch.cuda.current_device()

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    # ---- fused form of one minibatch (HIP kernels for everything but the MLP GEMMs) ----------------------
    def _fused_alloc(self, D, A):
        B = self.cfg.batch_size
        z = lambda *sh: torch.zeros(*sh, device=self.device, dtype=torch.float32)
        self._f = dict(obs=z(B, D), act=z(B), lp=z(B), adv=z(B), ret=z(B), dlogits=z(B, A), dvalues=z(B, 1))

    def _fused_fwd_bwd(self, idx, obs, act, logprob, adv, ret):
        """gather -> MLP forward (torch) -> loss fwd+bwd kernel -> MLP backward (torch autograd) into flat_grad"""
        cfg, f, B = self.cfg, self._f, self.cfg.batch_size
        di, st = self._dev_index(), self._stream()
        check(lib.pc_ppo_gather(di, idx.data_ptr(), B, obs.shape[1], obs.data_ptr(), act.data_ptr(), logprob.data_ptr(),
                                adv.data_ptr(), ret.data_ptr(), f["obs"].data_ptr(), f["act"].data_ptr(), f["lp"].data_ptr(),
                                f["adv"].data_ptr(), f["ret"].data_ptr(), st), "pc_ppo_gather")
        logits = self.agent.actor(f["obs"])
        values = self.agent.critic(f["obs"])
        check(lib.pc_ppo_loss(di, logits.data_ptr(), values.data_ptr(), f["act"].data_ptr(), f["lp"].data_ptr(),
                              f["adv"].data_ptr(), f["ret"].data_ptr(), B, logits.shape[1], cfg.clip_ratio, cfg.vf_coef,
                              cfg.ent_coef, f["dlogits"].data_ptr(), f["dvalues"].data_ptr(), self.metrics.data_ptr(), st),
              "pc_ppo_loss")
        self.flat_grad.zero_()
        torch.autograd.backward([logits, values], [f["dlogits"], f["dvalues"]])

    def _fused_apply(self):
        cfg = self.cfg
        check(lib.pc_clip_adam(self._dev_index(), self.flat_param.data_ptr(), self.flat_grad.data_ptr(), self.exp_avg.data_ptr(),
                               self.exp_avg_sq.data_ptr(), self.step_count.data_ptr(), self.lr_dev.data_ptr(),
                               self.flat_param.numel(), cfg.max_grad_norm, 1.0 / self.world_size, 0.9, 0.999, 1e-5,
                               self._stream()), "pc_clip_adam")

    def _exchange_and_average(self):
        """GradExchange's contract (flat_grad := the MEAN over ranks) on top of the configured transport."""
        self._sum_gradients()
        self.flat_grad.div_(self.world_size)

    def _sum_gradients(self):
        """The one exchange step per minibatch: flat_grad := SUM over ranks (the 1/W is folded into the clip + Adam kernels)."""
        ev = None
        if self.exchange_events is not None and not torch.cuda.is_current_stream_capturing():
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        if self.p2p is not None:
            self.p2p()
        else:
            import torch.distributed as dist
            dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM)
        if ev is not None:
            ev[1].record()
            self.exchange_events.append(ev)

    def _custom_apply(self):
        """clip + Adam after the gradient exchange of the custom (hand-written kernel) minibatch step: the step counter was
        advanced by the gradient kernels (apply = 2), the bucket holds the sum over ranks."""
        cfg = self.cfg
        check(lib.pc_clip_adam_advanced(self._dev_index(), self.flat_param.data_ptr(), self.flat_grad.data_ptr(), self.exp_avg.data_ptr(),
                                        self.exp_avg_sq.data_ptr(), self.step_count.data_ptr(), self.lr_dev.data_ptr(),
                                        self.flat_param.numel(), cfg.max_grad_norm, 1.0 / self.world_size, 0.9, 0.999, 1e-5,
                                        self._stream()), "pc_clip_adam_advanced")

    def custom_minibatch_step(self, idx, obs, act, logprob, adv, ret):
        """pc_ppo_minibatch: gather + forward + loss + backward (+ clip + Adam when single-rank) with no library GEMM."""
        cfg, a1, a2 = self.cfg, self.agent.actor[0], self.agent.actor[2]
        single = not self.collective
        check(lib.pc_ppo_minibatch(self._dev_index(), idx.data_ptr(), cfg.batch_size, a1.in_features, a1.out_features, a2.out_features,
                                   obs.data_ptr(), act.data_ptr(), logprob.data_ptr(), adv.data_ptr(), ret.data_ptr(),
                                   self.flat_param.data_ptr(), self.flat_grad.data_ptr(), self.exp_avg.data_ptr(),
                                   self.exp_avg_sq.data_ptr(), self.step_count.data_ptr(), self.lr_dev.data_ptr(), cfg.clip_ratio,
                                   cfg.vf_coef, cfg.ent_coef, cfg.max_grad_norm, 0.9, 0.999, 1e-5, self.metrics.data_ptr(),
                                   self._ws.data_ptr(), 1 if single else 2, self._stream()), "pc_ppo_minibatch")
        if not single:
            self._sum_gradients()
            self._custom_apply()

    def prepare_minibatches(self, idx_all, n_mb, obs, act, logprob, adv, ret):
        """pc_ppo_prepare: gather all train_iters x n_mb minibatches of the epoch in one launch (sample rows, per-sample
        scalars, advantage statistics), so that no minibatch step starts with dependent index -> row loads."""
        cfg, a1 = self.cfg, self.agent.actor[0]
        B, D = cfg.batch_size, a1.in_features
        pf = lib.pc_ppo_prepared_floats(B, D)
        total = idx_all.shape[0] * n_mb
        if getattr(self, "_prep", None) is None or self._prep.numel() != total * pf:
            self._prep = torch.empty(total * pf, device=self.device)
        rows = idx_all if idx_all.is_contiguous() and idx_all.shape[1] == n_mb * B else None
        if rows is not None:                      # minibatch m = it * n_mb + mb starts at element m * B
            check(lib.pc_ppo_prepare(self._dev_index(), rows.data_ptr(), B, total, B, D, obs.data_ptr(), act.data_ptr(),
                                     logprob.data_ptr(), adv.data_ptr(), ret.data_ptr(), self._prep.data_ptr(), self._stream()),
                  "pc_ppo_prepare")
        else:
            for it in range(idx_all.shape[0]):
                row = idx_all[it]
                check(lib.pc_ppo_prepare(self._dev_index(), row.data_ptr(), B, n_mb, B, D, obs.data_ptr(), act.data_ptr(),
                                         logprob.data_ptr(), adv.data_ptr(), ret.data_ptr(),
                                         self._prep[it * n_mb * pf:].data_ptr(), self._stream()), "pc_ppo_prepare")
        return pf

    def prepared_minibatch_step(self, m, pf):
        """pc_ppo_minibatch_prepared on block m of the prepared epoch (+ all-reduce and clip/Adam when multi-rank)."""
        cfg, a1, a2 = self.cfg, self.agent.actor[0], self.agent.actor[2]
        single = not self.collective
        check(lib.pc_ppo_minibatch_prepared(self._dev_index(), self._prep.data_ptr() + 4 * m * pf, cfg.batch_size, a1.in_features,
                                            a1.out_features, a2.out_features, self.flat_param.data_ptr(), self.flat_grad.data_ptr(),
                                            self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self.step_count.data_ptr(),
                                            self.lr_dev.data_ptr(), cfg.clip_ratio, cfg.vf_coef, cfg.ent_coef, cfg.max_grad_norm, 0.9,
                                            0.999, 1e-5, self.metrics.data_ptr(), self._ws.data_ptr(), 1 if single else 2,
                                            self._stream()), "pc_ppo_minibatch_prepared")
        if not single:
            self._sum_gradients()
            self._custom_apply()

    def _epoch_chain(self, n_mb_total):
        """pc_ppo_epoch_prepared: all minibatch steps of the epoch on the prepared blocks, two launches per minibatch (the clip +
        Adam step deferred into the next forward / backward launch) + one for the last gradient."""
        cfg, a1, a2 = self.cfg, self.agent.actor[0], self.agent.actor[2]
        if getattr(self, "_state2", None) is None:
            self._state2 = torch.empty(lib.pc_ppo_epoch_state_floats(a1.in_features, a1.out_features, a2.out_features), device=self.device)
        check(lib.pc_ppo_epoch_prepared(self._dev_index(), self._prep.data_ptr(), n_mb_total, cfg.batch_size, a1.in_features, a1.out_features,
                                        a2.out_features, self.flat_param.data_ptr(), self.flat_grad.data_ptr(), self.exp_avg.data_ptr(),
                                        self.exp_avg_sq.data_ptr(), self.step_count.data_ptr(), self.lr_dev.data_ptr(), cfg.clip_ratio,
                                        cfg.vf_coef, cfg.ent_coef, cfg.max_grad_norm, 0.9, 0.999, 1e-5, self.metrics.data_ptr(),
                                        self._ws.data_ptr(), self._state2.data_ptr(), self._stream()), "pc_ppo_epoch_prepared")

    def fused_minibatch_step(self, idx, obs, act, logprob, adv, ret):
        self._fused_fwd_bwd(idx, obs, act, logprob, adv, ret)
        if self.world_size > 1:
            self._sum_gradients()     # the 1/W is folded into pc_clip_adam
        self._fused_apply()

    def _epoch_body(self, idx_all, n_mb, args):
        """All minibatch steps of one epoch's update (train.py:223-261) with the hand-written kernels."""
        cfg = self.cfg
        B = cfg.batch_size
        if cfg.prepared_minibatches:
            pf = self.prepare_minibatches(idx_all, n_mb, *args)
            if cfg.deferred_adam and not self.collective:
                self._epoch_chain(cfg.train_iters * n_mb)
                return
            for m in range(cfg.train_iters * n_mb):
                self.prepared_minibatch_step(m, pf)
        else:
            for it in range(cfg.train_iters):
                for mb in range(n_mb):
                    self.custom_minibatch_step(idx_all[it, mb * B:(mb + 1) * B], *args)

    def _can_capture_update(self):
        """The whole epoch's update as ONE graph: always on a single rank; with an exchange step only when the all-reduce
        itself can be captured -- backend nccl (RCCL records its kernels into the capturing stream), not gloo."""
        if not self.collective:
            return True
        if self._capture_failed or not self.cfg.capture_collectives:
            return False
        if self.p2p is not None:
            return True         # the one-shot exchange is an ordinary kernel launch: capturable with any backend
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl"

    def _capture_epoch(self, idx_all, n_mb, args):
        """Capture _epoch_body.  With an exchange step the graph contains 80 x (K10, K11, RCCL all-reduce, clip+Adam): a replay
        then costs the host one call, as on a single rank.  If this RCCL / driver refuses to capture a collective, say so once
        and fall back to eager enqueueing (identical results)."""
        g = torch.cuda.CUDAGraph()
        if not self.collective:
            with torch.cuda.graph(g):
                self._epoch_body(idx_all, n_mb, args)
            return g
        import torch.distributed as dist
        if self.p2p is not None:      # plain kernels only: captured like the single-rank update
            with torch.cuda.graph(g):
                self._epoch_body(idx_all, n_mb, args)
            return g
        saved = [t.clone() for t in (self.flat_param, self.flat_grad, self.exp_avg, self.exp_avg_sq, self.step_count, self.metrics)]
        try:
            dist.all_reduce(torch.zeros(1, device=self.device))       # the communicator exists before the capture starts
            torch.cuda.synchronize(self.device)
            with torch.cuda.graph(g):
                self._epoch_body(idx_all, n_mb, args)
            return g
        except Exception as ex:      # noqa: BLE001 -- whatever the runtime raises, the eager path is the answer
            import sys
            print(f"[ppo_car_amd] rank {self.rank}: capturing the gradient all-reduce into the update graph failed ({ex!r}); "
                  "running the update eagerly", file=sys.stderr, flush=True)
            self._capture_failed = True
            torch.cuda.synchronize(self.device)
            for t, v in zip((self.flat_param, self.flat_grad, self.exp_avg, self.exp_avg_sq, self.step_count, self.metrics), saved):
                t.copy_(v)           # (a capture executes nothing; restored anyway in case a partial eager launch slipped through)
            return None

    def draw_indices(self, M):
        """train.py:225-230: per train iter a fresh shuffle of all M = n_steps*n_envs indices, of which only
        the first n_minibatches*batch_size are ever used -- i.e. a uniform sample without replacement.  Drawn
        on the host (the reference shuffles on the host too) for all iters at once, one async copy."""
        cfg = self.cfg
        K = min(self.n_minibatches * cfg.batch_size, M)   # a slice past the end of the shuffled array is just shorter
        turn = self._idx_turn
        self._idx_turn ^= 1
        if self._idx_events[turn] is not None:
            self._idx_events[turn].synchronize()          # the copy that last read this staging buffer has completed
        host_t = self._idx_hosts[turn]
        host = host_t.numpy()
        for i in range(cfg.train_iters):
            host[i, :K] = self._np_rng.permutation(M) if K == M else self._np_rng.choice(M, size=K, replace=False)
        self._idx_dev.copy_(host_t, non_blocking=True)
        if self._idx_events[turn] is not None:
            self._idx_events[turn].record(torch.cuda.current_stream(self.device))
        return self._idx_dev[:, :K]

    # ---- one minibatch, in the two halves the all-reduce separates ---------------------------------------
    def _fwd_bwd(self, obs, act, logprob, adv, ret):
        cfg = self.cfg
        loss, pl, vl, ent = ppo_loss(self.agent, obs, act, logprob, adv, ret, cfg.clip_ratio, cfg.vf_coef, cfg.ent_coef)
        self.flat_grad.zero_()                                                           # train.py:258
        loss.backward()                                                                  # :259
        return torch.stack([pl.detach(), vl.detach(), ent.detach(), loss.detach()])

    def _apply(self, terms):
        nn.utils.clip_grad_norm_(self.agent.parameters(), self.cfg.max_grad_norm)        # train.py:260
        self.optimizer.step()                                                            # :261
        with torch.no_grad():
            self.metrics += terms                                                        # :263-266

    def minibatch_step(self, obs, act, logprob, adv, ret):
        terms = self._fwd_bwd(obs, act, logprob, adv, ret)
        self.exchange()                                # the one all-reduce per minibatch
        self._apply(terms)

    # ---- HIP-graph form ------------------------------------------------------------------------------------
    def _build_graphs(self, obs, act, logprob, adv, ret):
        """Capture `idx -> gather -> forward -> loss -> backward` and `clip -> Adam` over the trajectory tensors
        given (their addresses are baked into the graphs; Buffer keeps them alive and in place across epochs)."""
        B = self.cfg.batch_size
        self._g_idx = torch.zeros(B, dtype=torch.int64, device=self.device)
        self._g_terms = torch.zeros(4, device=self.device)

        def half_a():
            i = self._g_idx
            if self.fused:
                self._fused_fwd_bwd(i, obs, act, logprob, adv, ret)
            else:
                self._g_terms.copy_(self._fwd_bwd(obs[i], act[i], logprob[i], adv[i], ret[i]))

        def half_b():
            if self.fused:
                self._fused_apply()
            else:
                self._apply(self._g_terms)

        # Warm-up on a side stream (allocator / lazy optimizer-state initialisation), then undo its effect: the optimizer
        # state is SAVED before and RESTORED after, so the graphs can be (re)built at any time -- also on a Trainer whose
        # state was just loaded from a checkpoint (train.py --resume), where the state is not "never stepped".
        saved_param, saved_metrics = self.flat_param.clone(), self.metrics.clone()
        if self.fused:
            saved_opt = {k: getattr(self, k).clone() for k in ("exp_avg", "exp_avg_sq", "step_count", "lr_dev")}
        else:
            if not self.optimizer.state:      # lazy state: one throw-away step creates it (undone below)
                saved_opt = None
            else:
                saved_opt = [{k: (v.clone() if torch.is_tensor(v) else v) for k, v in self.optimizer.state[p_].items()}
                             for p_ in self.agent.parameters()]
            saved_lr = self.optimizer.param_groups[0]["lr"].clone()
        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(3):
                half_a()
                half_b()
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        if self.fused:
            for k, v in saved_opt.items():
                getattr(self, k).copy_(v)
        else:
            for n_, p_ in enumerate(self.agent.parameters()):
                for k, v in self.optimizer.state[p_].items():
                    if torch.is_tensor(v):
                        if saved_opt is None:
                            v.zero_()          # the warm-up steps were the optimizer's first: back to "never stepped"
                        else:
                            v.copy_(saved_opt[n_][k])
            self.optimizer.param_groups[0]["lr"].copy_(saved_lr)
        self.flat_param.copy_(saved_param)
        self.metrics.copy_(saved_metrics)
        self._graph_a, self._graph_b = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        single = self.world_size == 1
        with torch.cuda.graph(self._graph_a):
            half_a()
            if single:
                half_b()
        if not single:
            with torch.cuda.graph(self._graph_b):
                half_b()
        self._graph_key = (obs.data_ptr(), act.data_ptr(), logprob.data_ptr(), adv.data_ptr(), ret.data_ptr(), obs.shape[0])

    def update(self, obs, act, logprob, adv, ret):
        """obs [M, D], the rest [M] (the flattened trajectories of train.py:209-214)."""
        cfg = self.cfg
        self.metrics.zero_()
        B = cfg.batch_size
        M = obs.shape[0]
        if cfg.full_sweep:
            n_mb = M // B
            idx_all = torch.stack([torch.randperm(M, device=self.device)[:n_mb * B] for _ in range(cfg.train_iters)])
        else:
            n_mb = self.n_minibatches
            idx_all = self.draw_indices(M)
        full = idx_all.shape[1] >= n_mb * B          # every minibatch has exactly B samples
        if self.fused and full and getattr(self, "_f", None) is None:
            self._fused_alloc(obs.shape[1], self.agent.actor[2].out_features)
        if self.custom and full and not cfg.full_sweep:
            # three tiny launches per minibatch, indices read in place from the epoch's index block; single rank +
            # graphs: the whole epoch's update (train_iters x n_mb minibatches) is ONE captured graph
            args = (obs, act, logprob, adv, ret)
            if self.graphs and self._can_capture_update():
                key = tuple(t.data_ptr() for t in args) + (M, idx_all.data_ptr())
                if self._epoch_graph is None or self._epoch_key != key:
                    torch.cuda.synchronize(self.device)
                    if cfg.prepared_minibatches:
                        self.prepare_minibatches(idx_all, n_mb, *args)      # (allocates outside the capture)
                        if getattr(self, "_state2", None) is None and cfg.deferred_adam and not self.collective:
                            a1_, a2_ = self.agent.actor[0], self.agent.actor[2]
                            self._state2 = torch.empty(lib.pc_ppo_epoch_state_floats(a1_.in_features, a1_.out_features, a2_.out_features),
                                                       device=self.device)
                        torch.cuda.synchronize(self.device)
                    self._epoch_graph = self._capture_epoch(idx_all, n_mb, args)
                    self._epoch_key = key
                if self._epoch_graph is not None:
                    self._epoch_graph.replay()
                else:                                  # the capture of the collective was refused: eager from now on
                    self._epoch_body(idx_all, n_mb, args)
            else:
                self._epoch_body(idx_all, n_mb, args)
            self._opt_started = True
            self.lr_dev.mul_(cfg.learning_rate_decay)                                    # StepLR(step_size=1), :147,:269
            return
        use_graph = self.graphs and full
        if use_graph and self._graph_key != (obs.data_ptr(), act.data_ptr(), logprob.data_ptr(), adv.data_ptr(),
                                             ret.data_ptr(), M):
            self._build_graphs(obs, act, logprob, adv, ret)
        for it in range(cfg.train_iters):                                                # :223
            for mb in range(n_mb):                                                       # :228
                idx = idx_all[it, mb * B:(mb + 1) * B]
                if idx.numel() == 0:
                    continue
                if use_graph:
                    self._g_idx.copy_(idx)
                    self._graph_a.replay()
                    if self.world_size > 1:
                        if self.fused:
                            self._sum_gradients()
                        else:
                            self.exchange()
                        self._graph_b.replay()
                elif self.fused and full:
                    self.fused_minibatch_step(idx, obs, act, logprob, adv, ret)
                else:
                    self.minibatch_step(obs[idx], act[idx], logprob[idx], adv[idx], ret[idx])
        self._opt_started = True
        if self.fused:
            self.lr_dev.mul_(cfg.learning_rate_decay)                                    # StepLR(step_size=1), :147,:269
        else:
            self.scheduler.step()                                                        # :269


class Trainer:
    def __init__(self, cfg: PPOConfig, device="cuda", rank=0, world_size=1):
        self.cfg, self.rank, self.world_size = cfg, rank, world_size
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        torch.manual_seed(cfg.seed)  # identical initial parameters on every rank (then broadcast anyway)
        track_id = None
        if isinstance(cfg.track, (list, tuple)) and len(cfg.track) > 1:
            import numpy as np
            i = np.arange(cfg.n_envs)
            nt = len(cfg.track)
            track_id = (i % nt if cfg.track_interleave else np.minimum((i // 32 * 32) * nt // cfg.n_envs, nt - 1)).astype(np.uint8)
        self.envs = VecCarEnv(cfg.n_envs, cfg.track, num_rays=cfg.num_rays, reward_scaling=cfg.reward_scaling,
                              device=self.device, dtype=cfg.env_dtype, track_id=track_id)
        self.obs_dim = (self.envs.obs_dim,)          # train.py:141
        self.act_dim = self.envs.act_dim             # train.py:142
        self.agent = Agent(self.obs_dim[0], self.act_dim).to(self.device)   # train.py:145
        self.agent.rng_seed = cfg.seed * 1000003 + rank
        self.agent.policy_precision = int(cfg.policy_precision)
        self.agent.policy_split = int(cfg.policy_split)
        if cfg.policy_range not in ("fallback", "raise"):
            raise ValueError(f"PPOConfig.policy_range must be 'fallback' or 'raise', not {cfg.policy_range!r}")
        self.agent.policy_range = cfg.policy_range
        for name, default in (("rollout_form", -1), ("rollout_epw", 0), ("rollout_fast", 1)):
            if int(getattr(cfg, name)) != default:
                self.envs.set_option(name, int(getattr(cfg, name)))
        # rollout_kernel="steps" (asked for by name: the tests' reference path) runs the GENERIC env-step kernel K1 -- the independent
        # implementation the persistent kernels, and K1f, which shares their env step, are compared with bit for bit.  The per-step rollout
        # that "auto" / "mega" fall back to (a torch policy, a shape pc_rollout refuses) takes pc_env_step's automatic choice.
        if cfg.rollout_kernel == "steps":
            self.envs.set_option("step_form", 1)
        self.learner = PPOLearner(self.agent, cfg, self.device, rank, world_size)
        self.optimizer, self.scheduler = self.learner.optimizer, self.learner.scheduler
        self.buffer = Buffer(self.obs_dim, cfg.n_steps, cfg.n_envs, self.device, cfg.gamma, cfg.gae_lambda)   # :152
        N = cfg.n_envs
        self.next_obs = torch.empty(N, *self.obs_dim, device=self.device)
        self.next_term = torch.zeros(N, device=self.device)      # train.py:165-166
        self.next_trunc = torch.zeros(N, device=self.device)
        self.actions = torch.empty(N, dtype=torch.int64, device=self.device)
        mixed = isinstance(cfg.track, (list, tuple)) and len(cfg.track) > 1
        self.envs.reset(options=None if mixed else {"track_path": cfg.track}, out=self.next_obs)   # train.py:159
        self.global_step_idx = 0
        self.epoch = 0
        self.start_time = time.time()
        self.profile_stride = 0      # bench.py: bracket every k-th env-step launch with events on the launch stream
        self.k1_events = []
        self.phase_events = None     # bench.py: list of (start, rollout_end, update_end) events per epoch
        self.rng_base = torch.zeros(1, dtype=torch.int64, device=self.device)   # device-side Philox offset base
        self._boot_val = self._rew_sum = None
        self._aux_valid = False
        self._rollout_graph = None
        self._eager_rollouts = 0
        self.rollout_mode = None
        self.mega_events = None      # bench.py: list of (start, end) events around each pc_rollout launch

    # ---- train.py:173-195 ---------------------------------------------------------------------------
    @torch.no_grad()
    def _rollout_body(self, events=False):
        cfg, buf, envs, agent = self.cfg, self.buffer, self.envs, self.agent
        T = cfg.n_steps
        buf.obs_buf[0].copy_(self.next_obs)
        buf.term_buf[0].copy_(self.next_term)       # flags that preceded obs 0 (train.py:176-177)
        buf.trunc_buf[0].copy_(self.next_trunc)
        fused = cfg.policy == "fused" and agent.pack_policy()   # weights are constant during the rollout: one LDS image
        for t in range(T):
            obs = buf.obs_buf[t]
            if cfg.policy == "torch":
                actions, logprobs, _, values = agent.get_action_and_value(obs)   # train.py:181
                buf.logprob_buf[t].copy_(logprobs)
                buf.val_buf[t].copy_(values.view(-1))
                buf.act_buf[t].copy_(actions)        # stored as float32 like the reference (buffer.py:13)
            elif fused:   # one kernel: both MLPs + draw; Philox offset = rollout-step index + the device-side base
                actions, _, _ = agent.act(obs, out_action=self.actions, out_logprob=buf.logprob_buf[t], out_value=buf.val_buf[t],
                                          out_action_f32=buf.act_buf[t], fused=True, repack=False, offset=t, offset_dev=self.rng_base)
            else:         # "sample": torch GEMMs + the sampling kernel
                actions, _, _ = agent.act(obs, out_action=self.actions, out_logprob=buf.logprob_buf[t], out_value=buf.val_buf[t],
                                          out_action_f32=buf.act_buf[t], fused=False)
            last = t == T - 1
            out = (self.next_obs if last else buf.obs_buf[t + 1], buf.rew_buf[t],
                   self.next_term if last else buf.term_buf[t + 1], self.next_trunc if last else buf.trunc_buf[t + 1])
            if events and self.profile_stride and t % self.profile_stride == 0:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                envs.step(actions, out=out)
                e1.record()
                self.k1_events.append((e0, e1))
            else:
                envs.step(actions, out=out)          # train.py:185 -- zero-copy into the buffer rows
        self.rng_base += T                           # next rollout draws from fresh Philox counters

    @torch.no_grad()
    def _rollout_mega(self):
        """pc_rollout: policy step + env step + Buffer.store for all n_steps in ONE persistent launch."""
        cfg, buf, agent = self.cfg, self.buffer, self.agent
        if not agent.pack_policy():
            return False
        buf.obs_buf[0].copy_(self.next_obs)
        buf.term_buf[0].copy_(self.next_term)
        buf.trunc_buf[0].copy_(self.next_trunc)
        ev = None
        if self.mega_events is not None:    # bench.py: HIP events on the launch stream around the one launch
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        if self._boot_val is None:
            self._boot_val = torch.empty(cfg.n_envs, device=self.device)       # the final observation's value (train.py:200)
            self._rew_sum = torch.empty(cfg.n_envs, device=self.device)        # per-env reward totals (train.py:272)
        rc = lib.pc_rollout(self.envs._h, agent._image_handle, agent._image.data_ptr(), cfg.n_steps, float(cfg.reward_scaling),
                              int(agent.rng_seed), 0, self.rng_base.data_ptr(), buf.obs_buf.data_ptr(), buf.act_buf.data_ptr(),
                              buf.rew_buf.data_ptr(), buf.val_buf.data_ptr(), buf.term_buf.data_ptr(), buf.trunc_buf.data_ptr(),
                              buf.logprob_buf.data_ptr(), self.next_obs.data_ptr(), self.next_term.data_ptr(),
                              self.next_trunc.data_ptr(), self._boot_val.data_ptr(), self._rew_sum.data_ptr(),
                              torch.cuda.current_stream(self.device).cuda_stream)
        if rc == -5:       # PC_ERR_UNSUPPORTED: shape outside the persistent kernel's menu
            return False
        check(rc, "pc_rollout")
        self._aux_valid = True   # the launch also delivered the bootstrap values and the reward totals of THIS rollout
        if ev is not None:
            ev[1].record()
            self.mega_events.append(ev)
        self.rng_base += cfg.n_steps
        return True

    def rollout(self):
        """n_steps vector-env steps into the buffer.
        mega : one persistent launch for the whole rollout (pc_rollout) -- large batches.
        steps: two kernels per step (fused policy step + env step); with use_graphs the whole sequence is captured
               into ONE HIP graph after a first eager pass and replayed (at small n_envs the per-step host work, ~50 us
               of Python / ctypes, exceeds the GPU work).
        All forms draw from the same Philox counters and produce bit-identical buffers."""
        cfg = self.cfg
        mode = cfg.rollout_kernel
        if mode == "auto":
            mode = "mega" if cfg.n_envs >= 256 else "steps"
        done = False
        self._aux_valid = False
        if mode == "mega" and cfg.policy == "fused" and self.device.type == "cuda" and not self.profile_stride:
            done = self._rollout_mega()
            self.rollout_mode = "mega" if done else "steps"
        if not done:
            graphable = cfg.use_graphs and cfg.policy == "fused" and self.device.type == "cuda" and not self.profile_stride
            if self._rollout_graph is not None and self._rollout_graph_precision != self.agent.policy_precision:
                self._rollout_graph = None      # the agent left the fp16x2 domain and switched arithmetic: the captured launches are the old form's
                self._eager_rollouts = 0
            if graphable and self._rollout_graph is None and self._eager_rollouts >= 1:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._rollout_body()
                self._rollout_graph = g
                self._rollout_graph_precision = self.agent.policy_precision
            if graphable and self._rollout_graph is not None:
                self._rollout_graph.replay()
                self.rollout_mode = "steps-graph"
            else:
                self._rollout_body(events=True)
                self._eager_rollouts += 1
                self.rollout_mode = "steps-eager"
        self.buffer.ptr = cfg.n_steps
        self.global_step_idx += cfg.n_envs * cfg.n_steps * self.world_size   # train.py:174, whole job

    # ---- train.py:197-269 ---------------------------------------------------------------------------
    def update(self):
        buf, agent = self.buffer, self.agent
        with torch.no_grad():
            # train.py:200 -- the persistent rollout kernel has already evaluated the critic on the final observation
            in_kernel = self._aux_valid and self.cfg.bootstrap_value == "kernel"
            next_values = (self._boot_val if in_kernel else agent.get_value(self.next_obs)).reshape(1, -1)
            adv, ret = buf.calculate_advantages(next_values, self.next_term.reshape(1, -1),
                                                self.next_trunc.reshape(1, -1))                   # :203
        obs, act, _val, logprob = buf.get()                                                      # :206
        self.learner.update(obs.view(-1, *self.obs_dim), act.view(-1), logprob.view(-1), adv.view(-1), ret.view(-1))
        self._aux_valid = False     # the in-kernel bootstrap values belong to THAT rollout and THOSE parameters only

    def run_epoch(self, sync=True):
        """One epoch = rollout + update.  Returns the reference's scalars (train.py:286-292) when sync.
        sync="lazy" (one rank): the epoch is only ENQUEUED; its scalars travel to pinned host memory behind it, and the call returns the
        scalars of the epoch BEFORE (None on the first call) -- the host never waits for the epoch it has just launched, so the device
        runs epoch after epoch back to back while the caller prints and logs one epoch late (flush_scalars() hands over the last one;
        train.py --lazy-logging.  Worth ~1 % at 65536 envs: the synchronous fetch of five floats per 17 ms epoch was never the cost)."""
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if self.phase_events is not None else None
        if ev:
            ev[0].record()
        self.rollout()
        if ev:
            ev[1].record()
        with torch.no_grad():   # train.py:272; the persistent rollout kernel delivers per-env totals (no second pass over rew_buf)
            rew_mean = (self._rew_sum.sum() / float(self.cfg.n_steps * self.cfg.n_envs)) if self._aux_valid else self.buffer.rew_buf.mean()
        self.update()
        if ev:
            ev[2].record()
            self.phase_events.append(ev)
        self.epoch += 1
        if not sync:
            return None
        if sync == "lazy" and self.world_size == 1 and self.device.type == "cuda":
            prev = self.flush_scalars()
            L = self.learner
            with torch.no_grad():
                lr = L.lr_dev.reshape(1).to(torch.float32) if L.fused else torch.full((1,), float(L.current_lr()), device=self.device)
                rng = getattr(self.agent, "_range_dev", None)      # the fp16x2 domain's status word of the last pack (it may have run inside a graph)
                rng = rng.to(torch.float32) if rng is not None else torch.zeros(1, device=self.device)
                dev = torch.cat([L.metrics / self.cfg.train_iters, rew_mean.reshape(1).to(torch.float32), lr, rng])
            host = torch.empty(7, dtype=torch.float32, pin_memory=True)
            host.copy_(dev, non_blocking=True)
            done = torch.cuda.Event()
            done.record()
            self._pending_scalars = (host, done, self.global_step_idx)
            return prev
        self.flush_scalars()        # (a switch from lazy to synchronous calls drops nothing silently: the pending epoch is waited for)
        self.check_exchange()       # (synchronises; the scalars below are fetched anyway) a timed-out exchange stops the job HERE
        if self.device.type == "cuda":
            self.agent.check_policy_range(sync=True)    # weights outside the policy arithmetic's domain: precision 0 from the next rollout on (or PolicyRangeError)
        m = (self.learner.metrics / self.cfg.train_iters).tolist()   # divided by train_iters, not by #minibatches
        avg_reward = float(rew_mean) / self.cfg.reward_scaling       # train.py:272-274
        if self.world_size > 1:
            import torch.distributed as dist
            t = torch.tensor(m + [avg_reward], device=self.device)
            dist.all_reduce(t)
            t /= self.world_size
            *m, avg_reward = t.tolist()
        elapsed = time.time() - self.start_time
        return {"losses/policy_loss": m[0], "losses/value_loss": m[1], "losses/entropy": m[2], "losses/total_loss": m[3],
                "charts/avg_reward": avg_reward, "charts/learning_rate": self.learner.current_lr(),
                "charts/SPS": self.global_step_idx / max(elapsed, 1e-9), "global_step": self.global_step_idx,
                "elapsed": elapsed}

    def flush_scalars(self):
        """The scalars of the last epoch run with sync="lazy" (waits for THAT epoch only), or None."""
        pend, self._pending_scalars = getattr(self, "_pending_scalars", None), None
        if pend is None:
            return None
        host, done, gstep = pend
        done.synchronize()
        m = host.tolist()
        lr = m[5]
        if m[6] != 0.0 and self.device.type == "cuda":
            self.agent.check_policy_range(sync=True)    # weights outside the policy arithmetic's domain: precision 0 from the next rollout on (or PolicyRangeError)
        elapsed = time.time() - self.start_time
        return {"losses/policy_loss": m[0], "losses/value_loss": m[1], "losses/entropy": m[2], "losses/total_loss": m[3],
                "charts/avg_reward": m[4] / self.cfg.reward_scaling, "charts/learning_rate": lr,
                "charts/SPS": gstep / max(elapsed, 1e-9), "global_step": gstep, "elapsed": elapsed}

    # ---- checkpoint / resume (the reference only saves agent.state_dict(), train.py:283,301) ----
    def state_dict(self):
        """Everything needed to continue a run bit-for-bit: policy, optimizer, lr, env state, RNG counters."""
        L = self.learner
        opt = ({"exp_avg": L.exp_avg, "exp_avg_sq": L.exp_avg_sq, "step_count": L.step_count, "lr_dev": L.lr_dev} if L.fused
               else {"optimizer": L.optimizer.state_dict(), "scheduler": L.scheduler.state_dict()})
        return {"agent": self.agent.state_dict(), "opt": opt, "fused": L.fused, "env": self.envs.get_state(),
                "next_obs": self.next_obs, "next_term": self.next_term, "next_trunc": self.next_trunc,
                "rng_base": self.rng_base, "np_rng": L._np_rng.bit_generator.state, "epoch": self.epoch,
                "global_step_idx": self.global_step_idx, "agent_rng_offset": self.agent._rng_offset,
                "elapsed": time.time() - self.start_time, "config": dataclasses.asdict(self.cfg)}

    def load_state_dict(self, sd):
        L = self.learner
        if sd["fused"] != L.fused:
            raise ValueError("checkpoint was written with a different update path (fused_update)")
        self._aux_valid = False     # bootstrap values / reward totals of an earlier rollout do not belong to the loaded state
        self._rollout_graph = None  # a captured per-step rollout belongs to the state it was captured for (it is re-captured after one eager pass)
        self._eager_rollouts = 0
        with torch.no_grad():
            self.agent.load_state_dict(sd["agent"])        # parameters are views into the flat buffer: copied in place
            if L.fused:
                for k in ("exp_avg", "exp_avg_sq", "step_count", "lr_dev"):
                    getattr(L, k).copy_(sd["opt"][k])
            else:
                L.optimizer.load_state_dict(sd["opt"]["optimizer"])
                L.scheduler.load_state_dict(sd["opt"]["scheduler"])
            self.envs.set_state(**sd["env"])
            self.next_obs.copy_(sd["next_obs"]); self.next_term.copy_(sd["next_term"]); self.next_trunc.copy_(sd["next_trunc"])
            self.rng_base.copy_(sd["rng_base"])
        L._np_rng.bit_generator.state = sd["np_rng"]
        L._opt_started = True
        self.epoch, self.global_step_idx = sd["epoch"], sd["global_step_idx"]
        self.start_time = time.time() - float(sd.get("elapsed", 0.0))     # charts/SPS = global_step / elapsed (train.py:292)
        self.agent._rng_offset = sd["agent_rng_offset"]

    def check_exchange(self):
        """exchange = "p2p": synchronise and raise ExchangeTimeout if an exchange kernel gave up waiting for a peer (its sums, and
        every later one's, are wrong).  Called by run_epoch(sync=True) and close(); a caller that queues epochs without
        synchronising (bench.py) calls it where it synchronises."""
        if self.learner.p2p is not None:
            self.learner.p2p.status()

    def close(self):
        try:
            self.check_exchange()           # a peer that never arrived at an exchange surfaces here at the latest
        finally:                            # ... and the handles are released either way
            if self.learner.p2p is not None:
                self.learner.p2p.close()
            self.envs.close()
