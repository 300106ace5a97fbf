"""Actor / critic of the reference (lib/model.py): two 1-hidden-layer MLPs (D->256->9, D->256->1),
orthogonal init, categorical policy.  Stays in PyTorch-ROCm (rocBLAS GEMMs); the sampling tail can
optionally run through the library's fused sample kernel (pc_sample)."""
import numpy as np
import torch
import torch.nn as nn

from ._capi import check, lib


def layer_init(layer, std=np.sqrt(2), bias_const=0.0):  # model.py:6-9
    torch.nn.init.orthogonal_(layer.weight, std)
    torch.nn.init.constant_(layer.bias, bias_const)
    return layer


class PolicyRangeError(RuntimeError):
    """Agent.policy_range = "raise": the weights left the numeric domain of the fused policy step's fp16 x 2 arithmetic."""


class Agent(nn.Module):
    def __init__(self, num_inputs, num_outputs, hidden_size=256):  # model.py:12-26
        super().__init__()
        self.actor = nn.Sequential(
            layer_init(nn.Linear(num_inputs, hidden_size)),
            nn.ReLU(),
            layer_init(nn.Linear(hidden_size, num_outputs), std=0.01),
        )
        self.critic = nn.Sequential(
            layer_init(nn.Linear(num_inputs, hidden_size)),
            nn.ReLU(),
            layer_init(nn.Linear(hidden_size, 1), std=1.0),
        )
        self._rng_offset = 0
        self.rng_seed = 0

    def forward(self, x):  # model.py:28-29
        return self.actor(x)

    def get_value(self, x):  # model.py:31-32
        return self.critic(x)

    def get_action_and_value(self, x, action=None):  # model.py:34-41
        logits = self.actor(x)
        # validate_args=False: the reference leaves argument validation on (a host-synchronising check of the
        # logits / actions, no arithmetic); it has to be off inside a captured HIP graph
        dist = torch.distributions.Categorical(logits=logits, validate_args=False)
        if action is None:
            action = dist.sample()
        logprob = dist.log_prob(action)
        entropy = dist.entropy()
        return action, logprob, entropy, self.get_value(x)

    def _std_mlp(self):
        return all(isinstance(m, nn.Sequential) and len(m) == 3 and isinstance(m[0], nn.Linear) and isinstance(m[1], nn.ReLU)
                   and isinstance(m[2], nn.Linear) for m in (self.actor, self.critic))

    # ---- the fused policy step's configuration lives in a pc_policy HANDLE of this agent (include/ppocar.h): arithmetic form
    # (`policy_precision`: 0 fp32-input MFMA, 1 bf16x3, 2 fp16x2, -1 = the library's default, fp16x2) and work decomposition
    # (`policy_split`: -1 automatic by batch size, 0 never, 1 always).  Two agents with different forms coexist; the library keeps
    # no process-wide setting.
    policy_precision = -1
    policy_split = -1

    def _policy_handle(self):
        """The pc_policy handle for the current (device, shape, policy_precision, policy_split); None outside the kernel's menu.
        Handles are cached per key and live as long as the agent."""
        import ctypes as C
        a1, a2 = self.actor[0], self.actor[2]
        dev = a1.weight.device
        di = dev.index if dev.index is not None else torch.cuda.current_device()
        cache = self.__dict__.setdefault("_policy_handles", {})
        key = (di, a1.in_features, a1.out_features, a2.out_features, max(-1, int(self.policy_precision)), max(-1, int(self.policy_split)))
        if key not in cache:
            h = C.c_void_p()
            rc = lib.pc_policy_create(*key, C.byref(h))
            cache[key] = h if rc == 0 else None
        return cache[key]

    def policy_form(self):
        """(precision, split, image_floats) the fused policy step of this agent runs with, or None."""
        import ctypes as C
        h = self._policy_handle() if self._std_mlp() else None
        if h is None:
            return None
        pr, sp, n = C.c_int(), C.c_int(), C.c_int64()
        check(lib.pc_policy_get(h, C.byref(pr), C.byref(sp), C.byref(n)), "pc_policy_get")
        return pr.value, sp.value, n.value

    def __del__(self):
        for h in self.__dict__.get("_policy_handles", {}).values():
            if h is not None:
                lib.pc_policy_destroy(h)

    # ---- the fp16 x 2 form's numeric domain (include/ppocar.h: pc_policy_pack_checked).  Its operands SATURATE at fp16's range in
    # their scaled domains (|W1| > 4094, |W2| > 1023.5, a hidden activation beyond 255.87) -- implausible on this env with sane
    # training, but a diverging run would be silently wrong.  Every pack therefore writes a range status on the device; it is read
    # back WITHOUT synchronising (an async copy into pinned memory, looked at on the next pack / wherever the caller synchronises
    # anyway), and when it is set the agent switches to precision 0 -- the exact fp32 chain, no domain limits -- loudly (stderr, once
    # per switch) and for good.  `policy_range = "raise"` raises PolicyRangeError instead; check_policy_range(sync=True) looks NOW.
    policy_range = "fallback"

    def check_policy_range(self, sync=False):
        """True if the weights of the last pack were inside the arithmetic form's domain (or nothing is known yet).  sync=True waits for the
        status of the last pack; otherwise only a status that has already arrived is looked at."""
        ev = self.__dict__.get("_range_event")
        if ev is None:
            if not sync or self.__dict__.get("_range_dev") is None:
                return True
            mask = int(self._range_dev.item())      # (no copy in flight -- e.g. the pack ran inside a graph: read the device word itself)
        else:
            if sync:
                ev.synchronize()
            elif not ev.query():
                return True
            mask = int(self._range_host[0])
            self._range_event = None
        if mask == 0:
            return True
        what = ", ".join(n for b, n in ((1, "a first-layer weight beyond 4094"), (2, "an output-layer weight beyond 1023.5"),
                                        (4, "a hidden unit that can exceed 255.87")) if mask & b)
        if self.policy_range == "raise":
            raise PolicyRangeError(f"the policy weights left the fp16x2 form's numeric domain ({what}): use policy_precision = 0")
        import sys
        print(f"[ppo_car_amd] policy weights left the fp16x2 form's numeric domain ({what}); the fused policy step now runs in "
              "precision 0 (fp32-input MFMA, the exact fp32 chain)", file=sys.stderr, flush=True)
        self.policy_precision = 0
        self._image_ok = False
        return False

    @torch.no_grad()
    def pack_policy(self):
        """Pack the current weights into the fused policy kernel's LDS image (once per rollout: the weights do
        not change while it runs).  Returns False when the shape is outside the kernel's menu."""
        self._image_ok = False
        self.check_policy_range()      # the status of an EARLIER pack, if it has arrived: may switch this pack to precision 0
        if not self._std_mlp():
            return False
        a1, a2, c1, c2 = self.actor[0], self.actor[2], self.critic[0], self.critic[2]
        form = self.policy_form()
        if form is None or c1.out_features != a1.out_features or c2.out_features != 1:
            return False
        n = form[2]
        dev = a1.weight.device
        if getattr(self, "_image", None) is None or self._image.numel() != n or self._image.device != dev:
            self._image = torch.empty(n, dtype=torch.float32, device=dev)
        h = self._policy_handle()
        if getattr(self, "_range_dev", None) is None or self._range_dev.device != dev:
            self._range_dev = torch.zeros(1, dtype=torch.int32, device=dev)
            self._range_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        check(lib.pc_policy_pack_checked(h, a1.weight.data_ptr(), a1.bias.data_ptr(), a2.weight.data_ptr(), a2.bias.data_ptr(),
                                           c1.weight.data_ptr(), c1.bias.data_ptr(), c2.weight.data_ptr(), c2.bias.data_ptr(),
                                           self._image.data_ptr(), self._range_dev.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
              "pc_policy_pack_checked")
        if form[0] == 2 and not torch.cuda.is_current_stream_capturing() and self.__dict__.get("_range_event") is None:
            # (one status in flight at a time: the pinned word is not overwritten before it was looked at.  Inside a graph capture the
            # status stays on the device; check_policy_range(sync=True) reads it there, where the caller synchronises anyway.)
            self._range_host.copy_(self._range_dev, non_blocking=True)
            self._range_event = torch.cuda.Event()
            self._range_event.record()
        self._image_ok = True
        self._image_handle = h       # the image belongs to the handle that packed it
        return True

    @torch.no_grad()
    def act(self, x, out_action=None, out_logprob=None, out_value=None, out_action_f32=None, out_logits=None, fused=True,
            repack=True, offset_dev=None, offset=None):
        """Rollout-time variant of get_action_and_value(x): same distribution, no autograd.
        fused=True: ONE HIP kernel (pc_policy_act) -- both MLPs on the fp32 matrix cores, the categorical
        draw, log_prob and the value; falls back to the two-kernel form when the shape is outside its menu.
        repack=False reuses the weight image of the last pack_policy() (the rollout packs once).
        fused=False: torch GEMMs for the MLPs + the sampling-tail kernel (pc_sample).
        Counter-based Philox stream keyed by (rng_seed, offset [+ *offset_dev]); offset defaults to a per-agent
        call counter.  Returns action int64 [N], logprob [N], value [N] (written into the out_* tensors when given)."""
        if offset is None:
            offset = self._rng_offset
            self._rng_offset += 1
        N = x.shape[0]
        dev = x.device
        action = out_action if out_action is not None else torch.empty(N, dtype=torch.int64, device=dev)
        logprob = out_logprob if out_logprob is not None else torch.empty(N, dtype=torch.float32, device=dev)
        di = dev.index if dev.index is not None else torch.cuda.current_device()
        stream = torch.cuda.current_stream(dev).cuda_stream
        ptr = lambda t: t.data_ptr() if t is not None else None
        if fused and x.is_contiguous() and x.dtype == torch.float32:
            h = self._policy_handle() if self._std_mlp() else None
            stale = not getattr(self, "_image_ok", False) or getattr(self, "_image_handle", None) is not h
            ok = h is not None and (self.pack_policy() if (repack or stale) else True)
            if ok:
                value = out_value if out_value is not None else torch.empty(N, dtype=torch.float32, device=dev)
                check(lib.pc_policy_act(h, x.data_ptr(), N, self._image.data_ptr(), int(self.rng_seed), int(offset), ptr(offset_dev),
                                          action.data_ptr(), ptr(out_action_f32), logprob.data_ptr(), value.data_ptr(), ptr(out_logits),
                                          stream), "pc_policy_act")
                return action, logprob, value
        if offset_dev is not None:
            raise NotImplementedError("a device-side RNG offset needs the fused policy kernel")
        logits = self.actor(x).contiguous()
        A = logits.shape[1]
        check(lib.pc_sample(di, logits.data_ptr(), N, A, int(self.rng_seed), int(offset), action.data_ptr(),
                            logprob.data_ptr(), None, stream), "pc_sample")
        value = self.critic(x).view(-1)
        if out_value is not None:
            out_value.copy_(value)
            value = out_value
        if out_action_f32 is not None:
            out_action_f32.copy_(action)
        if out_logits is not None:
            out_logits.copy_(logits)
        return action, logprob, value
