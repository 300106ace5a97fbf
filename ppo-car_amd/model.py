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

    @torch.no_grad()
    def pack_policy(self):
        """Pack the current weights into the fused policy kernel's LDS image (once per rollout: the weights do
        not change while it runs).  Returns False when the shape is outside the kernel's menu."""
        self._image_ok = False
        if not self._std_mlp():
            return False
        a1, a2, c1, c2 = self.actor[0], self.actor[2], self.critic[0], self.critic[2]
        D, H, A = a1.in_features, a1.out_features, a2.out_features
        n = lib.pc_policy_image_floats(D, H, A)
        if n < 0 or c1.out_features != H or c2.out_features != 1:
            return False
        dev = a1.weight.device
        if getattr(self, "_image", None) is None or self._image.numel() != n or self._image.device != dev:
            self._image = torch.empty(n, dtype=torch.float32, device=dev)
        di = dev.index if dev.index is not None else torch.cuda.current_device()
        check(lib.pc_policy_pack(di, D, H, A, a1.weight.data_ptr(), a1.bias.data_ptr(), a2.weight.data_ptr(), a2.bias.data_ptr(),
                                 c1.weight.data_ptr(), c1.bias.data_ptr(), c2.weight.data_ptr(), c2.bias.data_ptr(),
                                 self._image.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), "pc_policy_pack")
        self._image_ok = True
        self._image_prec = lib.pc_policy_precision(D, H, A)
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
            stale = True
            if self._std_mlp() and getattr(self, "_image_ok", False):
                a1_, a2_ = self.actor[0], self.actor[2]
                stale = self._image_prec != lib.pc_policy_precision(a1_.in_features, a1_.out_features, a2_.out_features)
            ok = self.pack_policy() if (repack or stale) else True
            if ok:
                a1, a2 = self.actor[0], self.actor[2]
                value = out_value if out_value is not None else torch.empty(N, dtype=torch.float32, device=dev)
                check(lib.pc_policy_act(di, x.data_ptr(), N, x.shape[1], a1.out_features, a2.out_features, self._image.data_ptr(),
                                        int(self.rng_seed), int(offset), ptr(offset_dev), action.data_ptr(),
                                        ptr(out_action_f32), logprob.data_ptr(), value.data_ptr(), ptr(out_logits), stream),
                      "pc_policy_act")
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
