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
        dist = torch.distributions.Categorical(logits=logits)
        if action is None:
            action = dist.sample()
        logprob = dist.log_prob(action)
        entropy = dist.entropy()
        return action, logprob, entropy, self.get_value(x)

    @torch.no_grad()
    def act(self, x, out_action=None, out_logprob=None):
        """Rollout-time variant of get_action_and_value(x): same distribution, the sampling tail
        (softmax, draw, log_prob) is one HIP kernel (pc_sample, counter-based Philox stream keyed by
        (rng_seed, call counter)).  Returns action int64 [N], logprob [N], value [N]."""
        logits = self.actor(x).contiguous()
        N, A = logits.shape
        action = out_action if out_action is not None else torch.empty(N, dtype=torch.int64, device=x.device)
        logprob = out_logprob if out_logprob is not None else torch.empty(N, dtype=torch.float32, device=x.device)
        dev = x.device.index if x.device.index is not None else torch.cuda.current_device()
        check(lib.pc_sample(dev, logits.data_ptr(), N, A, int(self.rng_seed), self._rng_offset, action.data_ptr(),
                            logprob.data_ptr(), None, torch.cuda.current_stream(x.device).cuda_stream), "pc_sample")
        self._rng_offset += 1
        return action, logprob, self.critic(x).view(-1)
