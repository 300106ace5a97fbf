"""Rollout buffer with the reference's interface (lib/buffer.py): same constructor, store(),
calculate_advantages(), get().  Storage is torch tensors on the GPU; the GAE(lambda) scan is the
HIP kernel behind pc_gae (bit-exact with the reference's torch expression, buffer.py:51-63)."""
import torch

from ._capi import check, lib


class Buffer:
    def __init__(self, obs_dim, size, num_envs, device, gamma=0.99, gae_lambda=0.95):
        # buffer.py:9-20 -- seven zero-initialised float32 tensors [T, N, ...]; actions are stored as float32
        self.capacity = size
        self.num_envs = num_envs
        self.device = torch.device(device)
        z = lambda *s: torch.zeros(s, dtype=torch.float32, device=self.device)
        self.obs_buf = z(size, num_envs, *obs_dim)
        self.act_buf = z(size, num_envs)
        self.rew_buf = z(size, num_envs)
        self.val_buf = z(size, num_envs)
        self.term_buf = z(size, num_envs)
        self.trunc_buf = z(size, num_envs)
        self.logprob_buf = z(size, num_envs)
        self.gamma, self.gae_lambda = gamma, gae_lambda
        self.ptr = 0

    def store(self, obs, act, rew, val, term, trunc, logprob):
        """buffer.py:22-34.  An argument that already IS the row view (obs_buf[ptr] etc., handed out by
        row()) is not copied again -- that is how the env kernel writes straight into the buffer."""
        p = self.ptr
        for buf, v in ((self.obs_buf, obs), (self.act_buf, act), (self.rew_buf, rew), (self.val_buf, val),
                       (self.term_buf, term), (self.trunc_buf, trunc), (self.logprob_buf, logprob)):
            row = buf[p]
            if not (torch.is_tensor(v) and v.data_ptr() == row.data_ptr() and v.shape == row.shape and v.dtype == row.dtype):
                row.copy_(v)
        self.ptr += 1

    def row(self, t):
        """Views of row t (obs, act, rew, val, term, trunc, logprob) for zero-copy producers."""
        return (self.obs_buf[t], self.act_buf[t], self.rew_buf[t], self.val_buf[t], self.term_buf[t], self.trunc_buf[t],
                self.logprob_buf[t])

    def calculate_advantages(self, last_vals, last_terminateds, last_truncateds):
        """buffer.py:36-64: GAE(lambda) with separate terminated / truncated masks -> (adv_buf, ret_buf)."""
        assert self.ptr == self.capacity, "Buffer not full"
        if self.device.type != "cuda":
            raise RuntimeError("Buffer.calculate_advantages runs the HIP GAE kernel: the buffer must live on the GPU")
        T, N = self.capacity, self.num_envs
        f = lambda t: t.detach().to(device=self.device, dtype=torch.float32).reshape(-1).contiguous()
        lv, lt, ltr = f(last_vals), f(last_terminateds), f(last_truncateds)
        assert lv.numel() == N and lt.numel() == N and ltr.numel() == N
        if getattr(self, "adv_buf", None) is None:   # allocated once: HIP-graph consumers keep their addresses
            self.adv_buf = torch.empty_like(self.rew_buf)
            self.ret_buf = torch.empty_like(self.rew_buf)
        adv, ret = self.adv_buf, self.ret_buf
        stream = torch.cuda.current_stream(self.device).cuda_stream
        check(lib.pc_gae(self.device.index if self.device.index is not None else torch.cuda.current_device(),
                         self.rew_buf.data_ptr(), self.val_buf.data_ptr(), self.term_buf.data_ptr(), self.trunc_buf.data_ptr(),
                         lv.data_ptr(), lt.data_ptr(), ltr.data_ptr(), float(self.gamma), float(self.gae_lambda), T, N,
                         adv.data_ptr(), ret.data_ptr(), stream), "pc_gae")
        return adv, ret

    def get(self):
        """buffer.py:66-73."""
        assert self.ptr == self.capacity
        self.ptr = 0
        return self.obs_buf, self.act_buf, self.val_buf, self.logprob_buf
