"""Developer tool: phase stamps of the PPO minibatch kernel K10 (ppo_fwdbwd_kernel), workgroup 0, from the -DPC_STAMPS build
(make -C ppo-car_amd/csrc stamps).  Phases: 0 entry, 1 first-level loads issued, 2 W1 rows in registers (+ statistics), 3 layer 1
done, 4 layer-2 GEMV done, 5 loss done, 6 backward done, 7 partial-gradient stores issued (the kernel ends when they have drained:
compare with the kernel's duration in the rocprofv3 table)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "build", "stamps_pkg")
sys.path.insert(0, PKG)
import torch  # noqa: E402
from ppo_car_amd import _capi  # noqa: E402
from ppo_car_amd.ppo import PPOConfig, Trainer  # noqa: E402

tr = Trainer(PPOConfig(n_envs=4096, n_steps=1024, num_rays=16, track=f"{ROOT}/tracks/big_track.json", use_graphs=False), device="cuda")
for _ in range(2):
    tr.run_epoch(sync=False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
tr.rollout()
e0.record(); tr.update(); e1.record(); torch.cuda.synchronize()
buf = (C.c_ulonglong * 16)()
_capi.lib.pc_debug_read_stamps_u.restype = C.c_int
assert _capi.lib.pc_debug_read_stamps_u(buf) == 16
t = [int(buf[i]) for i in range(8)]
names = ["loads issued", "W1 in registers", "layer 1", "layer-2 GEMV", "loss", "backward", "stores issued"]
print("K10 workgroup 0 (s_memtime ticks):")
for i, n in enumerate(names):
    print(f"  {n:18s} {t[i + 1] - t[i]:7d}")
u = [int(buf[i]) for i in range(16)]
print("  inside 'stores issued': small stores %d | barrier %d | tile write + barrier (actor) %d | float4 stores %d | barrier %d | "
      "tile write + barrier (critic) %d | float4 stores %d" % (u[8] - u[6], u[9] - u[8], u[10] - u[9], u[11] - u[10], u[12] - u[11],
                                                              u[13] - u[12], u[7] - u[13]))
print(f"  entry -> stores issued {t[7] - t[0]} ticks; eager update of one epoch (GAE + 80 minibatch steps): {e0.elapsed_time(e1):.2f} ms")
