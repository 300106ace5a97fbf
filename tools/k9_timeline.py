"""Developer tool: phase timeline of the big-form persistent rollout kernel (K9) from the -DPC_STAMPS build.

    make -C ppo-car_amd/csrc stamps         # build/stamps_pkg/: a copy of the package around libppocar.so built with -DPC_STAMPS
    python tools/k9_timeline.py [n_envs] [n_steps] [rollout_form]

Workgroup 0's eight waves stamp s_memtime at the phase boundaries of steps 64..71.  Waves w and w + 4 share a SIMD.
Phases: 0 step start, 1 operand split done / policy pass starts, 2 policy pass done, 3 draw + action stores done,
4 env pre-sweep done (action decode, physics, directions, gate casts), 5 sweep done (4 and 5: small form only -- the big
form has no registers to spare for stamps inside its env step), 6 refinement + bookkeeping + LDS row writes done,
7 reset fix-up + copy-out + flag stores done.
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "build", "stamps_pkg")
sys.path.insert(0, PKG)
import torch  # noqa: E402
import ppo_car_amd  # noqa: E402,F401  (the stamps copy)
from ppo_car_amd import _capi  # noqa: E402
from ppo_car_amd.ppo import PPOConfig, Trainer  # noqa: E402

assert _capi.lib_path().startswith(PKG), _capi.lib_path()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T = int(sys.argv[2]) if len(sys.argv) > 2 else 128
FORM = int(sys.argv[3]) if len(sys.argv) > 3 else -1      # PC_OPT_ROLLOUT_FORM: 0 = 32 envs per wave at any size, 4 = 16 envs per wave
tr = Trainer(PPOConfig(n_envs=N, n_steps=T, num_rays=16, track=f"{ROOT}/tracks/big_track.json", rollout_kernel="mega", use_graphs=False, rollout_form=FORM), device="cuda")
for _ in range(3):
    tr.rollout(); tr.buffer.ptr = 0
torch.cuda.synchronize()
NT, NPH = 8, 12
buf = (C.c_ulonglong * (8 * NT * NPH))()
_capi.lib.pc_debug_read_stamps.restype = C.c_int
n = _capi.lib.pc_debug_read_stamps(buf, len(buf))
assert n == len(buf), n
import numpy as np
st = np.array(buf, dtype=np.uint64).reshape(8, NT, NPH).astype(np.int64)
t0 = st.min()
names = (["split", "policy", "draw", "env-pre", "sweep", "post", "copy-out"] if tr.cfg.n_envs > 16384 else
         ["tiles", "barrier1", "draw+bar2", "env-pre", "sweep part", "xchg+post", "barrier3"])
big = tr.cfg.n_envs > 16384
if not big:              # the small form's env step carries three more stamps: 8 refinement done, 9 rare-path loop done, 10 verdicts met
    st = st[:, :, [0, 1, 2, 3, 4, 5, 8, 9, 10, 6, 7]]
    names = (["tiles", "barrier1", "draw", "env-pre", "sweep+min", "refine", "rare loop", "verdicts", "bookkeeping", "barrier2"] if N <= 4096 else   # a wave owns two envs
             ["tiles", "barrier1", "draw+bar2", "env-pre", "sweep+xchg", "refine", "rare loop", "verdict bar", "bookkeeping", "barrier3"])
if big:                  # the big form carries no stamps inside the env step (they made it spill): phases 3..6 are one
    st = st[:, :, [0, 1, 2, 3, 6, 7]]
    names = ["split", "policy", "draw", "env step", "copy-out"]
print(f"rollout mode {tr.rollout_mode} ({tr.envs.last_rollout_kernel()}); cycles (s_memtime ticks) per phase, mean over {NT} steps")
print("wave " + " ".join(f"{n:>9s}" for n in names) + "   step total")
for w in range(8):
    d = np.diff(st[w], axis=1).mean(0)
    step = np.diff(st[w, :, 0]).mean()
    print(f"{w:4d} " + " ".join(f"{x:9.0f}" for x in d) + f"   {step:9.0f}")
print("\ntimeline of waves 0 and 4 (same SIMD), step 66: phase start offsets")
for w in (0, 4):
    print(w, [int(x - st[0, 2, 0]) for x in st[w, 2]], "next step starts at", int(st[w, 3, 0] - st[0, 2, 0]))

if os.environ.get("K9_TIMELINE_PER_STEP"):      # one phase, every stamped step: rows = waves, columns = steps
    ph = names.index(os.environ["K9_TIMELINE_PER_STEP"])
    print(f"\nphase '{names[ph]}' per step")
    for w in range(8):
        print(w, [int(x) for x in (st[w, :, ph + 1] - st[w, :, ph])])
