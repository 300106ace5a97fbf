"""Developer tool: where the small form (K9s, PC_OPT_ROLLOUT_FORM 1) stops paying against the big form (K9, form 0): us per vector step
of the persistent rollout for a ladder of batch sizes, both forms (and both small-form workgroup sizes), same box, inside a short run.
usage: python tools/form_sweep.py [f32|f64] [T]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ppo_car_amd.ppo import PPOConfig, Trainer
dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 256
for N in (2048, 4096, 6144, 8192, 12288, 16384, 24576, 32768):
    row = []
    for form, epw in ((1, 16), (1, 32), (4, 0), (0, 128), (0, 256)):      # (4: 16 envs per wave, 128 per workgroup)
        try:
            tr = Trainer(PPOConfig(n_envs=N, n_steps=T, num_rays=16, track=f"{ROOT}/tracks/big_track.json", rollout_kernel="mega", seed=3, env_dtype=dtype,
                                   rollout_form=form, rollout_epw=epw, use_graphs=False), device="cuda")
            for _ in range(3):
                tr.run_epoch(sync=False)
            ts = []
            for _ in range(6):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); tr.rollout(); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3 / T)
                tr.buffer.ptr = 0
            row.append(f"form {form} epw {epw:3d} ({tr.envs.last_rollout_kernel() if tr.rollout_mode == 'mega' else tr.rollout_mode}): {min(ts):6.2f}")
            tr.close()
        except Exception as ex:
            row.append(f"form {form} epw {epw}: {type(ex).__name__}")
    print(f"N {N:6d} {dtype} | " + " | ".join(row), flush=True)
