"""Developer tool: the persistent rollout kernel with and without the 1/den table in LDS (PC_OPT_ROLLOUT_FORM 0 / 2), same box, same
policy: what the table's LDS traffic (scattered rows: bank conflicts) costs against forming den and its reciprocal per candidate."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ppo_car_amd.ppo import PPOConfig, Trainer
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
tr = Trainer(PPOConfig(n_envs=N, n_steps=T, num_rays=16, track=f"{ROOT}/tracks/big_track.json", rollout_kernel="mega", seed=3), device="cuda")
for _ in range(4):
    tr.run_epoch(sync=False)
torch.cuda.synchronize()
for rnd in range(3):
    for form in (0, 2):
        tr.envs.set_option("rollout_form", form)
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); tr.rollout(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / T)
            tr.buffer.ptr = 0
        print(f"form {form}: {min(ts):.3f} us per vector step (min of 3)", flush=True)
