#!/bin/bash
# Developer tool (GPU box): clock / power / temperature of the card sampled once a second while tools/mega_only.py keeps the
# persistent rollout kernel running back to back (the sustained-clock statement of DESIGN.md section 8).  Usage: tools/clock_power_probe.sh [seconds]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
SECS=${1:-20}
python3 - $SECS <<'PY' &
import sys, time, os
sys.path.insert(0, os.getcwd())
import torch
from ppo_car_amd.ppo import PPOConfig, Trainer
tr = Trainer(PPOConfig(n_envs=65536, n_steps=1024, num_rays=16, track="tracks/big_track.json", rollout_kernel="mega", seed=3), device="cuda")
t_end = time.time() + float(sys.argv[1])
n = 0
while time.time() < t_end:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); tr.run_epoch(sync=False); e1.record(); torch.cuda.synchronize()
    n += 1
    if n % 25 == 0:
        print(f"epoch {n}: {e0.elapsed_time(e1):.2f} ms", flush=True)
PY
PID=$!
sleep 6
for i in $(seq 1 $((SECS - 8))); do
  /opt/rocm/bin/rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (edge|junction|hotspot)" | tr -s ' ' | tr '\n' ';'
  echo
  sleep 1
done
wait $PID
