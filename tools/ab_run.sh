#!/bin/bash
# Developer tool: A/B timing of the persistent rollout kernel between builds of the working tree.
#   [AB_MIN=mask] tools/ab_run.sh build <name> [flags]    (here, hipcc)  -> build/ab_<name>_pkg: the package around a library built from the tree as it is
#                                           now, as a developer QUICK build (-DPC_DEV_MIN=mask, default 1: only the target's rollout kernel; 2: cfg1's, 4: cfg2's;
#                                           AB_MIN=0: every kernel, 75 s) -- the copies' _capi.py has the check that refuses such builds removed
#   [AB_MIXED=1] tools/ab_run.sh run [n_envs] [n_steps] [rays]  (on the GPU box; AB_MIXED: track.json + big_track.json in halves) -> every build/ab_*_pkg in turn, three rounds, inside a short training
#                                           run so that the policy is not the random initial one; the product library is not used.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
if [ "$1" = "build" ]; then
  d=build/ab_${2}_pkg
  mkdir -p $d/ppo-car_amd $d/ppo_car_amd
  MIN=${AB_MIN:-1}; if [ "$MIN" != "0" ]; then MINFLAG="-DPC_DEV_MIN=$MIN"; else MINFLAG=""; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++20 -ffp-contract=off -Wno-unused-function -Iinclude -Ippo-car_amd/csrc $MINFLAG $3 -shared \
      -o $d/ppo-car_amd/libppocar.so ppo-car_amd/csrc/ppocar.hip ppo-car_amd/csrc/track_json.cpp || exit 1
  cp ppo-car_amd/*.py $d/ppo-car_amd/ && cp ppo_car_amd/__init__.py $d/ppo_car_amd/
  sed -i 's/^if lib.pc_build_ablate() != 0:/if False:/' $d/ppo-car_amd/_capi.py
  exit 0
fi
N=${2:-65536}; T=${3:-1024}; R=${4:-16}
for round in $(seq 1 ${AB_ROUNDS:-4}); do
  for P in $ROOT/build/${AB_GLOB:-ab_*_pkg}; do
    python3 - $P $N $T $R <<'PY'
import sys, os
pkg, N, T, R = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
sys.path.insert(0, pkg)
import torch
from ppo_car_amd.ppo import PPOConfig, Trainer
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
track = [f"{root}/tracks/track.json", f"{root}/tracks/big_track.json"] if os.environ.get("AB_MIXED") else f"{root}/tracks/big_track.json"
import json
for extra in json.loads(os.environ.get("AB_CFGS", "[{}]")):      # e.g. AB_CFGS='[{"deferred_adam": true}, {"deferred_adam": false}]': PPOConfig variants of one build
  tag = os.path.basename(pkg) + ("" if not extra else " " + json.dumps(extra))
  tr = Trainer(PPOConfig(n_envs=N, n_steps=T, num_rays=R, track=track, rollout_kernel="mega", seed=3, **extra), device="cuda")
  for _ in range(4):
      tr.run_epoch(sync=False)
  torch.cuda.synchronize()
  ts = []
  for _ in range(int(os.environ.get("AB_REPS", "24"))):
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record(); tr.rollout(); e1.record(); torch.cuda.synchronize()
      ts.append(e0.elapsed_time(e1) * 1e3 / T)
      tr.update()
  ts.sort()
  import time
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(10):
      tr.run_epoch(sync=False)
  torch.cuda.synchronize(); ep = (time.perf_counter() - t0) / 10 * 1e3
  ue = []
  for _ in range(6):      # the update alone (GAE + minibatch loop) between events
      tr.rollout()
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record(); tr.update(); e1.record(); torch.cuda.synchronize()
      ue.append(e0.elapsed_time(e1))
  ue.sort()
  tr.close()
  print(f"{tag:44s} {ts[0]:.3f} (min) {ts[len(ts)//2]:.3f} (median) us per vector step of {N} envs, {R} rays; then 10 epochs back to back: {ep:.2f} ms per epoch = {N * T / ep / 1e3:.0f} M env-steps/s; update {ue[len(ue)//2]:.3f} ms", flush=True)
PY
  done
done
