#!/bin/bash
# Developer tool: timing ablations of the persistent rollout kernel.  Builds SEPARATE copies of the package around libraries
# compiled with -DPC_ABLATE=n (1: no policy MFMA pass, 2: no env step, 3: neither) under build/ablate_<n>_pkg/ -- the product
# library is untouched and refuses to load such a build; the copies' _capi.py has that check removed -- and times pc_rollout.
#   tools/ablate_run.sh build      (here, hipcc)        tools/ablate_run.sh run [n_envs] [n_steps]   (on the GPU box)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
if [ "$1" = "build" ]; then
  for n in 1 2 3; do
    d=build/ablate_${n}_pkg
    mkdir -p $d/ppo-car_amd $d/ppo_car_amd
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -Wno-unused-function -Iinclude -Ippo-car_amd/csrc -DPC_ABLATE=$n -shared \
        -o $d/ppo-car_amd/libppocar.so ppo-car_amd/csrc/ppocar.hip ppo-car_amd/csrc/track_json.cpp || exit 1
    cp ppo-car_amd/*.py $d/ppo-car_amd/ && cp ppo_car_amd/__init__.py $d/ppo_car_amd/
    python3 - "$d/ppo-car_amd/_capi.py" <<'PY'
import sys
p = sys.argv[1]
s = open(p).read()
s = s.replace("if lib.pc_build_ablate() != 0:", "if False:")
open(p, "w").write(s)
PY
  done
  exit 0
fi
N=${2:-65536}; T=${3:-512}
for n in 0 1 2 3; do
  if [ $n = 0 ]; then P=$ROOT; else P=$ROOT/build/ablate_${n}_pkg; fi
  python3 - $P $N $T $n <<'PY'
import sys, os
pkg, N, T, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
sys.path.insert(0, pkg)
import torch
from ppo_car_amd.ppo import PPOConfig, Trainer
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
tr = Trainer(PPOConfig(n_envs=N, n_steps=T, num_rays=16, track=f"{root}/tracks/big_track.json", rollout_kernel="mega", use_graphs=False), device="cuda")
for _ in range(3):
    tr.rollout(); tr.buffer.ptr = 0
torch.cuda.synchronize()
ts = []
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); tr.rollout(); e1.record(); torch.cuda.synchronize(); tr.buffer.ptr = 0
    ts.append(e0.elapsed_time(e1) * 1e3 / T)
print(f"ablate {n} ({['full', 'no policy MFMA pass', 'no env step', 'neither'][int(n)]}): {min(ts):.2f} us per vector step of {N} envs", flush=True)
PY
done
