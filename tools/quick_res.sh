#!/bin/bash
# Developer tool: a QUICK build (-DPC_DEV_MIN=<mask>: only the benchmarked kernels, see ppocar.hip) with the register / scratch / spill
# table of the rollout kernels and, optionally, the kernel's assembly.   usage: tools/quick_res.sh <mask | full> [extra hipcc flags]
# (full: the product's whole kernel menu, ~100 s: run it after touching env_step_fast / env_step_wave -- a change that is free in one
# instantiation has cost another one 14 spilled registers)
# -> /tmp/pc_quick/{lib.so,res.txt}; with -save-temps the .s lands in /tmp/pc_quick too.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=/tmp/pc_quick; mkdir -p $OUT; cd $OUT
MASK=$1; shift
if [ "$MASK" = "full" ]; then DEVFLAG=""; else DEVFLAG="-DPC_DEV_MIN=$MASK"; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++20 -ffp-contract=off -Wno-unused-function -I$ROOT/include -I$ROOT/ppo-car_amd/csrc \
    $DEVFLAG "$@" -shared -o $OUT/lib.so $ROOT/ppo-car_amd/csrc/ppocar.hip $ROOT/ppo-car_amd/csrc/track_json.cpp \
    -Rpass-analysis=kernel-resource-usage 2> $OUT/res.txt || { grep -E "error|Error" -A5 $OUT/res.txt | head -60; exit 1; }
python3 - $OUT/res.txt <<'PY'
import re, subprocess, sys
t = open(sys.argv[1]).read()
for b in re.split(r"remark: [^\n]*Function Name: ", t)[1:]:
    name = b.split("\n")[0].split()[0]
    if "rollout" not in name: continue
    g = lambda k: (re.search(k + r": (\d+)", b) or [None, "-"])[1]
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r"\(.*", "", dn).replace("void ", "")
    scr, occ, lds = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
    print(f"{dn:52s} VGPR {g('VGPRs'):>4} AGPR {g('AGPRs'):>3} scratch {scr:>5} sgpr-spill {g('SGPRs Spill'):>4} "
          f"vgpr-spill {g('VGPRs Spill'):>4} occ {occ} LDS {lds}")
PY
