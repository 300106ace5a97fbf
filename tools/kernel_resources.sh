#!/bin/bash
# Developer tool: VGPR / scratch / spill table of the rollout and env-step kernels (hipcc -Rpass-analysis=kernel-resource-usage).
# usage: tools/kernel_resources.sh [name-filter-regex]   (compiles the product TU once more into /tmp; ~2 min)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/pc_res
mkdir -p "$OUT"
cd "$ROOT/ppo-car_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++20 -ffp-contract=off -I../../include -I. -shared -o "$OUT/lib_res.so" \
    ppocar.hip track_json.cpp -Rpass-analysis=kernel-resource-usage 2> "$OUT/res.txt" || { tail -30 "$OUT/res.txt"; exit 1; }
python3 - "$OUT/res.txt" "${1:-rollout|env_step}" <<'PY'
import re, subprocess, sys
t = open(sys.argv[1]).read()
flt = re.compile(sys.argv[2])
for b in re.split(r"remark: [^\n]*Function Name: ", t)[1:]:
    name = b.split("\n")[0].split()[0]
    g = lambda k: (re.search(k + r": (\d+)", b) or [None, "-"])[1]
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r"\(.*", "", dn).replace("void ", "")
    if flt.search(dn):
        scr, occ = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")
        print(f"{dn:62s} VGPR {g('VGPRs'):>4} AGPR {g('AGPRs'):>3} scratch {scr:>5} sgpr-spill {g('SGPRs Spill'):>4} "
              f"vgpr-spill {g('VGPRs Spill'):>4} occ {occ}")
PY
