#!/bin/bash
# Developer tool (GPU box): the round's closing evidence in one call -- the full -m gpu suite, the default bench line, rocprofv3 --stats of
# the bench command in the float64 dtype, the PMC passes of both float64 persistent kernels, the determinism check.  Output under gpurun_out/ (copy what is cited into profiles/).
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
mkdir -p gpurun_out/r5
python -m pytest tests -m gpu -q > gpurun_out/final_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/final_pytest.log | cut -c1-200
(time python bench.py > gpurun_out/r5_bench_target.json 2> gpurun_out/r5_bench_target.err) 2>&1 | grep real
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5_bench_target.json"))
e = d["exact_f64_value"]
print(round(d["value"] / 1e6), round(d["ms_per_step"], 2), round(d["roofline"]["frac"], 3), "| f64", round(e["value"] / 1e6), round(e["ms_per_step"], 2),
      round(e["roofline"]["launch_us"]), round(e["roofline"]["frac"], 3), round(e["roofline"]["hbm_frac"], 3), "|", {k: (round(v["value"] / 1e6), round(v["exact_f64"]["value"] / 1e6), v["exact_f64"]["kernel"]) for k, v in d["other_workloads"].items()})
PY
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5/prof_f64 -- python3 bench.py --env-dtype f64 --steps 3 --warmup 1 --no-cpu-baseline --no-extras \
    > gpurun_out/r5/bench_under_prof_f64.json 2> gpurun_out/r5/prof_f64.err
cp $(ls gpurun_out/r5/prof_f64/*/*kernel_stats.csv | head -1) gpurun_out/r5/f64_kernel_stats.csv; rm -rf gpurun_out/r5/prof_f64
head -4 gpurun_out/r5/f64_kernel_stats.csv | cut -c1-140
tools/pmc_k9d.sh r5d 2>&1 | tail -1          # the default dispatch of an F64 handle: K9 in its literal form (rollout_kernel<..., true>)
python tools/pmc_summary.py gpurun_out/r5d/pmc r5_k9lit_raw 65536 16 f64 rollout_kernel 256 > gpurun_out/r5d/summary.json 2>/dev/null
K9D_FAST=0 tools/pmc_k9d.sh r5d_filter 2>&1 | tail -1   # the filter form (PC_OPT_ROLLOUT_FAST = 0): rollout_f64_kernel
python tools/pmc_summary.py gpurun_out/r5d_filter/pmc r5_k9d_raw 65536 16 f64 rollout_f64_kernel 256 > gpurun_out/r5d_filter/summary.json 2>/dev/null
python tools/determinism_check.py 4 f64 2>&1 | grep -v amdgpu
