#!/bin/bash
# Developer tool (GPU box): the round's closing evidence in one call -- the full -m gpu suite, the default bench line, rocprofv3 --stats of the
# bench command (target / cfg1 / cfg2), the PMC passes of the target and configs[1] kernels, the shipped kernels' soak against the per-step kernels, K1f's soak against K1, the step-kernel probe.
# Output under gpurun_out/<tag>/ (copy what is cited into profiles/).   usage: tools/final_evidence.sh <tag> [skip-tests]
TAG=${1:-r6}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
O=gpurun_out/$TAG
mkdir -p $O
if [ "$2" != "skip-tests" ]; then
  python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log | cut -c1-200
fi
(time python bench.py > $O/bench_target.json 2> $O/bench_target.err) 2>&1 | grep real
python - $O <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + "/bench_target.json"))
e = d["exact_f64_value"]
print(round(d["value"] / 1e6), round(d["ms_per_step"], 2), round(d["roofline"]["frac"], 3), "| f64", round(e["value"] / 1e6), "| strict f64+fp32", round(d["strict_f64_fp32_value"].get("value", 0) / 1e6),
      d["strict_f64_fp32_value"].get("kernel"), "|", {k: (round(v["value"] / 1e6), round(v["exact_f64"]["value"] / 1e6), v["exact_f64"]["kernel"]) for k, v in d["other_workloads"].items()},
      "| parity", d["parity_check"]["ok"], d["parity_check"]["rare_branches"]["ok"], "| cpu scaling", round(d["cpu_baseline"]["env_only_scaling"], 2))
PY
export TMPDIR=/tmp
for w in target cfg1 cfg2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_under_rocprof_$w.json 2> $O/prof_$w.err
  cp $(ls $O/prof_$w/*/*kernel_stats.csv | head -1) $O/${w}_kernel_stats.csv; rm -rf $O/prof_$w
  head -3 $O/${w}_kernel_stats.csv | cut -c1-150
done
tools/pmc_k9.sh $TAG/k9 2>&1 | tail -1; cp $O/k9/pmc_summary.json $O/k9_pmc_summary.json
tools/pmc_k9s.sh $TAG/k9s 2>&1 | tail -1
python tools/pmc_summary.py gpurun_out/$TAG/k9s/pmc ${TAG}_k9s_raw 4096 16 f32 rollout_small_kernel 1024 > $O/k9s_pmc_summary.json 2> $O/k9s_pmc_summary.err; tail -2 $O/k9s_pmc_summary.err
S="python tools/soak_rollout.py . --out $O/soak_shipped.jsonl"
$S --launches 110 --rays 16 --n-steps 1024 > /dev/null 2>&1 && $S --launches 400 --rays 16 --n-envs 4096 --n-steps 1024 > /dev/null 2>&1 && $S --launches 600 --rays 32 > /dev/null 2>&1 && \
  $S --launches 150 --rays 16 --n-envs 32768 --n-steps 1024 --mixed > /dev/null 2>&1 && $S --launches 600 --rays 16 --dtype f64 > /dev/null 2>&1
python tools/soak_steps.py --launches 20 --out $O/soak_shipped.jsonl > /dev/null 2>&1 && python tools/soak_steps.py --launches 20 --dtype f64 --out $O/soak_shipped.jsonl > /dev/null 2>&1
python tools/step_forms_probe.py 16 2> /dev/null | grep -v amdgpu.ids > $O/step_forms_probe.txt
cut -c1-260 $O/soak_shipped.jsonl
