#!/bin/bash
# Developer tool (GPU box): the rocprofv3 evidence behind profiles/ -- kernel-trace stats of the benchmark command and the
# PMC passes of the persistent rollout kernel (separate passes, --kernel-trace + --pmc only).  Usage: tools/collect_profiles.sh <tag>
# Output: gpurun_out/<tag>/...; summarise with tools/pmc_summary.py and copy what is cited into profiles/.
set -o pipefail
TAG=${1:-r3}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
for w in target cfg1 cfg2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$w -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench_under_prof_$w.json 2> $OUT/prof_$w.err || exit 1
  f=$(ls $OUT/prof_$w/*/*kernel_stats.csv | head -1); cp $f $OUT/${w}_kernel_stats.csv
  rm -rf $OUT/prof_$w
  echo "stats $w done"
done
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
         "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
         "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc/p$i -- python3 tools/mega_only.py 65536 1024 > $OUT/pmc_p$i.log 2>&1 || exit 1
  echo "pmc pass $i done"
done
# keep only the small CSVs
find $OUT/pmc -name "*agent_info.csv" -delete
du -sh $OUT
