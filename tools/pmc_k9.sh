#!/bin/bash
# Developer tool (GPU box): the five PMC passes of the persistent rollout kernel only (see collect_profiles.sh).  Usage: tools/pmc_k9.sh <tag>
set -o pipefail
TAG=${1:-r3}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
         "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
         "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc/p$i -- python3 tools/mega_only.py 65536 1024 > $OUT/pmc_p$i.log 2>&1 || exit 1
  echo "pmc pass $i done"
done
find $OUT/pmc -name "*agent_info.csv" -delete
python3 tools/pmc_summary.py $OUT/pmc ${TAG}_rollout_kernel_pmc_N65536_T1024 65536 16 f32 rollout_kernel 1024 > $OUT/pmc_summary.json
