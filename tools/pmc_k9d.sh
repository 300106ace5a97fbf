#!/bin/bash
# Developer tool (GPU box): three PMC passes of the float64 persistent rollout kernel (K9d) at the target shape.  Usage: tools/pmc_k9d.sh <tag>
set -o pipefail
TAG=${1:-r5d}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
i=0
for c in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
         "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
         "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc/p$i -- python3 tools/mega_only.py 65536 ${K9D_T:-256} mega f64 ${K9D_FAST:-1} > $OUT/pmc_p$i.log 2>&1 || exit 1
  echo "pmc pass $i done"
done
find $OUT/pmc -name "*agent_info.csv" -delete
