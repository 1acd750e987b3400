#!/bin/bash
# SQ counters of the one-shot and the persistent direct Gram kernel on the same box: two passes over tools/exp/gram_pers_probe.py
# usage: gpurun -- bash tools/exp/pmc_gram_pers.sh <tag> [images]
TAG=${1:-x}; N=${2:-100000}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
O="$R/gpurun_out/pmc_gram_pers_$TAG"; rm -rf "$O"; mkdir -p "$O"
PASS_A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
PASS_B="GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR"
for PASS in A B; do
  C=$PASS_A; [ $PASS = B ] && C=$PASS_B
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$O/probe_$PASS" -o t -- python "$R/tools/exp/gram_pers_probe.py" eucm $N > "$O/probe_$PASS.log" 2>&1
  echo "pass $PASS rc=$?"
done
python "$R/tools/pmc_aggregate.py" "$O" "$R/gpurun_out/pmc_gram_pers_$TAG.csv"
find "$O" -name '*.csv' -size +1M -delete
grep "gram_valu" "$R/gpurun_out/pmc_gram_pers_$TAG.csv" | cut -d, -f2- | sed 's/vg::vg_gram_valu_//'
