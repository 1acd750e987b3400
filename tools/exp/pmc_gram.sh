#!/bin/bash
# SQ counters of the fused Gram kernel ALONE at one size (own runs, --pmc + --kernel-trace only), aggregated on the box:
# issue vs stall split, and GRBM_GUI_ACTIVE for the clock the launch really ran at.
# usage: gpurun -- bash tools/exp/pmc_gram.sh [model] [images] [tag]
R=${GRAFT_REPO_ROOT:-$(pwd)}; M=${1:-eucm}; N=${2:-10000}; TAG=${3:-r04}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/pmc_gram_${M}_${N}_$TAG; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/gram_A -o t -- python $R/tools/exp/gram_probe.py $M $N > $O/a.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --kernel-trace --output-format csv -d $O/gram_B -o t -- python $R/tools/exp/gram_probe.py $M $N > $O/b.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/gram_T -o t -- python $R/tools/exp/gram_probe.py $M $N > $O/t.log 2>&1
python $R/tools/pmc_aggregate.py $O $R/gpurun_out/pmc_gram_${M}_${N}_$TAG.csv
cp $O/gram_T/t_kernel_stats.csv $R/gpurun_out/pmc_gram_${M}_${N}_${TAG}_kernel_stats.csv 2>/dev/null
find $O -name '*.csv' -size +1M -delete
