# counters of the LM loop's kernels at 100 k images (tools/prof_solve.py <model> 100000): SQ issue / wait split, memory traffic
R=${GRAFT_REPO_ROOT:-$(pwd)}; M=${1:-eucm}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/pmc_solve_100k_$M; rm -rf $O; mkdir -p $O
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
P2="GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM"
P3="FETCH_SIZE"
P4="WRITE_SIZE"
P5="TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"
for i in 1 2 3 4 5; do eval C=\$P$i
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/p$i -o t -- python $R/tools/prof_solve.py $M 100000 > $O/p$i.log 2>&1; echo "pass $i rc=$?"
done
python $R/tools/pmc_aggregate.py $O $R/gpurun_out/pmc_solve_100k_$M.csv
find $O -name '*.csv' -size +1M -delete
