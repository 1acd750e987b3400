#!/bin/bash
# SQ counters (issue / stall split) of the FP64 Gram kernels, the emit kernels and the f5 streaming kernels.
# Own rocprofv3 runs with --pmc + --kernel-trace only, two passes of <= 8 SQ counters each; the per-dispatch CSVs are
# AGGREGATED ON THE BOX (tools/pmc_aggregate.py: mean per kernel and counter) because they exceed the 8 MiB that
# travels back -- the round-3 tables were empty for exactly that reason.
# usage: gpurun -- bash tools/pmc_sq.sh <tag>        -> gpurun_out/pmc_sq_<tag>.csv ; tools/pmc_sq_summary.py <tag>
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
O="$R/gpurun_out/pmc_sq_$TAG"
rm -rf "$O"; mkdir -p "$O"
PASS_A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
PASS_B="GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR"
run() {  # name, pass letter, counters, command...
  local name=$1 pass=$2 counters=$3; shift 3
  timeout 600 rocprofv3 --pmc $counters --kernel-trace --output-format csv -d "$O/${name}_$pass" -o t -- "$@" > "$O/${name}_$pass.log" 2>&1
  echo "pmc $name pass $pass rc=$?"
}
for PASS in A B; do
  C=$PASS_A; [ $PASS = B ] && C=$PASS_B
  run headline $PASS "$C" python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-secondary-configs --headline-kernels-only
  run config3 $PASS "$C" python "$R/tools/bench_configs.py" 20 --config 3
  run config5 $PASS "$C" python "$R/tools/bench_configs.py" 20 --config 5
  run local $PASS "$C" python "$R/tools/bench_local.py" 20
  run calib $PASS "$C" python "$R/tools/bench_calib.py" --only mono_eucm_10k --no-cli --runs 1   # vg_pose_lm_kernel (f2) inside the front end
done
python "$R/tools/pmc_aggregate.py" "$O" "$R/gpurun_out/pmc_sq_$TAG.csv"
echo "aggregate rc=$?"
find "$O" -name '*.csv' -size +1M -delete
du -sh "$O"
