#!/bin/bash
# Memory-side counters of the emit launch where its rate drops: 50 k / 100 k / 200 k images, the tile map of
# contiguous eighths (W = 0) against windows of 8 x 16 tiles (W = 16).  Own rocprofv3 runs with --pmc + --kernel-trace only,
# <= 4 counters per pass; aggregated on the box (tools/pmc_aggregate.py).
# usage: gpurun -- bash tools/pmc_emit_mem.sh <tag>   -> gpurun_out/pmc_emit_mem_<tag>.csv
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
O="$R/gpurun_out/pmc_emit_mem_$TAG"
rm -rf "$O"; mkdir -p "$O"
P1="GRBM_GUI_ACTIVE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum"
P2="TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_64B_sum TCC_TAG_STALL_sum"
P3="TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum"
P4="TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum"
P5="TCC_REQ_sum TCC_WRITE_sum TCC_WRITEBACK_sum TCC_STREAMING_REQ_sum"
for CASE in "50000 0" "50000 16" "100000 0" "100000 16" "200000 0" "200000 16"; do
  set -- $CASE
  for PASS in 1 2 3 4 5; do
    eval C=\$P$PASS
    name="n$1_w$2"
    timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$O/${name}_p$PASS" -o t -- python "$R/tools/emit_pmc_case.py" $1 $2 12 > "$O/${name}_p$PASS.log" 2>&1
    echo "pmc $name pass $PASS rc=$?"
  done
done
# kernel durations under the counter passes (serialised launches): mean per run of the emit kernel
python - "$O" <<'PY' > "$R/gpurun_out/pmc_emit_mem_${TAG}_durations.txt"
import csv, glob, os, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)):
    run = os.path.relpath(f, sys.argv[1]).split(os.sep)[0]
    v = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3 for r in csv.DictReader(open(f)) if "vg_emit_kernel" in r["Kernel_Name"]]
    if v:
        print("%-16s emit launches %2d  mean %.1f us  min %.1f  max %.1f" % (run, len(v), sum(v[2:]) / max(len(v) - 2, 1), min(v), max(v)))
PY
python "$R/tools/pmc_aggregate.py" "$O" "$R/gpurun_out/pmc_emit_mem_$TAG.csv"
echo "aggregate rc=$?"
find "$O" -name '*.csv' -size +1M -delete
du -sh "$O"
