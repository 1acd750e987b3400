#!/bin/bash
# One GPU-box session: smoke, GPU parity tests, bench, rocprofv3 kernel trace + PMC passes.
# Outputs -> gpurun_out/.   usage: gpurun -- bash tools/gpu_check.sh [tag] [skip-tests]
TAG=${1:-r01}
SKIP=${2:-}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R" || exit 1
mkdir -p gpurun_out
{ rocm-smi --showproductname 2>/dev/null | grep -E "Card|GPU" | head -4; echo "nproc $(nproc)"; grep -m1 'model name' /proc/cpuinfo; } > gpurun_out/box_$TAG.txt
if [ -z "$SKIP" ]; then
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1
echo "smoke rc=$?" | tee -a gpurun_out/smoke_$TAG.log
timeout 1500 python -m pytest tests -m gpu -x -q -s > gpurun_out/pytest_gpu_$TAG.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu_$TAG.log
tail -4 gpurun_out/pytest_gpu_$TAG.log
fi
timeout 600 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
echo "bench rc=$?"; cat gpurun_out/bench_$TAG.json; tail -3 gpurun_out/bench_$TAG.err
timeout 600 python bench.py --images 100000 --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/bench_100k_$TAG.json 2> gpurun_out/bench_100k_$TAG.err
echo "bench100k rc=$?"; cat gpurun_out/bench_100k_$TAG.json
cd /tmp && export TMPDIR=/tmp
P="$R/gpurun_out/prof_$TAG"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$P/trace" -o t -- python "$R/bench.py" --steps 100 --warmup 10 --no-cpu-baseline --headline-kernels-only > "$P.trace.log" 2>&1
echo "rocprof trace rc=$?"
# the product entry point under the kernel trace: `calib` on the headline-size JSON -- vg_pose_lm_kernel, the solve's
# kernels and the projection launch of the residual report in one table
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$P/trace_calib" -o t -- python "$R/tools/bench_calib.py" --only mono_eucm_10k --no-cli --runs 3 > "$P.trace_calib.log" 2>&1
echo "rocprof calib trace rc=$?"
f=$(find "$P/trace_calib" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$R/gpurun_out/calib_kernel_stats_$TAG.csv" && head -12 "$f"
find "$P/trace_calib" -name '*kernel_trace.csv' -delete
timeout 900 python "$R/tools/bench_calib.py" --out "$R/gpurun_out/calib_e2e_$TAG.json" --md "$R/gpurun_out/calib_e2e_$TAG.md" --tag $TAG > "$R/gpurun_out/calib_e2e_$TAG.log" 2>&1
echo "bench_calib rc=$?"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$P/pmc_$C" -o t -- python "$R/bench.py" --steps 20 --warmup 2 --no-cpu-baseline --headline-kernels-only > "$P.pmc_$C.log" 2>&1
  echo "rocprof pmc $C rc=$?"
done
# SQ counters: tools/pmc_sq.sh (own passes, aggregated on the box -- the per-dispatch CSV of an 8-counter pass is larger
# than what the size filter below lets travel, which is how the SQ tables of r02g..r03f came back empty)
bash "$R/tools/pmc_sq.sh" "$TAG"
cd /tmp
find "$P" -type f | head -40
f=$(find "$P/trace" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -8 "$f"
k=$(find "$P/trace" -name '*kernel_trace.csv' | head -1)
[ -n "$k" ] && python "$R/tools/kernel_context.py" "$k" "gram_valu_kernel<0, 1, true, 3>" > "$R/gpurun_out/gram_kernel_by_context_$TAG.txt" 2>&1
# means per (pass, kernel, counter), computed HERE: whatever the size filter below removes, the summary can still be made
python "$R/tools/pmc_aggregate.py" "$P" "$R/gpurun_out/pmc_traffic_$TAG.csv"
find "$P" -name '*.csv' -size +8M -delete
