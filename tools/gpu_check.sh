#!/bin/bash
# One GPU-box session: smoke, GPU parity tests, bench, rocprofv3 kernel trace.  Outputs -> gpurun_out/.
# usage: gpurun -- bash tools/gpu_check.sh [tag]
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R" || exit 1
mkdir -p gpurun_out
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/box_$TAG.txt
nproc >> gpurun_out/box_$TAG.txt; grep -m1 'model name' /proc/cpuinfo >> gpurun_out/box_$TAG.txt
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1
echo "smoke rc=$?" | tee -a gpurun_out/smoke_$TAG.log
timeout 1500 python -m pytest tests -m gpu -x -q -s > gpurun_out/pytest_gpu_$TAG.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu_$TAG.log
tail -5 gpurun_out/pytest_gpu_$TAG.log
timeout 600 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
echo "bench rc=$?"; cat gpurun_out/bench_$TAG.json; tail -3 gpurun_out/bench_$TAG.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_$TAG" -o trace -- python "$R/bench.py" --steps 100 --warmup 10 --no-cpu-baseline > "$R/gpurun_out/prof_$TAG.log" 2>&1
echo "rocprof rc=$?"
find "$R/gpurun_out/prof_$TAG" -name '*stats*' | head; 
f=$(find "$R/gpurun_out/prof_$TAG" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -12 "$f"
# keep the merged output small: drop the raw per-dispatch trace if it is huge
find "$R/gpurun_out/prof_$TAG" -name '*kernel_trace.csv' -size +20M -delete
