#!/bin/bash
# rocprofv3 evidence for EVERY BASELINE config: per config, one --kernel-trace --stats run of each kind
# of pass (emit / normal-equation build / full solve) and the FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, counters
# with --kernel-trace only).  Outputs -> gpurun_out/prof_configs_<tag>/; tools/prof_configs_summary.py turns them into the
# tracked files under profiles/.   usage: gpurun -- bash tools/prof_configs.sh <tag> [configs...]
TAG=${1:-r03}
shift
CONFIGS=${@:-2 3 4 5}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R" || exit 1
mkdir -p gpurun_out
timeout 900 python tools/bench_configs.py 200 > gpurun_out/bench_configs_$TAG.txt 2> gpurun_out/bench_configs_$TAG.err
echo "bench_configs rc=$?"; tail -8 gpurun_out/bench_configs_$TAG.txt
cd /tmp && export TMPDIR=/tmp
P="$R/gpurun_out/prof_configs_$TAG"
mkdir -p "$P"
for C in $CONFIGS; do
  for KIND in emit jtj solve; do
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$P/c${C}_${KIND}" -o t -- python "$R/tools/bench_configs.py" 200 --config $C --only $KIND > "$P/c${C}_${KIND}.log" 2>&1
    echo "config $C $KIND trace rc=$?"
  done
  for KIND in emit jtj; do
    for CN in FETCH_SIZE WRITE_SIZE; do
      timeout 300 rocprofv3 --pmc $CN --kernel-trace --output-format csv -d "$P/c${C}_${KIND}_$CN" -o t -- python "$R/tools/bench_configs.py" 30 --config $C --only $KIND > "$P/c${C}_${KIND}_$CN.log" 2>&1
      echo "config $C $KIND pmc $CN rc=$?"
    done
  done
done
# calibration of the two counters on known byte counts (bench.py streams 512 MiB each way with the emit store pattern)
for CN in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $CN --kernel-trace --output-format csv -d "$P/cal_$CN" -o t -- python "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline > "$P/cal_$CN.log" 2>&1
  echo "calibration pmc $CN rc=$?"
done
find "$P" -name '*kernel_trace.csv' -size +6M -delete
find "$P" -name '*.csv' -size +8M -delete
du -sh "$P"
