#!/bin/bash
# ONE measurement session on a GPU box -> everything tools/collect_profiles.sh <tag> turns into the tracked evidence under
# profiles/<tag>_*: smoke + GPU tests + bench.py (10 k and 100 k images) + kernel trace + FETCH / WRITE PMC passes + SQ passes
# (tools/gpu_check.sh), every BASELINE config with its own trace and traffic passes (tools/prof_configs.sh), the f5 bench, the LM
# loops per kernel (tools/exp/prof_solve.sh), the product entry point end to end (tools/bench_calib.py, inside gpu_check.sh).
#   usage: gpurun --timeout 3300 -- bash tools/full_profile.sh <tag>      then here: bash tools/collect_profiles.sh <tag>
TAG=${1:?tag}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R" || exit 1
bash tools/gpu_check.sh $TAG
cd "$R"; bash tools/prof_configs.sh $TAG
cd "$R"; timeout 600 python tools/bench_local.py > gpurun_out/bench_local_$TAG.txt 2>&1; echo "bench_local rc=$?"
for WN in rig:5000 eucm:10000 stereo:2000; do
  W=${WN%%:*}; N=${WN##*:}
  cd "$R"; timeout 600 bash tools/exp/prof_solve.sh $W $N > gpurun_out/prof_solve_${W}_$TAG.txt 2>&1; echo "prof_solve $W rc=$?"
done
du -sh "$R/gpurun_out"
