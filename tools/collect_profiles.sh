#!/bin/bash
# gpurun_out/ of one measurement session (tools/gpu_check.sh <tag>; tools/prof_configs.sh <tag>; tools/bench_local.py;
# tools/exp/prof_solve.sh rig / eucm) -> the tracked evidence under profiles/<tag>_*.   usage: bash tools/collect_profiles.sh <tag>
set -e
T=${1:?tag}
R=$(cd "$(dirname "$0")/.." && pwd); cd "$R"
python tools/prof_summary.py $T > /dev/null
python tools/pmc_sq_summary.py $T > /dev/null
python tools/prof_configs_summary.py $T > /dev/null
cp gpurun_out/bench_$T.json profiles/${T}_bench_n1.json
cp gpurun_out/bench_100k_$T.json profiles/${T}_bench_100k.json
cp gpurun_out/box_$T.txt profiles/${T}_box.txt
grep -E "passed|failed|Gram blocks" gpurun_out/pytest_gpu_$T.log | tail -3 > profiles/${T}_pytest_gpu_tail.txt
cp gpurun_out/gram_kernel_by_context_$T.txt profiles/${T}_gram_kernel_by_context.txt
cp gpurun_out/pmc_sq_$T.csv profiles/${T}_pmc_sq_raw.csv
for f in calib_kernel_stats_$T.csv calib_e2e_$T.json calib_e2e_$T.md; do
  [ -f gpurun_out/$f ] && cp gpurun_out/$f profiles/${T}_${f/_$T/}
done
if [ -f gpurun_out/bench_local_$T.txt ]; then
  { echo "# tools/bench_local.py, round tag \`$T\` (one MI355X, HIP-event timings)"; echo; echo '```'; grep -v amdgpu.ids gpurun_out/bench_local_$T.txt; echo '```'; } > profiles/${T}_local.md
fi
if [ -f gpurun_out/prof_solve_rig_$T.txt ]; then
  { echo "# LM loops per kernel, rocprofv3 --kernel-trace --stats, tag $T (tools/exp/prof_solve.sh)"; echo
    echo "## config 5 (rig, 5000 frames), host-driven loop"; echo '```'; grep -v "^[EW]2026\|amdgpu.ids" gpurun_out/prof_solve_rig_$T.txt | cut -c1-140; echo '```'
    if [ -f gpurun_out/prof_solve_eucm_$T.txt ]; then echo; echo "## EUCM mono, 10000 images, device-resident loop"; echo '```'; grep -v "^[EW]2026\|amdgpu.ids" gpurun_out/prof_solve_eucm_$T.txt | cut -c1-140; echo '```'; fi
    if [ -f gpurun_out/prof_solve_stereo_$T.txt ]; then echo; echo "## config 3 (stereo, 2000 pairs), device-resident loop"; echo '```'; grep -v "^[EW]2026\|amdgpu.ids" gpurun_out/prof_solve_stereo_$T.txt | cut -c1-140; echo '```'; fi
  } > profiles/${T}_lm_loops.md
fi
cat profiles/${T}_pytest_gpu_tail.txt
ls profiles | grep "^${T}_"
