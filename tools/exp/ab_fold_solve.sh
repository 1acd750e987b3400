#!/bin/bash
# same-box A/B of the reduced solve inside vg_backsub_solve_kernel: VG_SOLVER_ONE_WAVE_FOLD=1 = the first wave alone (before round 4)
for k in 1 2; do for h in 0 1; do
  for w in "stereo 2000" "eucm 10000" "eucm 1000" "mei 10000" "ucm 10000" "mei 1000"; do
    echo "one_wave=$h $w: $(VG_SOLVER_ONE_WAVE_FOLD=$h REPS=5 python tools/prof_solve.py $w | sort -t' ' -k6 -g | awk '{print $2, $3, $5, $6}' | sort -k3 -g | head -1)"
  done
done; done
