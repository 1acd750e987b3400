# same-box A/B of an LM-loop switch: HOOK=VG_SOLVER_LDS_SOLVE bash tools/exp/ab_hook.sh   (hook = 1: the old route)
HOOK=${HOOK:-VG_SOLVER_LDS_SOLVE}
for i in 1 2; do for h in 1 0; do
  for w in "eucm 10000" "eucm 1000" "mei 10000" "ucm 10000"; do
    r=$(env $HOOK=$h python tools/prof_solve.py $w 2>/dev/null | tail -1)
    echo "$HOOK=$h $w: $r"
  done
done; done
