# same-box A/B of vg_problem_solve between the default library and variants (python -m visgeom_amd._build --variant NAME -DFLAG)
VARIANTS=${VARIANTS:-"default base"}
for i in 1 2; do for v in $VARIANTS; do
  for w in "eucm 10000" "eucm 100000" "ucm 10000"; do
    if [ $v = default ]; then r=$(python tools/exp/solve_probe.py $w 4 2>/dev/null | tail -1); else r=$(AB_LIB=visgeom_amd/lib/variants/libvisgeom_amd_$v.so python tools/exp/solve_probe.py $w 4 2>/dev/null | tail -1); fi
    echo "$v: $r"
  done
done; done
