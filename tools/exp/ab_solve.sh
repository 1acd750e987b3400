for i in 1 2; do for v in base default; do
  for w in "eucm 10000" "eucm 1000" "mei 10000"; do
    if [ $v = default ]; then r=$(python tools/prof_solve.py $w 2>/dev/null | tail -1); else r=$(AB_LIB=visgeom_amd/lib/variants/libvisgeom_amd_$v.so python tools/prof_solve.py $w 2>/dev/null | tail -1); fi
    echo "$v $w: $r"
  done
  if [ $v = default ]; then python tools/bench_configs.py 20 --config 3 2>/dev/null | grep "^| config 3" | sed "s/.*| \([0-9.]* ([0-9]*, [0-9.]*)\) |.*/$v stereo solve \1/"; else AB_LIB=visgeom_amd/lib/variants/libvisgeom_amd_$v.so python tools/bench_configs.py 20 --config 3 2>/dev/null | grep "^| config 3" | sed "s/.*| \([0-9.]* ([0-9]*, [0-9.]*)\) |.*/$v stereo solve \1/"; fi
done; done
