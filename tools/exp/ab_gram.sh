# same-box A/B of the fused Gram kernels: base = the library built from the previous commit, default = the working tree,
# further variants by name (python -m visgeom_amd._build --variant NAME -DFLAG)
VARIANTS=${VARIANTS:-"base default"}
for i in 1 2 3; do
for v in $VARIANTS; do
  if [ $v = default ]; then python tools/exp/gram_probe.py eucm 10000 2>&1 | grep gram_fused
  else AB_LIB=visgeom_amd/lib/variants/libvisgeom_amd_$v.so python tools/exp/gram_probe.py eucm 10000 2>&1 | grep gram_fused; fi
done; done
for m in "mei 10000" "ucm 10000" "eucm 100000" "eucm 1000"; do
for v in $VARIANTS; do
  if [ $v = default ]; then python tools/exp/gram_probe.py $m 2>&1 | grep gram_fused
  else AB_LIB=visgeom_amd/lib/variants/libvisgeom_amd_$v.so python tools/exp/gram_probe.py $m 2>&1 | grep gram_fused; fi
done; done
