cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
W3=visgeom_amd/lib/variants/libvisgeom_amd_waves3.so
{
for rep in 1 2; do
  timeout 200 python tools/exp/gram_probe.py eucm 10000
  AB_LIB=$W3 timeout 200 python tools/exp/gram_probe.py eucm 10000
done
timeout 200 python tools/exp/gram_probe.py eucm 100000
AB_LIB=$W3 timeout 200 python tools/exp/gram_probe.py eucm 100000
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r05k_gram_waves3_ab.txt
cat gpurun_out/r05k_gram_waves3_ab.txt
AB_LIB=visgeom_amd/lib/variants/libvisgeom_amd_stamps.so timeout 300 python tools/exp/gram_stamps_probe.py eucm 10000 2>&1 | grep -v amdgpu.ids > gpurun_out/r05k_gram_stamps_eucm_10k.txt
cat gpurun_out/r05k_gram_stamps_eucm_10k.txt
