cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
T=r05q
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed\|error" gpurun_out/${T}_pytest.log | tail -5; grep -B5 -A25 "Error\|FAILED" gpurun_out/${T}_pytest.log | head -80
REF=visgeom_amd/lib/variants/libvisgeom_amd_nowalk.so
for rep in 1 2; do
  echo "== new walk (rep $rep)"; timeout 600 python tools/bench_configs.py 200 2>/dev/null | tail -6
  echo "== no inline walk (rep $rep)"; AB_LIB=$REF timeout 600 python tools/bench_configs.py 200 2>/dev/null | tail -6
done > gpurun_out/${T}_configs_ab.txt 2>&1
cat gpurun_out/${T}_configs_ab.txt
