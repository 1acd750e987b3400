cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
T=r05d
NONT=visgeom_amd/lib/variants/libvisgeom_amd_nont.so
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed\|error" gpurun_out/${T}_pytest.log | tail -5
for rep in 1 2; do
  timeout 600 python tools/bench_configs.py 200 2>/dev/null | tail -6
  AB_LIB=$NONT timeout 600 python tools/bench_configs.py 200 2>/dev/null | tail -6
done > gpurun_out/${T}_bench_configs_ab.txt 2>&1
cat gpurun_out/${T}_bench_configs_ab.txt
timeout 600 python tools/bench_local.py > gpurun_out/${T}_local.txt 2>&1; tail -12 gpurun_out/${T}_local.txt
timeout 1200 python bench.py --steps 20 --warmup 3 > gpurun_out/${T}_bench_n1.json 2> gpurun_out/${T}_bench_n1.err; echo "bench rc=$?"; head -c 2500 gpurun_out/${T}_bench_n1.json; echo; python - <<'PY'
import json
d=json.load(open("gpurun_out/r05d_bench_n1.json"))
print("emit_sweep", [(r["images"], r["route"], round(r["kernel_us"],1), round(r["frac"],3), round(r["frac_whole_step"],3)) for r in d.get("emit_sweep",[])])
print("eucm_100k", d["eucm_100k"]["roofline"]["frac"], d["eucm_100k"]["roofline"]["frac_whole_step"])
print("jtj", d["jtj"])
print("pcie", d["pcie_inclusive"])
PY
