cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
T=r05f
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed\|error" gpurun_out/${T}_pytest.log | tail -5; tail -30 gpurun_out/${T}_pytest.log | head -60
timeout 1500 python bench.py > gpurun_out/${T}_bench_n1.json 2> gpurun_out/${T}_bench_n1.err; echo "bench rc=$?"; tail -3 gpurun_out/${T}_bench_n1.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05f_bench_n1.json"))
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"])
print("pcie", json.dumps(d["pcie_inclusive"], indent=None))
print("emit_sweep", [(r["images"], r["route"], round(r["kernel_us"],1), round(r["frac"],3), round(r["frac_whole_step"],3)) for r in d.get("emit_sweep",[])])
print("eucm_100k", d["eucm_100k"]["roofline"]["frac"], d["eucm_100k"]["roofline"]["frac_whole_step"])
print("sharded_solve", {k:(v.get("iterations"), v.get("solve_ms_max_over_ranks")) for k,v in d["sharded_solve"].items()})
print("calib", d["calib_e2e"]["total_s"], d["calib_e2e"]["phases"])
print("c3", d["config3_stereo"]["emit"]["roofline"]["frac"], d["config3_stereo"]["emit"]["roofline"]["frac_whole_step"], "c5", d["config5_rig"]["emit"]["roofline"]["frac"], d["config5_rig"]["emit"]["roofline"]["frac_whole_step"])
PY
