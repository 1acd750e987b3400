cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
T=r05o
timeout 300 python -m pytest tests -m gpu -x -q -k "refine or frontend or pose" > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed\|error" gpurun_out/${T}_pytest.log | tail -5; grep -B5 -A25 "Error\|FAILED" gpurun_out/${T}_pytest.log | head -60
timeout 200 python tools/bench_calib.py --only mono_eucm_10k,mono_mei_10k --no-cli --out gpurun_out/${T}_calib.json --md gpurun_out/${T}_calib.md --tag $T > gpurun_out/${T}_calib.log 2>&1; echo "calib rc=$?"; tail -22 gpurun_out/${T}_calib.md
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05o_calib.json"))
for r in d["pose_init"]: print(r["workload"], "kernel ms", r["kernel_ms"], "call ms", r["refine_call_ms"], "frac", r["roofline"]["frac"], r["iterations_mean"], r["iterations_max"])
PY
