cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
T=r05t
timeout 900 python -m pytest tests -m gpu -q -k "frontend or accumulated or refine or small_angle or calibration" > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed\|error" gpurun_out/${T}_pytest.log | tail -5; grep -B5 -A30 "Error\|FAILED" gpurun_out/${T}_pytest.log | head -100
timeout 400 python tools/bench_calib.py --no-cli --out gpurun_out/${T}_calib.json --md gpurun_out/${T}_calib.md --tag $T > gpurun_out/${T}_calib.log 2>&1; echo "calib rc=$?"; head -14 gpurun_out/${T}_calib.md; tail -5 gpurun_out/${T}_calib.log
