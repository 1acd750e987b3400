cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
T=r05g
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed\|error" gpurun_out/${T}_pytest.log | tail -5; tail -40 gpurun_out/${T}_pytest.log | head -80
timeout 600 python tools/bench_configs.py 200 2>/dev/null | tail -6
timeout 600 python tools/exp/solve_stress.py 2>&1 | tail -5
