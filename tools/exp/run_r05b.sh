cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
T=r05b
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed\|error" gpurun_out/${T}_pytest.log | tail -5
timeout 900 python tools/bench_calib.py --out gpurun_out/${T}_calib_e2e.json --md gpurun_out/${T}_calib_e2e.md --tag $T > gpurun_out/${T}_calib_e2e.log 2>&1; echo "bench_calib rc=$?"; tail -22 gpurun_out/${T}_calib_e2e.md
timeout 900 python tools/bench_configs.py 200 > gpurun_out/${T}_bench_configs.txt 2> gpurun_out/${T}_bench_configs.err; echo "bench_configs rc=$?"; tail -30 gpurun_out/${T}_bench_configs.txt
timeout 1200 python bench.py > gpurun_out/${T}_bench_n1.json 2> gpurun_out/${T}_bench_n1.err; echo "bench rc=$?"; tail -c 6000 gpurun_out/${T}_bench_n1.json; tail -5 gpurun_out/${T}_bench_n1.err
