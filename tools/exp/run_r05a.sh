cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{ nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; grep -m1 'model name' /proc/cpuinfo; } > gpurun_out/r05a_box.txt
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05a_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r05a_pytest.log
timeout 600 python tools/exp/calib_r04_baseline.py 10000 > gpurun_out/r05a_calib_r04_baseline.txt 2>&1; echo "baseline rc=$?"; cat gpurun_out/r05a_calib_r04_baseline.txt | tail -3
timeout 900 python tools/bench_calib.py --out gpurun_out/r05a_calib_e2e.json --md gpurun_out/r05a_calib_e2e.md --tag r05a > gpurun_out/r05a_calib_e2e.log 2>&1; echo "bench_calib rc=$?"; tail -40 gpurun_out/r05a_calib_e2e.log
