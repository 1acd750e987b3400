cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
T=r05j
timeout 300 python tools/exp/launch_walk_probe.py > gpurun_out/${T}_launch_walk.txt 2>&1; echo "probe rc=$?"; grep -v amdgpu.ids gpurun_out/${T}_launch_walk.txt | tail -20
timeout 900 python -m pytest tests -m gpu -x -q -k "parity or gram or rig or full_configs or block_group or host_adapter" > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed\|error" gpurun_out/${T}_pytest.log | tail -5; tail -30 gpurun_out/${T}_pytest.log | grep -v "^\.\|RCCL\|HIP ver\|ROCm\|Hostname\|Librccl" | head -40
