#!/bin/bash
# the rig's LM loop (config 5, G = 45): host-driven vs device-resident, then the device-resident loop's kernels and gaps
# usage: gpurun -- bash tools/exp/rig_device_loop.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
python $R/tools/exp/rig_loop_probe.py
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/tr
VG_SOLVER_DEVICE_LOOP=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $R/tools/prof_solve.py rig 5000 > /tmp/tr.log 2>&1
grep -v "^[EWI]2026" /tmp/tr.log | tail -3
python $R/tools/exp/trace_gaps.py $(find /tmp/tr -name "*kernel_trace.csv" | head -1) | head -${LINES_OUT:-24}
