#!/bin/bash
# kernels and gaps of an LM loop (rocprofv3 kernel trace -> tools/exp/trace_gaps.py).  usage: gpurun -- bash tools/exp/loop_trace.sh <rig|stereo|eucm|mei|ucm> [n]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/tr
REPS=${REPS:-4} rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $R/tools/prof_solve.py $1 ${2:-10000} > /tmp/tr.log 2>&1
grep -v "^[EWI]2026" /tmp/tr.log | tail -2
python $R/tools/exp/trace_gaps.py $(find /tmp/tr -name "*kernel_trace.csv" | head -1) | grep -v "rocclr" | head -${LINES_OUT:-22}
