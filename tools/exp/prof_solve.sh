#!/bin/bash
# per-kernel time of the LM loop (rocprofv3 --kernel-trace --stats).  usage: gpurun -- bash tools/exp/prof_solve.sh <rig|eucm|mei> [n]
R=${GRAFT_REPO_ROOT:-$(pwd)}; W=${1:-mei}; N=${2:-10000}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_solve_$W; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python $R/tools/prof_solve.py $W $N > $O/run.log 2>&1
tail -4 $O/run.log
f=$(find $O -name '*kernel_stats.csv' | head -1); python3 - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:16]:
    print("%-70s calls %6s avg %8.2f us total %6.1f%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
PY
find $O -name '*.csv' -size +4M -delete
