#!/bin/bash
# SQ counters of the fused Gram kernels (own run, --kernel-trace only): issue vs stall split.  usage: gpurun -- bash tools/exp/pmc_gram.sh [model]
R=${GRAFT_REPO_ROOT:-$(pwd)}; M=${1:-eucm}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/pmc_gram_$M; mkdir -p $O
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/a -o t -- python $R/tools/exp/gram_probe.py $M > $O/a.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/b -o t -- python $R/tools/exp/gram_probe.py $M > $O/b.log 2>&1
python3 - <<PY
import csv,glob,collections
for tag in "ab":
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % tag, recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"][:60]
            if "gram" not in k: continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,v in acc.items():
            print(k, {c: sum(x)/len(x) for c,x in v.items()}, "n=%d" % len(next(iter(v.values()))))
PY
