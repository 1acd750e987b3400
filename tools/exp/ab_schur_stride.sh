cd $GRAFT_REPO_ROOT
for v in default evenstride; do
  for w in "eucm 10000" "rig 5000"; do
    set -- $w
    if [ $v = default ]; then unset AB_LIB; else export AB_LIB=visgeom_amd/lib/variants/libvisgeom_amd_$v.so; fi
    bash tools/exp/prof_solve.sh $1 $2 2>&1 | grep -i "schur_rows\|strided_sum\|backsub" | sed "s/^/$v $1: /"
  done
done
# LDS bank-conflict share of the rows + Gram kernel under both strides (own counter pass)
cd /tmp && export TMPDIR=/tmp
for v in default evenstride; do
  if [ $v = default ]; then unset AB_LIB; else export AB_LIB=visgeom_amd/lib/variants/libvisgeom_amd_$v.so; fi
  for w in "eucm 10000" "rig 5000" "stereo 2000"; do
    set -- $w
    O=$GRAFT_REPO_ROOT/gpurun_out/pmc_schur_${v}_$1; rm -rf $O; mkdir -p $O
    timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $O -o t -- python $GRAFT_REPO_ROOT/tools/prof_solve.py $1 $2 > $O/run.log 2>&1
    python3 - "$O" "$v $1" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "vg_schur_rows_gram_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
if m:
    print("%s: vg_schur_rows_gram_kernel dispatches %d  bank-conflict cycles %.0f  LDS active %.0f  share %.3f  LDS insts %.0f waves %.0f" % (
        sys.argv[2], len(acc["SQ_LDS_IDX_ACTIVE"]), m["SQ_LDS_BANK_CONFLICT"], m["SQ_LDS_IDX_ACTIVE"], m["SQ_LDS_BANK_CONFLICT"] / max(m["SQ_LDS_IDX_ACTIVE"], 1), m["SQ_INSTS_LDS"], m["SQ_WAVES"]))
PY
    find $O -name '*.csv' -size +1M -delete
  done
done
