# same-box A/B of the emit kernel between the default library and a variant: bash tools/exp/ab_emit_variant.sh <variant name>
cd ${GRAFT_REPO_ROOT:-.}
V=${1:?variant}
for i in 1 2 3; do for v in default $V; do
  if [ $v = default ]; then unset AB_LIB; else export AB_LIB=visgeom_amd/lib/variants/libvisgeom_amd_$v.so; fi
  python - <<'PY'
import os, sys
sys.path.insert(0, ".")
from visgeom_amd import _build
if os.environ.get("AB_LIB"): _build.LIB = os.path.join(os.getcwd(), os.environ["AB_LIB"])
from visgeom_amd import benchlib, synthetic
out = []
for model, sizes in (("eucm", [5000, 10000, 25000, 50000, 100000]), ("mei", [10000])):
    d = synthetic.make_mono(model, max(sizes), 1)
    for r in benchlib.emit_sweep(d, model, sizes, reps=300):
        out.append("%s %d: %.2f (%.3f)" % (model, r["images"], r["kernel_us"], r["frac"]))
print("%-10s" % os.path.basename(os.environ.get("AB_LIB", "default")).replace("libvisgeom_amd_", "").replace(".so", ""), " | ".join(out))
PY
done; done
