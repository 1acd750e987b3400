#!/bin/bash
# The production library (no debug hooks) on the GPU: the whole GPU suite, smoke(), the bench headline + jtj -> one text record.
#   gpurun -- 'bash tools/production_check.sh gpurun_out/r06_production_build.txt'
out=${1:-gpurun_out/production_build.txt}
mkdir -p "$(dirname "$out")"
export VISGEOM_AMD_LIBRARY=production
{
  echo "# production library: $(ls -l visgeom_amd/lib/production/libvisgeom_amd.so)"
  echo "# exported vg_* symbols: $(nm -D --defined-only visgeom_amd/lib/production/libvisgeom_amd.so | grep -c ' T vg_') (vg_debug_set: $(nm -D --defined-only visgeom_amd/lib/production/libvisgeom_amd.so | grep -c vg_debug_set))"
  echo "## VISGEOM_AMD_LIBRARY=production python -m pytest tests -m gpu -q"
  python -m pytest tests -m gpu -q -rs 2>&1 | tail -60
  echo "## smoke()"
  python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -2
  echo "## bench.py (headline line: value, ms_per_step, roofline.frac, jtj)"
  python bench.py --steps 20 --warmup 5 > /tmp/bench_prod.json 2>/tmp/bench_prod.err
  python - <<'PY'
import json
d = json.loads(open("/tmp/bench_prod.json").read().strip().splitlines()[-1])
print(json.dumps({k: d.get(k) for k in ("metric", "value", "unit", "ms_per_step", "library")}))
print("roofline", json.dumps({k: d["roofline"].get(k) for k in ("kernel", "achieved", "frac", "avg_launch_ms")}))
print("jtj", json.dumps(d.get("jtj"))[:900])
PY
  unset VISGEOM_AMD_LIBRARY
  echo "## the same bench on the hooks library (same box)"
  python bench.py --steps 20 --warmup 5 > /tmp/bench_hooks.json 2>/tmp/bench_hooks.err
  python - <<'PY'
import json
d = json.loads(open("/tmp/bench_hooks.json").read().strip().splitlines()[-1])
print(json.dumps({k: d.get(k) for k in ("metric", "value", "unit", "ms_per_step", "library")}))
print("roofline", json.dumps({k: d["roofline"].get(k) for k in ("kernel", "achieved", "frac", "avg_launch_ms")}))
print("jtj", json.dumps(d.get("jtj"))[:900])
PY
} > "$out" 2>&1
tail -40 "$out"
