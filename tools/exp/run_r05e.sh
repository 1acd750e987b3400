cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python tools/exp/emit_sweep_probe.py > gpurun_out/r05e_emit_sweep_ab.txt 2>&1
python - <<'PY'
import re, collections
rows = collections.defaultdict(dict)
for ln in open("gpurun_out/r05e_emit_sweep_ab.txt"):
    m = re.match(r"(.{24}) +(\d+) images .*kernel +([\d.]+) us .*step +([\d.]+) us", ln)
    if m:
        rows[int(m.group(2))].setdefault(m.group(1).strip(), []).append((float(m.group(3)), float(m.group(4))))
names = ["nt=never  route=default", "nt=always route=default", "nt=never  route=inline", "nt=always route=inline", "nt=never  route=prep", "nt=always route=prep", "library defaults"]
print("step us (min of the two sweeps) | " + " | ".join(names))
for n in sorted(rows):
    print(n, " | ".join("%.1f" % min(s for _, s in rows[n].get(k, [(0, 0)])) for k in names))
PY
