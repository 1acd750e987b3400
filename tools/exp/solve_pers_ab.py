"""vg_problem_solve with the one-shot / persistent direct Gram kernel, same process, alternating through the hook `gram_persistent`.
usage: python tools/exp/solve_pers_ab.py [model] [images] [repeats]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from visgeom_amd import synthetic as S
from visgeom_amd import capi as _capi
from visgeom_amd.problem import CalibrationProblem

model = sys.argv[1] if len(sys.argv) > 1 else "eucm"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
d = S.make_mono(model, n, 1 if model == "eucm" else 4)
res = {}
for r in range(reps):
    for name, hook in (("one-shot", 1), ("by size", 0)):
        _capi.debug_set("gram_persistent", hook)
        p = CalibrationProblem(0)
        c = p.add_camera(model, d["init_intrinsics"]); s = p.add_transform(False, d["init_poses"])
        p.add_dataset(c, [(s, 0)], d["board"], d["corners"]); p.finalize()
        summ = p.solve(max_num_iterations=100)
        if r:
            res.setdefault(name, []).append((summ["total_seconds"] * 1e3, summ["host_seconds"] * 1e3, summ["evaluate_seconds"] * 1e3 / max(1, summ["num_iterations"]), summ["num_iterations"], summ["final_cost"]))
        p.close()
_capi.debug_set("gram_persistent", 0)
for name, v in res.items():
    print("%s n=%d %-9s total ms: %s | set-up: %s | ms/it: %s | iterations %d cost %.9e" % (model, n, name, " ".join("%.3f" % a[0] for a in v), " ".join("%.3f" % a[1] for a in v),
                                                                                 " ".join("%.4f" % a[2] for a in v), v[-1][3], v[-1][4]))
