"""Persistent form of the direct Gram kernel against the one-shot kernel, same process, alternating through the hook
`gram_persistent` (1 = never, 2 / 3 = the four- / eight-wave shape whenever it applies, 0 = the library's choice by size).  usage: python tools/exp/gram_pers_probe.py [model] [images ...]"""
import os, sys, torch
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root)
from visgeom_amd import CalibrationProblem, synthetic
from visgeom_amd import capi as _capi

model = sys.argv[1] if len(sys.argv) > 1 else "eucm"
sizes = [int(x) for x in sys.argv[2:]] or [10000]
for n in sizes:
    d = synthetic.make_mono(model, n, 1)
    p = CalibrationProblem(0)
    cam = p.add_camera(model, d["init_intrinsics"]); seq = p.add_transform(False, d["init_poses"])
    ds = p.add_dataset(cam, [(seq, 0)], d["board"], d["corners"]); p.finalize()
    gram, gsum = p.alloc_gram(ds)
    gram2, gsum2 = torch.empty_like(gram), torch.empty_like(gsum)
    def t(fn, reps=300):
        for _ in range(30): fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): fn()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps * 1e3)
        return best
    def it_fused():
        p.prepare(); p.gram_fused(ds, gram)
    def it_sum():
        p.prepare(); p.gram_fused_sum(ds, gram, gsum)
    res = {}
    for rep in range(3):
        for name, hook in (("one-shot", 1), ("four waves", 2), ("eight waves", 3), ("by size", 0)):
            _capi.debug_set("gram_persistent", hook)
            res.setdefault(name, []).append((t(it_fused), t(it_sum)))
    _capi.debug_set("gram_persistent", 1); p.prepare(); p.gram_fused_sum(ds, gram, gsum)
    _capi.debug_set("gram_persistent", 2); p.prepare(); p.gram_fused_sum(ds, gram2, gsum2)
    torch.cuda.synchronize()
    _capi.debug_set("gram_persistent", 0)
    print("%s n=%d  blocks rel diff %.2e  sum rel diff %.2e" % (model, n, float((gram - gram2).abs().max() / gram.abs().max()),
                                                                float((gsum - gsum2).abs().max() / gsum.abs().max())))
    for name, v in res.items():
        print("   %-10s gram_fused us: %s | gram_fused_sum us: %s" % (name, " ".join("%.2f" % a for a, _ in v), " ".join("%.2f" % b for _, b in v)))
