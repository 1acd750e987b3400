"""Per-dataset emit rate of the rig (config 5) and the stereo pair (config 3): which dataset streams slowly.
usage: python tools/exp/rig_emit_probe.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.argv = [sys.argv[0]]
from tools import bench_configs as B  # noqa: E402

for cfg in (3, 5):
    p, dss, gt, name = B.build(cfg)
    outs = [p.alloc_outputs(ds) for ds, _, _, _ in dss]
    p.prepare()
    p.evaluate_all(outs)
    print(name)
    for (ds, m, L, n), (res, ji, jm) in zip(dss, outs):
        nbytes = n * 96 * (32 + 16 * (B.KOF[m] + 6 * L))
        t = B.timed(lambda: p.evaluate_dataset(ds, res, ji, jm))
        print("  dataset %d %s L=%d: %.1f us, %.0f GB/s (%.3f of 8 TB/s), %d B / observation" % (ds, m, L, t * 1e6, nbytes / t / 1e9, nbytes / t / 8e12,
                                                                                               32 + 16 * (B.KOF[m] + 6 * L)))
    nb = sum(n * 96 * (32 + 16 * (B.KOF[m] + 6 * L)) for _, m, L, n in dss)
    from visgeom_amd import capi
    for hook, what in ((1, "contiguous pieces of equal tile counts"), (2, "contiguous pieces of equal bytes"), (4, "eighths of every dataset, problem order"), (0, "default: eighths, widest rows first"), (4, "eighths of every dataset, problem order"), (0, "default: eighths, widest rows first")):
        capi.debug_set("emit_equal_tiles", hook)
        t = B.timed(lambda: p.evaluate_all(outs))
        print("  all datasets, merged launch, %s: %.1f us, %.0f GB/s" % (what, t * 1e6, nb / t / 1e9))
    capi.debug_set("emit_equal_tiles", 0)
