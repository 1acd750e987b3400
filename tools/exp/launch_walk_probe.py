"""Walker workgroups inside the merged emit / Gram launches against the chain-prep launch in front of them, same
box, alternating through the hook `no_launch_walk` (1: chain-prep launch; 0: walkers, self-walking datasets first; 2: walkers, datasets
in widest-first order); the rows and Gram blocks of the routes must be bit-identical.
    python tools/exp/launch_walk_probe.py [config ...]"""
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import torch  # noqa: E402

from visgeom_amd import benchlib, capi  # noqa: E402

cfgs = [int(c) for c in sys.argv[1:]] or [3, 5]
for cfg in cfgs:
    p, dss, gt, name = benchlib.build(cfg)
    f = benchlib.passes(p, dss)
    outs, grams = f["keep"]
    ref = None
    for rep in range(2):
        for hook, label in ((1, "chain-prep launch"), (0, "walkers, self-walking first"), (2, "walkers, widest first")):
            capi.debug_set("no_launch_walk", hook)
            f["emit"]()
            f["jtj"]()
            torch.cuda.synchronize()
            snap = [t.clone() for o in outs for t in ([o[0], o[1]] + list(o[2]))] + [g.clone() for g, _ in grams] + [s.clone() for _, s in grams]
            if ref is None:
                ref = snap
            same = all(torch.equal(a, b) for a, b in zip(ref, snap))
            t_emit = benchlib.timed(f["emit"], 300) * 1e6
            t_jtj = benchlib.timed(f["jtj"], 300) * 1e6
            print("%-50s %-30s emit step %7.2f us   JtJ iteration %7.2f us   rows / Gram blocks identical to the first route: %s" % (
                name, label, t_emit, t_jtj, same), flush=True)
    capi.debug_set("no_launch_walk", 0)
    p.close()
