"""Helper of tests/test_gpu_production.py (run as a subprocess, once per library): every product pass on small builds of
BASELINE configs 2-5 -> one .npz of raw results.  The library is the one VISGEOM_AMD_LIBRARY selects (visgeom_amd/capi.py)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from visgeom_amd import benchlib, capi  # noqa: E402

out = {}
out["has_hooks"] = np.array([int(capi.has_debug_hooks())])
out["library"] = np.array([os.path.abspath(capi.lib_path())])
for cfg, images in ((2, 1000), (3, 300), (4, 400), (5, 200)):
    p, dss, gt, name = benchlib.build(cfg, 0, images)
    f = benchlib.passes(p, dss)
    f["emit"]()                       # chain prep + (merged) emit launch
    f["jtj"]()                        # chain prep + (merged) fused Gram + sums
    torch.cuda.synchronize()
    outs, grams = f["keep"]
    for k, (res, ji, jm) in enumerate(outs):
        out["c%d_d%d_res" % (cfg, k)] = res.cpu().numpy()
        out["c%d_d%d_ji" % (cfg, k)] = ji.cpu().numpy()
        for l, t in enumerate(jm):
            out["c%d_d%d_jm%d" % (cfg, k, l)] = t.cpu().numpy()
        # second-pass Gram from the emitted rows (the matrix-core kernel)
        g2 = torch.empty_like(grams[k][0])
        p.gram_from_rows(dss[k][0], res, ji, jm, g2)
        out["c%d_d%d_gram_rows" % (cfg, k)] = g2.cpu().numpy()
    for k, (g, s) in enumerate(grams):
        out["c%d_d%d_gram" % (cfg, k)] = g.cpu().numpy()
        out["c%d_d%d_gram_sum" % (cfg, k)] = s.cpu().numpy()
    s = p.solve(max_num_iterations=15)
    out["c%d_solution" % cfg] = p.get_parameters()
    out["c%d_solve" % cfg] = np.array([s["num_iterations"], s["final_cost"], s["initial_cost"]], dtype=np.float64)
    p.close()
np.savez(sys.argv[1], **out)
print("wrote", sys.argv[1], "from", capi.lib_path())
