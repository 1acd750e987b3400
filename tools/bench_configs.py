#!/usr/bin/env python3
"""One-GPU measurements of BASELINE.json's configs 2-5 (config 1 is the plumbing case of tests/test_gpu_frontend.py):
emit pass (residual + all Jacobian blocks to HBM), fused normal-equation build, full LM solve.  Prints a markdown
table + one JSON line per config; tools only, bench.py stays the driver's contract (config named by `metric`).

Every config carries two `roofline` objects computed as bench.py's: `roofline_emit` (HBM: algorithmic bytes of the pass,
16 + 16 + 16 (K + 6 L) per observation, / the emit launch(es) by HIP events / 8 TB/s) and `roofline_jtj` (FP64 vector pipe:
exact flop counts per observation -- evaluate 200 / 197 / 346 for EUCM / UCM / Mei + 48 per extra chain member, Gram
2 (P + 1)(P + 2) -- / the fused Gram launch / 78.6 TFLOP/s).  tools/prof_configs.sh runs one config at a time under
rocprofv3 (--kernel-trace --stats, then the FETCH_SIZE / WRITE_SIZE passes) so that every kernel of every config has a
tracked summary under profiles/.

usage: python tools/bench_configs.py [reps] [--config N] [--only emit|jtj|solve]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from visgeom_amd import CalibrationProblem, synthetic  # noqa: E402
from visgeom_amd import capi as _capi  # noqa: E402

_capi.hooks_from_env()  # legacy VG_* switches -> vg_debug_set

_args = [a for a in sys.argv[1:]]
ONLY_CONFIG = int(_args[_args.index("--config") + 1]) if "--config" in _args else None
ONLY = _args[_args.index("--only") + 1] if "--only" in _args else None
_pos = [a for i, a in enumerate(_args) if not a.startswith("--") and (i == 0 or _args[i - 1] not in ("--config", "--only"))]
REPS = int(_pos[0]) if _pos else 100
KOF = {"eucm": 6, "ucm": 5, "mei": 10}
EVAL_FLOPS = {"eucm": 200, "ucm": 197, "mei": 346}   # counted from the restatement
HBM_PEAK, FP64_PEAK = 8.0e12, 78.6e12


def gram_flops_per_obs(model, L):
    P = KOF[model] + 6 * L
    return EVAL_FLOPS[model] + 48 * (L - 1) + 2 * (P + 1) * (P + 2)


def timed(fn, reps=REPS):
    for _ in range(max(3, reps // 10)):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3   # seconds


def build(cfg):
    p = CalibrationProblem(0)
    if cfg == 2 or cfg == 4:
        model, n = ("eucm", 1000) if cfg == 2 else ("mei", 10000)
        d = synthetic.make_mono(model, n, cfg)
        cam = p.add_camera(model, d["init_intrinsics"])
        seq = p.add_transform(False, d["init_poses"])
        dss = [(p.add_dataset(cam, [(seq, 0)], d["board"], d["corners"]), model, 1, n)]
        gt = [d["gt_intrinsics"]]
        name = "config %d: %s mono, %d images" % (cfg, model.upper(), n)
    elif cfg == 3:
        s = synthetic.make_stereo(2000)
        c1 = p.add_camera("eucm", s["init_intrinsics1"])
        c2 = p.add_camera("eucm", s["init_intrinsics2"])
        x12 = p.add_transform(True, s["init_xi12"])
        seq = p.add_transform(False, s["init_poses"])
        dss = [(p.add_dataset(c1, [(seq, 0)], s["board"], s["corners1"]), "eucm", 1, 2000),
               (p.add_dataset(c2, [(x12, 1), (seq, 0)], s["board"], s["corners2"]), "eucm", 2, 2000)]
        gt = [s["gt_intrinsics1"], s["gt_intrinsics2"]]
        name = "config 3: stereo 2 x EUCM + xiCam12, 2000 pairs"
    else:
        r = synthetic.make_rig(5000)
        cams = [p.add_camera(m, r["init_intrinsics"][k]) for k, m in enumerate(r["models"])]
        x1k = [p.add_transform(True, r["init_xi1k"][k]) for k in range(3)]
        seq = p.add_transform(False, r["init_poses"])
        dss = [(p.add_dataset(cams[0], [(seq, 0)], r["board"], r["corners"][0]), r["models"][0], 1, 5000)]
        for k in range(3):
            dss.append((p.add_dataset(cams[k + 1], [(x1k[k], 1), (seq, 0)], r["board"], r["corners"][k + 1]), r["models"][k + 1], 2, 5000))
        gt = r["gt_intrinsics"]
        name = "config 5: rig [UCM, EUCM, EUCM, Mei], 5000 frames"
    p.finalize()
    return p, dss, gt, name


def main():
    rows = []
    for cfg in ((ONLY_CONFIG,) if ONLY_CONFIG else (2, 3, 4, 5)):
        p, dss, gt, name = build(cfg)
        outs = [p.alloc_outputs(ds) for ds, _, _, _ in dss]
        grams = [p.alloc_gram(ds) for ds, _, _, _ in dss]
        n_obs = sum(n * 96 for _, _, _, n in dss)
        bytes_emit = sum(n * 96 * (32 + 16 * (KOF[m] + 6 * L)) for _, m, L, n in dss)

        def emit():
            p.prepare()
            p.evaluate_all(outs)   # every dataset of the problem in one pass (vg_problem_evaluate: merged launches)

        def emit_only():
            p.evaluate_all(outs)

        def emit_per_dataset():
            for (ds, _, _, _), (res, ji, jm) in zip(dss, outs):
                p.evaluate_dataset(ds, res, ji, jm)

        def jtj():
            p.prepare()
            if len(dss) > 1:   # every dataset in one pass (vg_problem_gram_fused_sum: merged Gram launch + ONE sum launch)
                p.gram_fused_sum_all([g for g, _ in grams], [s for _, s in grams])
            else:
                for (ds, _, _, _), (gram, gsum) in zip(dss, grams):
                    p.gram_fused_sum(ds, gram, gsum)

        def gram_only():   # the fused Gram launch(es) alone: what roofline_jtj prices
            if len(dss) > 1:
                p.gram_fused_all([g for g, _ in grams])
            else:
                p.gram_fused(dss[0][0], grams[0][0])

        if ONLY in ("emit", "jtj", "solve"):   # one kind of launch only: a clean rocprofv3 / PMC pass
            if ONLY == "emit":
                print(json.dumps({"config": name, "only": "emit", "step_ms": timed(emit) * 1e3}))
            elif ONLY == "jtj":
                print(json.dumps({"config": name, "only": "jtj", "jtj_ms": timed(jtj) * 1e3}))
            else:
                s = p.solve(max_num_iterations=200)
                print(json.dumps({"config": name, "only": "solve", "solve_ms": s["total_seconds"] * 1e3, "iterations": s["num_iterations"]}))
            p.close()
            continue
        t_emit, t_emit_only, t_jtj = timed(emit), timed(emit_only), timed(jtj)
        p.prepare()
        gram_only()
        t_gram_only = timed(gram_only)
        flops_jtj = sum(n * 96 * gram_flops_per_obs(m, L) for _, m, L, n in dss)
        t_per_ds = timed(emit_per_dataset)
        spread = sorted(timed(emit_only, max(20, REPS // 5)) for _ in range(5))   # min / median / max of 5 runs
        x0 = p.get_parameters()
        best = None
        for _ in range(3):
            p.set_parameters(x0)
            torch.cuda.synchronize()
            s = p.solve(max_num_iterations=200)
            if best is None or s["total_seconds"] < best["total_seconds"]:
                best = s
        x = p.get_parameters()
        err, off = 0.0, 0
        for g in gt:
            err = max(err, float(np.max(np.abs(x[off:off + g.size] - g) / np.maximum(np.abs(g), 1.0))))
            off += g.size
        row = {"config": name, "observations": n_obs, "emit_ms": t_emit * 1e3, "evals_per_s": n_obs / t_emit,
               "emit_kernels_GBps": bytes_emit / t_emit_only / 1e9, "emit_bytes": bytes_emit, "jtj_fused_ms": t_jtj * 1e3,
               "emit_per_dataset_launches_GBps": bytes_emit / t_per_ds / 1e9,
               "emit_only_ms_min_median_max": [spread[0] * 1e3, spread[2] * 1e3, spread[4] * 1e3],
               "frac_of_hbm_peak": bytes_emit / t_emit_only / 8e12,
               "roofline_emit": {"bound": "hbm", "achieved": bytes_emit / t_emit_only / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                                 "frac": bytes_emit / t_emit_only / HBM_PEAK, "frac_whole_step": bytes_emit / t_emit / HBM_PEAK,
                                 "algorithmic_bytes": bytes_emit, "emit_launches_ms": t_emit_only * 1e3, "step_ms": t_emit * 1e3},
               "roofline_jtj": {"bound": "fp64-vector", "achieved": flops_jtj / t_gram_only / 1e12, "peak": FP64_PEAK / 1e12,
                                "unit": "TFLOP/s", "frac": flops_jtj / t_gram_only / FP64_PEAK,
                                "frac_whole_iteration": flops_jtj / t_jtj / FP64_PEAK, "algorithmic_flops": flops_jtj,
                                "flops_per_observation": {"%s L=%d" % (m, L): gram_flops_per_obs(m, L) for _, m, L, _ in dss},
                                "gram_launch_ms": t_gram_only * 1e3, "iteration_ms": t_jtj * 1e3},
               "solve_ms_per_iteration": best["total_seconds"] * 1e3 / max(best["num_iterations"], 1),
               "solve_ms": best["total_seconds"] * 1e3, "solve_iterations": best["num_iterations"],
               "termination": best["termination"], "global_columns": best["num_global_columns"],
               "max_rel_intrinsics_error_vs_generating": err}
        rows.append(row)
        print(json.dumps(row))
        p.close()
    if not rows:
        return
    print()
    print("| config | observations | step (prep + emit) ms | evals/s | emit launches: GB/s, frac of 8 TB/s (whole step) | fused Gram launch: ms, TFLOP/s, frac of 78.6 (whole iteration) | JtJ ms/iter | LM solve ms (iterations, ms/iteration) | G | intrinsics vs generating |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        e, j = r["roofline_emit"], r["roofline_jtj"]
        print("| %s | %d | %.4f | %.3e | %.0f, %.3f (%.3f) | %.4f, %.1f, %.3f (%.3f) | %.4f | %.2f (%d, %.3f) | %d | %.1e |" % (
            r["config"], r["observations"], r["emit_ms"], r["evals_per_s"], e["achieved"], e["frac"], e["frac_whole_step"],
            j["gram_launch_ms"], j["achieved"], j["frac"], j["frac_whole_iteration"], r["jtj_fused_ms"],
            r["solve_ms"], r["solve_iterations"], r["solve_ms_per_iteration"], r["global_columns"],
            r["max_rel_intrinsics_error_vs_generating"]))


if __name__ == "__main__":
    main()
