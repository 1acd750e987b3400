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

from visgeom_amd import _build as _b  # noqa: E402

if os.environ.get("AB_LIB"):   # same-box A/B against a variant library (python -m visgeom_amd._build --variant NAME ...)
    _b.LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.environ["AB_LIB"])
from visgeom_amd import capi as _capi  # noqa: E402

_capi.hooks_from_env()  # legacy VG_* switches -> vg_debug_set

_args = [a for a in sys.argv[1:]]
ONLY_CONFIG = int(_args[_args.index("--config") + 1]) if "--config" in _args else None
ONLY = _args[_args.index("--only") + 1] if "--only" in _args else None
_pos = [a for i, a in enumerate(_args) if not a.startswith("--") and (i == 0 or _args[i - 1] not in ("--config", "--only"))]
REPS = int(_pos[0]) if _pos else 100
from visgeom_amd.benchlib import EVAL_FLOPS, FP64_PEAK, HBM_PEAK, KOF, gram_flops_per_obs, passes  # noqa: E402
from visgeom_amd.benchlib import build as _build_config, timed as _timed  # noqa: E402


def timed(fn, reps=REPS):
    return _timed(fn, reps)


def build(cfg):
    return _build_config(cfg, 0)


def main():
    rows = []
    for cfg in ((ONLY_CONFIG,) if ONLY_CONFIG else (2, 3, 4, 5)):
        p, dss, gt, name = build(cfg)
        f = passes(p, dss)
        emit, emit_only, emit_per_dataset, jtj, gram_only = f["emit"], f["emit_only"], f["emit_per_dataset"], f["jtj"], f["gram_only"]
        n_obs = sum(n * 96 for _, _, _, n in dss)
        bytes_emit = sum(n * 96 * (32 + 16 * (KOF[m] + 6 * L)) for _, m, L, n in dss)

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
