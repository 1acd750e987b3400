"""The product entry point end to end, phase by phase: the reference's program is `calib a.json`
(test/calibration/generic_calibration.cpp:32-44 -> addResiduals -> compute, unified_calibration.cpp:350-356,39-89).  Generates
calibration files at BASELINE's sizes (config 2: EUCM mono 1 k images; the headline: EUCM mono 10 k images; config 3: stereo 2 k
pairs; config 4's camera: Mei mono 10 k), poses initialised from scratch (estimateInitialGrid, :1066-1158), runs them through
vg_calibration_* with the library's per-phase clock and through the `calib` executable under a wall clock, and times the pose
initialisation kernel (f2) on its own.

    python tools/bench_calib.py [--out gpurun_out/calib_e2e.json] [--md profiles/rNN_calib_e2e.md] [--small]

Needs a GPU (the refinement and the solve have no CPU path)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from visgeom_amd import benchlib  # noqa: E402


def markdown(rows, poses, tag):
    ph = benchlib.CALIB_PHASES
    out = ["# `calib a.json` end to end, per phase (%s)" % tag, "",
           "Seconds of wall clock inside the library (`vg_calibration_get_timings`), best of the runs by total; `cli` = the `calib` "
           "executable under a wall clock (process start, HIP initialisation, code-object load included).  GPU phases: "
           + ", ".join(benchlib.CALIB_GPU_PHASES) + " (kernels and the copies they need); the rest is host work.", "",
           "| workload | JSON MB | " + " | ".join(p[:-2] for p in ph) + " | total | cli | host / GPU | images/s |",
           "|---|---|" + "---|" * (len(ph) + 4)]
    for r in rows:
        out.append("| %s | %.1f | " % (r["workload"].split(",")[0] + "," + r["workload"].split(",")[1], r["json_megabytes"]) +
                   " | ".join("%.4f" % r["phases"][p] for p in ph) +
                   " | %.3f | %s | %.1f | %.0f |" % (r["total_s"], "%.2f" % r["cli_wall_s"] if "cli_wall_s" in r else "-",
                                                     r["host_phases_s"] / max(r["gpu_phases_s"], 1e-9), r["images_per_second_end_to_end"]))
    out += ["", "Refinement kernel (`vg_pose_lm_kernel`) inside those runs, solve, accuracy:", "",
            "| workload | refine kernel ms | images | LM iterations mean / max | solve iterations | solve ms | max rel. intrinsics error |",
            "|---|---|---|---|---|---|---|"]
    for r in rows:
        out.append("| %s | %.3f | %d | %.1f / %d | %d | %.2f | %.2e |" % (
            r["workload"].split(",")[0] + "," + r["workload"].split(",")[1], r["refine_kernel_s"] * 1e3, r["refine_images"],
            r["refine_iterations_mean"], r["refine_iterations_max"], r["solve"]["num_iterations"], r["solve"]["total_seconds"] * 1e3,
            r["max_rel_intrinsics_error_vs_generating"]))
    if poses:
        out += ["", "Pose initialisation on its own (f2, `vg_refine_poses` from the 4-corner poses at the initial intrinsics):", "",
                "| workload | call ms (H2D + kernel + D2H) | iterations mean / p99 / max | converged | max pose error vs generating |",
                "|---|---|---|---|---|"]
        for r in poses:
            out.append("| %s | %.3f | %.1f / %.0f / %d | %d | %.3g |" % (
                r["workload"].split(",")[0] + "," + r["workload"].split(",")[1], r["refine_call_ms"], r["iterations_mean"],
                r["iterations_p99"], r["iterations_max"], r["converged"], r["max_pose_error_vs_generating"]))
        out += ["", "(CPU baseline of this phase: `bench.py`, section `pose_init.cpu_baseline`.)"]
    return "\n".join(out) + "\n"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--md", default=None)
    ap.add_argument("--tag", default="this run")
    ap.add_argument("--small", action="store_true", help="a tenth of every size (rehearsal)")
    ap.add_argument("--runs", type=int, default=2)
    ap.add_argument("--no-cli", action="store_true")
    ap.add_argument("--only", default=None, help="comma-separated subset of mono_eucm_1k,mono_eucm_10k,stereo_2k,mono_mei_10k")
    a = ap.parse_args()
    scale = 10 if a.small else 1
    plan = [("mono_eucm_1k", "mono_eucm", 1000), ("mono_eucm_10k", "mono_eucm", 10000), ("stereo_2k", "stereo", 2000),
            ("mono_mei_10k", "mono_mei", 10000)]
    if a.only:
        plan = [x for x in plan if x[0] in a.only.split(",")]
    rows, poses = [], []
    for name, workload, n in plan:
        r = benchlib.calib_e2e(workload, n // scale, runs=a.runs, cli=not a.no_cli)
        r["name"] = name
        rows.append(r)
        print(json.dumps(r), flush=True)
    for model, n in (("eucm", 10000), ("mei", 10000)):
        if a.only and ("mono_%s_10k" % model) not in a.only.split(","):
            continue
        r = benchlib.pose_init(model, n // scale)
        poses.append(r)
        print(json.dumps(r), flush=True)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            json.dump({"calib_e2e": rows, "pose_init": poses}, f, indent=1)
    md = markdown(rows, poses, a.tag)
    if a.md:
        with open(a.md, "w") as f:
            f.write(md)
    print(md)


if __name__ == "__main__":
    main()
