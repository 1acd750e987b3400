"""Measurement helpers shared by bench.py (the driver's contract line and its per-config sections) and tools/bench_configs.py:
BASELINE.json's configs 2-5 built on one GPU, timed with HIP events on the launch stream, priced against the roofline that
bounds each pass.  Not part of the product path (nothing here is called by the library or the solver).

Algorithmic work per observation:
  emit pass   16 (observation) + 16 (residual pair) + 16 (K + 6 L) (Jacobian rows) bytes         -> HBM, 8 TB/s
  fused Gram  EVAL_FLOPS[model] + 48 (L - 1) + 2 (P + 1)(P + 2) flops, P = K + 6 L               -> FP64 vector pipe, 78.6 TFLOP/s
"""
import numpy as np

KOF = {"eucm": 6, "ucm": 5, "mei": 10}
EVAL_FLOPS = {"eucm": 200, "ucm": 197, "mei": 346}   # counted from the restatement
HBM_PEAK, FP64_PEAK = 8.0e12, 78.6e12
N_CORNERS = 96


def gram_flops_per_obs(model, L):
    P = KOF[model] + 6 * L
    return EVAL_FLOPS[model] + 48 * (L - 1) + 2 * (P + 1) * (P + 2)


def emit_bytes_per_obs(model, L):
    return 32 + 16 * (KOF[model] + 6 * L)


def timed(fn, reps, warm_s=0.02):
    """seconds per call: HIP events on torch's current stream (the stream the problems launch on) around `reps` calls, behind a
    warm-up of at least max(3, reps / 10) calls AND `warm_s` seconds of them: the first ~10 ms of launches after a host-side pause
    run up to 10 % slower (clock ramp; profiles/r06b_emit_ramp.txt: launches 8-16 of a 100 k-image series)"""
    import time

    import torch

    n_warm, batch, t0 = 0, 4, time.perf_counter()
    while n_warm < max(3, reps // 10) or time.perf_counter() - t0 < warm_s:
        tb = time.perf_counter()
        for _ in range(batch):
            fn()
        n_warm += batch
        torch.cuda.synchronize()
        if time.perf_counter() - tb < 1e-3 and batch < 256:   # keep the device busy between the host's looks at the clock
            batch *= 2
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def emit_sweep(d, model, sizes, device=0, reps=60):
    """The emit step over n images, n in `sizes`: the first n of one generated set, the set repeated where n exceeds it (the
    kernel's work does not depend on the values).  Per size: the route (one launch with the chain walked in the emit kernel, or
    chain prep + emit on prepared frames), kernel microseconds (back-to-back launches, HIP events, after a warm-up of at least
    30 launches) and the fraction of the 8 TB/s HBM peak at the algorithmic byte count.  What the curve shows is explained in
    profiles/r06_emit_drop.md: up to ~230 MB the output stays in the 256 MiB Infinity Cache; around 1 GB a quarter of every
    launch still lands in the cache and is overwritten there by the next launch (a hump above the DRAM write rate); from
    ~2 GB on the figure is the part's streaming-write rate."""
    from . import CalibrationProblem, capi

    rows = []
    K = KOF[model]
    have = d["corners"].shape[0]
    for n in sizes:
        rep = (n + have - 1) // have
        poses = d["init_poses"] if rep == 1 else np.tile(d["init_poses"], (rep, 1))
        corners = d["corners"] if rep == 1 else np.tile(d["corners"], (rep, 1, 1))
        p = CalibrationProblem(device)
        cam = p.add_camera(model, d["init_intrinsics"])
        seq = p.add_transform(False, poses[:n])
        ds = p.add_dataset(cam, [(seq, 0)], d["board"], corners[:n])
        del poses, corners
        p.finalize()
        res, ji, jm = p.alloc_outputs(ds)

        def step():
            p.prepare()
            p.evaluate_dataset(ds, res, ji, jm)

        def emit():
            p.evaluate_dataset(ds, res, ji, jm)

        nbytes = n * N_CORNERS * emit_bytes_per_obs(model, 1)
        r = max(12, min(reps, int(60e9 / nbytes)))
        # (timed() warms up by time: with a few launches only, the first of the two measurements of a mid-size point fell into the clock ramp)
        import torch

        t_emit, t_step = timed(emit, r), timed(step, r)
        one = capi.load().vg_dataset_single_launch(p._h, ds) == 1
        out_mb = n * N_CORNERS * 16 * (K + 7) / 1e6
        rows.append({"images": n, "output_MB": out_mb, "route": "inline-chain" if one else "prep + emit",
                     "kernel_us": t_emit * 1e6, "step_us": t_step * 1e6, "frac": nbytes / t_emit / HBM_PEAK,
                     "frac_whole_step": nbytes / t_step / HBM_PEAK,
                     "regime": "inside the 256 MiB Infinity Cache" if out_mb * 1e6 < 256 * 2 ** 20 else
                               "cache-assisted (part of every launch is overwritten in the Infinity Cache by the next)" if out_mb < 1700 else "DRAM streaming"})
        p.close()
        del res, ji, jm
        torch.cuda.empty_cache()
    return rows


def build(cfg, device=0, images=None):
    """BASELINE.json config `cfg` (2..5) -> (problem, [(dataset id, model, chain length, images)], generating intrinsics, name).
    `images` scales the image count down (rehearsals); None = the configuration's own size."""
    from . import CalibrationProblem, synthetic

    p = CalibrationProblem(device)
    if cfg == 2 or cfg == 4:
        model, n = ("eucm", 1000) if cfg == 2 else ("mei", 10000)
        n = images or n
        d = synthetic.make_mono(model, n, cfg)
        cam = p.add_camera(model, d["init_intrinsics"])
        seq = p.add_transform(False, d["init_poses"])
        dss = [(p.add_dataset(cam, [(seq, 0)], d["board"], d["corners"]), model, 1, n)]
        gt = [d["gt_intrinsics"]]
        name = "config %d: %s mono, %d images" % (cfg, model.upper(), n)
    elif cfg == 3:
        n = images or 2000
        s = synthetic.make_stereo(n)
        c1 = p.add_camera("eucm", s["init_intrinsics1"])
        c2 = p.add_camera("eucm", s["init_intrinsics2"])
        x12 = p.add_transform(True, s["init_xi12"])
        seq = p.add_transform(False, s["init_poses"])
        dss = [(p.add_dataset(c1, [(seq, 0)], s["board"], s["corners1"]), "eucm", 1, n),
               (p.add_dataset(c2, [(x12, 1), (seq, 0)], s["board"], s["corners2"]), "eucm", 2, n)]
        gt = [s["gt_intrinsics1"], s["gt_intrinsics2"]]
        name = "config 3: stereo 2 x EUCM + xiCam12, %d pairs" % n
    elif cfg == 5:
        n = images or 5000
        r = synthetic.make_rig(n)
        cams = [p.add_camera(m, r["init_intrinsics"][k]) for k, m in enumerate(r["models"])]
        x1k = [p.add_transform(True, r["init_xi1k"][k]) for k in range(3)]
        seq = p.add_transform(False, r["init_poses"])
        dss = [(p.add_dataset(cams[0], [(seq, 0)], r["board"], r["corners"][0]), r["models"][0], 1, n)]
        for k in range(3):
            dss.append((p.add_dataset(cams[k + 1], [(x1k[k], 1), (seq, 0)], r["board"], r["corners"][k + 1]), r["models"][k + 1], 2, n))
        gt = r["gt_intrinsics"]
        name = "config 5: rig [UCM, EUCM, EUCM, Mei], %d frames" % n
    else:
        raise ValueError("config must be 2..5")
    p.finalize()
    return p, dss, gt, name


def passes(p, dss):
    """the closures every measurement times: (emit step, emit launches only, JtJ iteration, Gram launch(es) only, per-dataset emit)"""
    outs = [p.alloc_outputs(ds) for ds, _, _, _ in dss]
    grams = [p.alloc_gram(ds) for ds, _, _, _ in dss]

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

    return {"emit": emit, "emit_only": emit_only, "emit_per_dataset": emit_per_dataset, "jtj": jtj, "gram_only": gram_only,
            "keep": (outs, grams)}


def section(cfg, reps=200, device=0, images=None, solve_runs=2, comm=None, allreduce=None):
    """One BASELINE config as a bench.py section: emit step, merged Gram iteration and full LM solve, each emit / Gram pass with
    its own `roofline` object and the name of the kernel that rocprofv3 --kernel-trace shows for it."""
    import torch

    p, dss, gt, name = build(cfg, device, images)
    f = passes(p, dss)
    multi = len(dss) > 1
    n_obs = sum(n * N_CORNERS for _, _, _, n in dss)
    bytes_emit = sum(n * N_CORNERS * emit_bytes_per_obs(m, L) for _, m, L, n in dss)
    flops_jtj = sum(n * N_CORNERS * gram_flops_per_obs(m, L) for _, m, L, n in dss)
    t_emit, t_emit_only = timed(f["emit"], reps), timed(f["emit_only"], reps)
    t_jtj = timed(f["jtj"], reps)
    p.prepare()
    f["gram_only"]()
    t_gram = timed(f["gram_only"], reps)
    x0 = p.get_parameters()
    best = None
    for _ in range(max(1, solve_runs)):
        p.set_parameters(x0)
        torch.cuda.synchronize()
        s = p.solve(max_num_iterations=200, comm=comm, allreduce=allreduce)
        if best is None or s["total_seconds"] < best["total_seconds"]:
            best = s
    x = p.get_parameters()
    err, off = 0.0, 0
    for g in gt:
        err = max(err, float(np.max(np.abs(x[off:off + g.size] - g) / np.maximum(np.abs(g), 1.0))))
        off += g.size
    from . import capi

    one_launch = (not multi) and capi.load().vg_dataset_single_launch(p._h, dss[0][0]) == 1
    # HBM bytes per emit launch from the committed PMC passes of this configuration (tools/prof_configs.sh + prof_configs_summary.py);
    # only at the configuration's own size
    traffic, traffic_source = None, None
    if images is None:
        try:
            import json
            import os

            ent = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json"))).get("config%d_emit" % cfg)
            if ent:
                traffic = ent["hbm_bytes_per_launch"]
                traffic_source = "profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/prof_configs.sh, tag %s" % ent.get("tag")
        except Exception:
            traffic = None
    row = {
        "workload": name, "observations": n_obs,
        "emit": {"step_ms": t_emit * 1e3, "evals_per_s": n_obs / t_emit,
                 "launches": ("vg_chain_prep_multi_kernel + one emit launch for all datasets" if multi else
                              "one launch: the emit kernel walks the single-member chain" if one_launch else "chain prep + emit"),
                 "roofline": {"bound": "hbm", "kernel": "vg_emit_multi_kernel" if multi else "vg_emit_kernel", "achieved": bytes_emit / t_emit_only / 1e9,
                              "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": bytes_emit / t_emit_only / HBM_PEAK,
                              "frac_whole_step": bytes_emit / t_emit / HBM_PEAK, "algorithmic_bytes_per_launch": bytes_emit,
                              "avg_launch_ms": t_emit_only * 1e3, "traffic": traffic, "traffic_source": traffic_source}},
        "jtj": {"ms_per_iter": t_jtj * 1e3,
                "launches": "chain prep + merged Gram launch + one partial-sum launch" if multi else "fused Gram launch + partial-sum launch",
                "roofline": {"bound": "fp64", "kernel": "vg_gram_valu_multi_kernel" if multi else "vg_gram_valu_kernel", "achieved": flops_jtj / t_gram / 1e12,
                             "peak": FP64_PEAK / 1e12, "unit": "TFLOP/s", "frac": flops_jtj / t_gram / FP64_PEAK,
                             "frac_whole_iteration": flops_jtj / t_jtj / FP64_PEAK, "algorithmic_flops_per_launch": flops_jtj,
                             "flops_per_observation": {"%s L=%d" % (m, L): gram_flops_per_obs(m, L) for _, m, L, _ in dss},
                             "avg_launch_ms": t_gram * 1e3}},
        "solve": {"total_ms": best["total_seconds"] * 1e3, "iterations": best["num_iterations"],
                  "ms_per_iteration": best["total_seconds"] * 1e3 / max(best["num_iterations"], 1), "termination": best["termination"],
                  "final_cost": best["final_cost"], "global_columns": best["num_global_columns"],
                  "max_rel_intrinsics_error_vs_generating": err},
    }
    p.close()
    return row


# ---------------------------------------------------------------------------------------------------------------------
# The product entry point end to end: `calib a.json` (test/calibration/generic_calibration.cpp:32-44 -> addResiduals ->
# compute, unified_calibration.cpp:350-356,39-89) on a generated calibration file, phase by phase.

CALIB_PHASES = ("read_files_s", "parse_json_s", "geometric_init_s", "corner_upload_s", "refine_total_s", "global_init_s", "assemble_s", "solve_s",
                "readback_s", "residual_eval_s", "residual_format_s")
# phases whose time is GPU work (kernels + the copies they need); the rest is host work
CALIB_GPU_PHASES = ("corner_upload_s", "refine_total_s", "global_init_s", "solve_s", "residual_eval_s")


def write_calib_workload(directory, workload, images):
    """write the calibration JSON of a workload ('mono_eucm' / 'mono_mei': one camera, ir_data corners, poses initialised from
    scratch; 'stereo': the shape of data/calib_stereo_example.json) -> (path, dict of what was generated)"""
    import os

    from . import synthetic

    os.makedirs(directory, exist_ok=True)
    if workload in ("mono_eucm", "mono_mei"):
        model = workload.split("_")[1]
        d = synthetic.make_mono(model, images, 1)
        path = synthetic.write_calibration_json(directory, d, model, name=workload)
        n_obs = images * N_CORNERS
        gt = {"cam": d["gt_intrinsics"]}
    elif workload == "stereo":
        d = synthetic.make_stereo(images)
        path = synthetic.write_stereo_json(directory, d, name=workload)
        n_obs = 2 * images * N_CORNERS
        gt = {"camera1": d["gt_intrinsics1"], "camera2": d["gt_intrinsics2"]}
    else:
        raise ValueError(workload)
    size = sum(os.path.getsize(os.path.join(directory, f)) for f in os.listdir(directory) if f.startswith(workload) and f.endswith(".json"))
    return path, {"observations": n_obs, "json_bytes": size, "gt": gt}


def calib_e2e(workload="mono_eucm", images=10000, directory=None, runs=2, cli=True, device=0):
    """One calibration through the library's front end (vg_calibration_*: the code `calib` runs) with its per-phase clock, best of
    `runs` by total; optionally the `calib` executable itself under a wall clock.  Returns a JSON-able dict."""
    import os
    import shutil
    import subprocess
    import tempfile
    import time

    from .calibration import GenericCameraCalibration

    own = directory is None
    directory = directory or tempfile.mkdtemp(prefix="vg_calib_")
    try:
        t0 = time.perf_counter()
        path, info = write_calib_workload(directory, workload, images)
        t_write = time.perf_counter() - t0
        best = None
        for _ in range(max(1, runs)):
            c = GenericCameraCalibration(device)
            t0 = time.perf_counter()
            c.addResiduals(path)
            c.compute()
            n_ds = c.num_datasets()
            for i in range(n_ds):
                c.writeImageResidual(i, os.path.join(directory, "image_error_%d.txt" % i))
            total = time.perf_counter() - t0
            tm = c.timings()
            row = {"total_s": total, "phases": {k: tm[k] for k in CALIB_PHASES}, "refine_kernel_s": tm["refine_kernel_s"],
                   "refine_images": tm["refine_images"], "refine_iterations_mean": tm["refine_iterations"] / max(tm["refine_images"], 1),
                   "refine_iterations_max": tm["refine_max_iterations"], "residual_lines": tm["residual_lines"],
                   "corner_uploads": tm["corner_uploads"], "corner_upload_megabytes": tm["corner_upload_bytes"] / 1e6,
                   "solve": {k: c.summary[k] for k in ("num_iterations", "termination", "final_cost", "total_seconds")},
                   "max_rel_intrinsics_error_vs_generating": max(
                       float(np.max(np.abs(c.intrinsics(name) - g) / np.maximum(np.abs(g), 1.0))) for name, g in info["gt"].items())}
            c.close()
            if best is None or row["total_s"] < best["total_s"]:
                best = row
        gpu = sum(best["phases"][k] for k in CALIB_GPU_PHASES)
        host = sum(v for k, v in best["phases"].items() if k not in CALIB_GPU_PHASES)
        out = {"workload": "%s, %d images x %d corners, poses from scratch (estimateInitialGrid), JSON -> report + image_error files"
                           % (workload, images, N_CORNERS),
               "observations": info["observations"], "json_megabytes": info["json_bytes"] / 1e6, "generate_and_write_json_s": t_write,
               **best, "gpu_phases_s": gpu, "host_phases_s": host,
               "dominant_phase": max(best["phases"].items(), key=lambda kv: kv[1])[0],
               "images_per_second_end_to_end": images / best["total_s"]}
        if cli:
            exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "calib")
            walls = []
            for _ in range(max(1, runs)):
                t0 = time.perf_counter()
                r = subprocess.run([exe, path], cwd=directory, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=600)
                walls.append(time.perf_counter() - t0)
                if r.returncode != 0:
                    raise RuntimeError("calib failed: %s" % r.stderr.decode("utf-8", "replace")[-500:])
            out["cli_wall_s"] = min(walls)   # process start, HIP initialisation and module load included
        return out
    finally:
        if own:
            shutil.rmtree(directory, ignore_errors=True)


def pose_init(model="eucm", images=10000, reps=5, device=0, keep_inputs=False):
    """f2: the 4-corner geometric pose of every image (host) and vg_refine_poses (vg_pose_lm_kernel: one independent LM per image,
    SoftLOneLoss(25), in one launch) from those poses at the initial intrinsics -- what estimateInitialGrid does per image
    (unified_calibration.cpp:1066-1158).  Kernel time = HIP events around the launch (vg_calibration_timings.refine_kernel_s of a
    front-end run carries the same clock); priced against the FP64 vector peak with
    (iterations + 1) x N x (EVAL_FLOPS + 2 x 7 x 8) flops per image (one [J | r] evaluation + 7 x 7 Gram per LM iteration)."""
    import ctypes
    import time

    from . import capi, synthetic
    from .calibration import refine_poses

    L = capi.load()
    d = synthetic.make_mono(model, images, 1)
    intr = d["init_intrinsics"]
    board, corners = d["board"], d["corners"]
    N = board.shape[0]
    idx = [0, synthetic.BOARD_COLS - 1, synthetic.BOARD_COLS * (synthetic.BOARD_ROWS - 1), N - 1]
    dp = ctypes.POINTER(ctypes.c_double)
    b4 = np.ascontiguousarray(board[idx])
    start = np.zeros((images, 6))
    t0 = time.perf_counter()
    for i in range(images):
        c4 = np.ascontiguousarray(corners[i, idx])
        capi.check(L.vg_initial_grid_pose(capi.MODELS[model], intr.ctypes.data_as(dp), b4.ctypes.data_as(dp), c4.ctypes.data_as(dp),
                                          start[i].ctypes.data_as(dp)))
    t_geo_py = time.perf_counter() - t0   # through ctypes, one call per image: an upper bound of the host loop in the library
    # the product route (the calibration front end, a host that owns a vg_problem): the corners are resident in the problem the
    # poses are for -- vg_dataset_refine_poses moves only the 6-vectors and the per-image results
    from . import CalibrationProblem

    pr = CalibrationProblem(device)
    cam = pr.add_camera(model, intr)
    seq = pr.add_transform(False, start)
    dsr = pr.add_dataset(cam, [(seq, 0)], board, corners)
    pr.finalize()
    best, it, kernel_s = None, None, None
    for _ in range(max(1, reps)):
        ks = []
        t0 = time.perf_counter()
        poses, it, cost, term = pr.refine_poses(dsr, start, kernel_seconds=ks)
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
        kernel_s = ks[0] if kernel_s is None or ks[0] < kernel_s else kernel_s
    pr.close()
    # the stand-alone entry with everything in host memory (vg_refine_poses): the same launch behind an upload of the corners
    best_host = None
    for _ in range(max(1, reps)):
        t0 = time.perf_counter()
        poses_h, it_h, _, _ = refine_poses(model, intr, board, corners, start, device=device)
        dt = time.perf_counter() - t0
        best_host = dt if best_host is None or dt < best_host else best_host
    assert np.array_equal(poses_h, poses) and np.array_equal(it_h, it)   # the same kernel on the same inputs
    flops = float(np.sum((it.astype(np.float64) + 1) * N * (EVAL_FLOPS[model] + 2 * 7 * 8)))
    out = {"workload": "%s, %d images x %d corners, start = 4-corner pose at the initial intrinsics" % (model, images, N),
           "kernel": "vg_pose_lm_kernel<%s>" % model, "kernel_ms": kernel_s * 1e3,
           "roofline": {"bound": "fp64", "achieved": flops / kernel_s / 1e12, "peak": FP64_PEAK / 1e12, "unit": "TFLOP/s",
                        "frac": flops / kernel_s / FP64_PEAK,
                        "note": "latency bound: every half-wave runs its image's whole LM (a dependent evaluate -> 32-lane sum -> 6 x 6 "
                                "Cholesky chain per iteration), a wave lives as long as its slower image"},
           "refine_call_ms": best * 1e3, "refine_call_route": "vg_dataset_refine_poses: corners resident in the problem; poses up, results back",
           "refine_call_over_kernel": best / kernel_s,
           "host_pointer_call_ms": best_host * 1e3, "host_pointer_route": "vg_refine_poses: %.1f MB of corners uploaded from pageable host memory per call" % (corners.nbytes / 1e6),
           "iterations_mean": float(it.mean()), "iterations_p99": float(np.percentile(it, 99)),
           "iterations_max": int(it.max()), "converged": int(np.sum(term <= 2)), "algorithmic_flops": flops,
           "max_pose_error_vs_generating": float(np.max(np.abs(poses - d["gt_poses"]))),
           "geometric_init_through_ctypes_ms": t_geo_py * 1e3}
    if keep_inputs:   # for the caller's CPU baseline leg (bench.py: the checker is only ever timed there, never from this package)
        out["_inputs"] = {"model": model, "intrinsics": intr, "board": board, "corners": corners, "start": start}
    return out
