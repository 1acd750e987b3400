#!/usr/bin/env python3
"""bench.py -- corner residual + Jacobian evaluations per second on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: one rank per GPU.  Launched bare, bench.py re-executes itself through `python -m torch.distributed.run
   --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`; launched by torch.distributed.run it reads RANK / WORLD_SIZE.)

Workload (BASELINE.json metric: "corner residual+Jacobian evals/sec ... (EUCM 10k imgs)"):
  synthetic EUCM mono, 10 000 images x 96 corners (8 x 12 board) PER GPU, chain [xiCamBoard DIRECT],
  evaluated at the perturbed point of SURVEY 8(d); all Jacobian blocks requested (6 intrinsics + 6 pose).
One step = one full evaluation of the hot path at the current parameters (vg_problem_prepare + vg_dataset_evaluate):
  the transform chain of every image + residual pair + 2 x 12 Jacobian entries per (image, corner), written to HBM in
  the Ceres block layout.  For this single-member DIRECT chain the emit kernel derives the per-image frames itself,
  so the step is ONE launch; multi-member chains and the Gram kernels use the separate chain-prep kernel.
Inputs are resident in HBM before the timed region.  Multi-GPU: images are sharded over ranks
(weak scaling, no data-path collective in this pass).  The normal-equation build that needs the exchange step is
reported under "jtj" (same weak-scaled set) and under "sharded_mei" (BASELINE.json config 4: Mei, 10 000 images x 96
corners in TOTAL, split over the N ranks -- strong scaling; one iteration = fused J^T J + ONE packed in-place RCCL
all-reduce of [H | g | cost | n_failed] on the device buffer through the native vg_comm entry), and the full LM solves
of sharded problems under "sharded_solve" (Mei 10 k and EUCM 100 k images in total: two collectives per iteration on the
critical path).

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# floating-point operations of ONE evaluation of the restatement per observation, chain of one member (point transform,
# projection, d(u,v)/dX, d(u,v)/d(intrinsics), residual, the 2 x 6 pose rows); counted in DESIGN.md section 5.3
EVAL_FLOPS = {"eucm": 200, "ucm": 197, "mei": 346}
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--images", type=int, default=10000, help="images per GPU")
    ap.add_argument("--model", default="eucm", choices=["eucm", "ucm", "mei"])
    ap.add_argument("--sharded-images", type=int, default=10000, help="total images of the sharded Mei case (config 4)")
    ap.add_argument("--sharded-solve-images", type=int, default=100000,
                    help="total images of the large sharded LM solve (0: skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary-configs", action="store_true",
                    help="skip the per-config sections (BASELINE configs 3 and 5, the beyond-L3 stream)")
    ap.add_argument("--headline-kernels-only", action="store_true",
                    help="profiling passes (tools/gpu_check.sh): skip the sections that launch the headline's emit kernel at OTHER sizes "
                         "(emit_sweep, the chunked host route of pcie_inclusive) and the calib / pose-init sections, so that the "
                         "kernel's average in a rocprofv3 trace of this command is the headline launch's")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="CPU-baseline budget per leg")
    return ap.parse_args()


def _cgroup_cpu_max():
    """the container's CPU quota as the kernel states it ("max 100000" = none), or None"""
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            return open(f).read().strip()
        except OSError:
            continue
    return None


def cpu_baseline(d, model, budget_s):
    """The oracle (a C port of the reference's arithmetic, oracle/vg_oracle.c) timed on this box's host
    cores over the SAME workload: whole passes over the 10 k-image set, residual + all Jacobian blocks,
    repeated for ~budget_s seconds per leg.  This is the only place bench.py touches oracle/."""
    import numpy as np

    from oracle import vgo

    n, N = d["corners"].shape[0], d["board"].shape[0]
    K = d["init_intrinsics"].size
    pv = np.concatenate([d["init_intrinsics"], d["init_poses"].ravel()])
    m = vgo.MODELS[model]
    out = (np.empty((n, 2 * N)), np.empty((n, 2 * N, K)), [np.empty((n, 2 * N, 6))])
    seq = np.arange(n)

    def leg(threads):
        vgo.eval_dataset(m, [0], d["board"], d["corners"], pv, 0, [K], [6], seq, threads=threads, out=out)  # warm
        passes, t0 = 0, time.perf_counter()
        while True:
            vgo.eval_dataset(m, [0], d["board"], d["corners"], pv, 0, [K], [6], seq, threads=threads, out=out)
            passes += 1
            el = time.perf_counter() - t0
            if el >= budget_s:
                return passes, el, passes * n * N / el

    omp_max = vgo.max_threads()
    try:
        allowed = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        allowed = os.cpu_count() or omp_max
    p1, t1, v1 = leg(1)
    # "all host cores"  is not one number on the boxes of this pool: the container sees 256 logical CPUs but is
    # scheduled on far fewer (tools/exp/cpu_scaling.py: linear to 16 threads, 3.7e8 evals/s, and FALLING beyond -- 8e7 at 128
    # threads, 7e6 at 256: oversubscribed OpenMP teams).  So the baseline is the BEST thread count of a short sweep (0.5 s per
    # count, powers of two up to every CPU this process may run on), then timed for the full budget.
    def probe(threads):
        t0, passes = time.perf_counter(), 0
        while time.perf_counter() - t0 < 0.5:
            vgo.eval_dataset(m, [0], d["board"], d["corners"], pv, 0, [K], [6], seq, threads=threads, out=out)
            passes += 1
        return passes * n * N / (time.perf_counter() - t0)

    counts = sorted({c for c in (2, 4, 8, 16, 32, 64, 128, 256, 512, omp_max, allowed) if 2 <= c <= max(allowed, omp_max)})
    sweep = {c: probe(c) for c in counts} if counts else {1: v1}
    cores = max(sweep, key=lambda k: sweep[k])
    pc, tc, vc = leg(cores)
    legs = dict(sweep)
    legs[cores] = max(vc, sweep[cores])
    if v1 > vc:   # a box that gives this process one core's worth of time
        cores, pc, tc, vc = 1, p1, t1, v1
    # J^T J build on the CPU: evaluate (above) + per-image Gram of [J | r] + sum, all cores
    gout = (np.empty((n, K + 7, K + 7)), np.empty((K + 7, K + 7)))
    vgo.dataset_gram(out[0], out[1], out[2], threads=cores, out=gout)
    reps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < max(1.0, budget_s / 4):
        vgo.dataset_gram(out[0], out[1], out[2], threads=cores, out=gout)
        reps += 1
    gram_ms = (time.perf_counter() - t0) / reps * 1e3
    cpu_model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = None
    # physical cores of the host (sockets x cores per socket from /proc/cpuinfo): what "all host cores" would be without the
    # container's CPU quota.  The extrapolation is LINEAR in the measured single-thread rate -- an upper bound for this port (no
    # memory-bandwidth or turbo derating), stated so that the quota-bound ratio is never quoted alone.
    phys = None
    try:
        ids = set()
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":", 1)[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    ids.add((pid, cid))
                pid = cid = None
        phys = len(ids) or None
    except OSError:
        phys = None
    quota = None
    try:
        q, per = (_cgroup_cpu_max() or "").split()[:2]
        quota = None if q == "max" else float(q) / float(per)
    except (ValueError, AttributeError):
        quota = None
    return {"value": vc, "unit": "evals/s", "cores": cores, "kind": "port",
            "cgroup_cpu_quota_cpus": quota, "physical_cores": phys,
            "extrapolated_all_physical_cores": {"value": v1 * phys, "unit": "evals/s", "cores": phys,
                                                "how": "single_thread_value x physical cores, linear (an upper bound: no memory-bandwidth or clock derating)"} if phys else None,
            # the count actually used next to what the box offers -- logical CPUs (hardware_concurrency), the CPUs
            # this process may run on, OpenMP's own maximum; `cores` = the threads of the leg reported as `value`
            "hardware_concurrency": os.cpu_count(), "cpus_allowed": affinity, "omp_max_threads": omp_max,
            "thread_sweep_evals_per_s": {str(k): v for k, v in sorted(legs.items())}, "cgroup_cpu_max": _cgroup_cpu_max(),
            "sample": "%d passes (%.1f s) over the same %d-image x %d-corner set with %d OpenMP threads; "
                      "single thread: %d passes (%.1f s)" % (pc, tc, n, N, cores, p1, t1),
            "single_thread_value": v1, "cpu_model": cpu_model,
            "jtj_ms_per_iter": n * N / vc * 1e3 + gram_ms, "jtj_gram_only_ms": gram_ms}


def pose_init_cpu_baseline(inp, sample=48):
    """CPU baseline of the pose initialisation (f2): the same per-image problem -- one GenericProjectionJac block, chain {DIRECT},
    intrinsics constant, from the same 4-corner start -- solved one image at a time by scipy's trust-region least squares on the
    oracle's residuals and Jacobian rows (the checker, timed here as the baseline only; tests/test_gpu_refine.py holds the GPU
    result to this solve).  The reference does the same with one Ceres problem per image, sequentially
    (unified_calibration.cpp:1137-1155)."""
    from scipy.optimize import least_squares

    from oracle import vgo

    m = vgo.MODELS[inp["model"]]
    board, corners, intr, start = inp["board"], inp["corners"], inp["intrinsics"], inp["start"]
    sample = min(sample, corners.shape[0])
    t0 = time.perf_counter()
    for b in range(sample):
        least_squares(lambda x: vgo.eval_block(m, [0], board, corners[b], [intr, x], want_jac=False)[0], start[b],
                      jac=lambda x: vgo.eval_block(m, [0], board, corners[b], [intr, x])[1][1], method="trf", x_scale="jac",
                      xtol=1e-8, ftol=1e-6, gtol=1e-10, max_nfev=500)
    dt = time.perf_counter() - t0
    return {"kind": "port", "what": "scipy TRF on the oracle's rows, one image at a time", "cores": 1, "sample": "%d images" % sample,
            "value": sample / dt, "unit": "images/s"}


def respawn(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, one per GPU, and pass rank 0's JSON
    line through.  (`--gpus 1` never comes here: it runs in this process, exactly as before.)"""
    import socket
    import subprocess

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    a = parse()
    if a.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(respawn(a.gpus))
    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    a.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    # VG_BENCH_BACKEND=gloo lets the multi-rank control flow be rehearsed with several ranks on ONE GPU
    # (RCCL refuses two ranks per device); the driver's runs use the default, nccl = RCCL over xGMI.
    backend = os.environ.get("VG_BENCH_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(local_rank)
    dist = None
    # VG_BENCH_FORCE_DIST=1 (under a launcher): take the multi-rank code path -- process group, native communicator,
    # collectives -- with a world of ONE rank, the only way to run it on a one-GPU box
    if world > 1 or (os.environ.get("VG_BENCH_FORCE_DIST") and "RANK" in os.environ):
        import torch.distributed as dist_

        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    def all_reduce_(t, op=None):
        # RCCL reduces device tensors in place; the gloo rehearsal path goes through host memory
        kw = {} if op is None else {"op": op}
        if backend == "nccl":
            dist.all_reduce(t, **kw)
        else:
            h = t.cpu()
            dist.all_reduce(h, **kw)
            t.copy_(h)

    from visgeom_amd import CalibrationProblem, synthetic

    # the native RCCL communicator (vg_comm): the normal-equation blocks are reduced through it, on the device buffers.
    # torch.distributed only carries the unique id and the fences; with the gloo rehearsal backend (several ranks on ONE
    # GPU, which RCCL refuses) the blocks go through torch instead.
    comm = None
    if dist is not None and backend == "nccl":
        from visgeom_amd import distributed as vdist_

        # created in a helper thread with a deadline and agreed on by all ranks: if the native communicator cannot be
        # set up on this node, every rank falls back to torch.distributed for the small sums instead of hanging the bench
        import threading

        box = {}

        uid = [vdist_.Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)           # the id travels over torch.distributed, in the main thread

        def _make():
            try:
                torch.cuda.set_device(local_rank)        # the current device is a per-thread setting
                box["comm"] = vdist_.Comm(uid[0], world, rank, local_rank)   # ncclCommInitRank: collective over all ranks
            except Exception as e:  # noqa: BLE001
                box["error"] = repr(e)

        # the agreement runs over a gloo side group on HOST tensors: every rank takes it unconditionally, also one whose
        # helper thread is still inside ncclCommInitRank (that thread is abandoned; a device collective next to it could
        # not be matched, a host one can)
        side = dist.new_group(backend="gloo")
        th = threading.Thread(target=_make, daemon=True)
        th.start()
        th.join(float(os.environ.get("VG_BENCH_COMM_TIMEOUT", "120")))
        ok = torch.tensor([1 if ("comm" in box and not th.is_alive()) else 0], dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=side)
        if int(ok.item()) == 1 and "comm" in box:
            comm = box["comm"]
            if rank == 0:
                print("[bench] native RCCL communicator: world size %d" % comm.n_ranks, file=sys.stderr)
        else:
            print("[bench] rank %d: native RCCL communicator unavailable (%s): sums go through torch.distributed" %
                  (rank, box.get("error", "timeout")), file=sys.stderr)

    def sum_over_ranks_(t):
        if comm is not None:
            comm.allreduce_sum(t)
        elif dist is not None:
            all_reduce_(t)

    def allreduce_latency_us(sizes, calls=1000):
        """Isolated latency of the path's collective: an in-place sum of `size` doubles on a device buffer through the route the
        solver uses (vg_comm_allreduce_sum = ncclAllReduce when the native communicator exists, torch.distributed otherwise),
        every call bracketed by a stream synchronisation; median / p10 / p90 over `calls` calls per size, in microseconds.
        One rank without a communicator has no collective: None."""
        if comm is None and dist is None:
            return None
        res_ = {}
        for size in sizes:
            buf = torch.zeros(int(size), dtype=torch.float64, device="cuda")
            for _ in range(20):
                sum_over_ranks_(buf)
            torch.cuda.synchronize()
            ts = np.empty(calls)
            for i in range(calls):
                t0 = time.perf_counter()
                sum_over_ranks_(buf)
                torch.cuda.synchronize()
                ts[i] = time.perf_counter() - t0
            tt = torch.tensor([float(np.median(ts)), float(np.percentile(ts, 10)), float(np.percentile(ts, 90))], dtype=torch.float64, device="cuda")
            if dist is not None:
                all_reduce_(tt, op=dist.ReduceOp.MAX)
            m = tt.cpu().numpy() * 1e6
            res_[str(int(size))] = {"doubles": int(size), "bytes": int(size) * 8, "median_us": float(m[0]), "p10_us": float(m[1]), "p90_us": float(m[2])}
        return res_

    def solver_message_sizes(widths, G):
        """doubles in the two in-place all-reduces of one LM iteration (vg_solver_impl.hpp): the summed Gram blocks of all datasets
        (n_ds x Wmax^2) + 5 step scalars per evaluation; the Schur complement (G + 1)^2 + the count of bad pose blocks per linear solve"""
        return len(widths) * max(widths) ** 2 + 5, (G + 1) ** 2 + 1

    cfg_index = 1  # the metric's configuration: EUCM mono, 10 k images x 96 corners
    d = synthetic.make_mono(a.model, a.images, cfg_index, first_image=rank * a.images)
    n_img, N = d["corners"].shape[0], d["board"].shape[0]
    K = d["init_intrinsics"].size
    n_obs = n_img * N

    p = CalibrationProblem(local_rank)
    cam = p.add_camera(a.model, d["init_intrinsics"])
    seq = p.add_transform(False, d["init_poses"])
    ds = p.add_dataset(cam, [(seq, 0)], d["board"], d["corners"])
    p.finalize()
    res, ji, jm = p.alloc_outputs(ds)

    def step():
        p.prepare()
        p.evaluate_dataset(ds, res, ji, jm)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    import gc

    gc.collect()
    gc.disable()   # no collector pause between two launches of the timed loop
    # bring the device to its steady clocks before the contract's W warm-up steps: the first ~300 evaluations after an
    # idle period of a few milliseconds run up to 5 % slower (tools/exp/step_overhead_probe.py), whatever W and K the
    # caller picked; nothing but the fence may sit between this and the timed loop
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < 0.1:
        for _ in range(50):
            step()
        torch.cuda.synchronize()
    for _ in range(a.warmup):
        step()
    fence()
    # The K steps are bracketed by barrier + synchronize on both sides; the clock of a rank stops when ITS device is done
    # (the closing barrier follows at once, outside the interval) and the job's time is the MAX over ranks: the time from
    # the common start to the last rank's finish, without the latency of the closing collective itself -- at the K = 20
    # the driver uses, an NCCL barrier inside the interval would be a fifth of it.
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    fence()
    gc.enable()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        all_reduce_(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert p.failed_count(ds) == 0
    value = world * n_obs * a.steps / elapsed
    from visgeom_amd import capi

    single_launch = capi.load().vg_dataset_single_launch(p._h, ds) == 1
    # The contract's line, complete from here on; the sections below add their objects as they finish.
    out = {
        "metric": "corner residual+Jacobian evals/sec",
        "value": value,
        "unit": "evals/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": elapsed / a.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "library": ("hooks build: " if capi.has_debug_hooks() else "production build (no debug hooks): ") + os.path.relpath(capi.lib_path(), ROOT),
        "config": {"workload": "%s mono, %d images x %d corners (8x12 board) per GPU, chain [xiCamBoard DIRECT], "
                               "residual + all Jacobian blocks (K=%d intrinsics + 6 pose) emitted to HBM in Ceres "
                               "block layout; step = vg_problem_prepare + vg_dataset_evaluate (%s)" % (a.model.upper(), n_img, N, K, "one launch: the emit kernel walks the single-member chain itself" if single_launch else "chain-prep kernel + emit kernel"),
                   "images_per_gpu": n_img, "corners_per_image": N, "camera_model": a.model, "chain": ["DIRECT"],
                   "seed": int(d["seed"]), "sharding": "images sharded over ranks, no data-path collective"},
    }
    # The secondary sections with several ranks contain collectives that have never run on more than one GPU in this
    # project's own sessions.  A deadline keeps the headline: if they are not through in time, rank 0
    # prints the line with what is finished and every rank leaves.
    import threading

    secondary_done = threading.Event()

    def _deadline():
        limit = float(os.environ.get("VG_BENCH_SECONDARY_TIMEOUT", "420" if world > 1 else "1500"))
        if secondary_done.wait(limit):
            return
        line = None
        for _ in range(20):
            try:
                snap = dict(out)
                snap["secondary_sections"] = "not finished within %.0f s: %s missing" % (
                    limit, ", ".join(k for k in ("roofline", "jtj", "sharded_mei", "sharded_solve", "solve", "pcie_inclusive", "config3_stereo", "config5_rig", "eucm_100k", "calib_e2e", "pose_init") if k not in snap))
                line = json.dumps(snap)
                break
            except RuntimeError:   # the main thread added a key meanwhile
                time.sleep(0.01)
        if rank == 0 and line is not None:
            print(line, flush=True)
        os._exit(0)

    threading.Thread(target=_deadline, daemon=True).start()

    # ---- roofline of the dominant kernel (emit): HIP events on the launch stream, K launches ----
    bytes_per_obs = 16 + 16 + 16 * (K + 6)  # obs read + residual write + Jacobian rows  
    stream = torch.cuda.current_stream()
    p.prepare()
    torch.cuda.synchronize()
    n_roof = max(a.steps, 300)  # the kernel's average needs a few hundred launches whatever K the caller asked for
    for _ in range(100):        # steady clocks again after the host-side pause between the sections
        p.evaluate_dataset(ds, res, ji, jm)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_roof)]
    for e0, e1 in ev:
        e0.record(stream)
        p.evaluate_dataset(ds, res, ji, jm)
        e1.record(stream)
    torch.cuda.synchronize()
    per_launch_ms = np.array([e0.elapsed_time(e1) for e0, e1 in ev])
    # back-to-back launches bracketed once (includes the ~1.5 us inter-kernel gap)
    b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    b0.record(stream)
    for _ in range(n_roof):
        p.evaluate_dataset(ds, res, ji, jm)
    b1.record(stream)
    torch.cuda.synchronize()
    b2b_ms = b0.elapsed_time(b1) / n_roof
    # Average launch duration = the back-to-back figure (it is what rocprofv3 --kernel-trace reports for this
    # kernel: 33.85 us vs 33.77 us here in profiles/r01b_*); an event pair around every single launch adds
    # ~2 us of marker overhead per launch and is kept only as a cross-check.
    emit_ms = b2b_ms
    achieved = bytes_per_obs * n_obs / (emit_ms * 1e-3) / 1e9
    # HBM traffic per launch cannot be measured from inside this process (PMC counters need rocprofv3): it is read from the
    # committed summary of the separate --pmc passes of this same command (tools/gpu_check.sh, tools/prof_summary.py)
    traffic, traffic_source = None, None
    prof = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(prof):
        try:
            entry = json.load(open(prof)).get("%s_%d" % (a.model, a.images), {})
            traffic = entry.get("hbm_bytes_per_launch")
            if traffic is not None:
                traffic_source = "profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, tag %s" % entry.get("tag")
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                # single-member DIRECT chain, output within reach of the Infinity Cache: the emit kernel derives the
                # frames itself and the step is this ONE launch; larger sets run chain prep + emit
                "kernel": "vg_emit_kernel<%s,jac,frames-in-LDS%s>" % (a.model, ",inline-chain" if single_launch else ""),
                # what the "hbm" label means at this size: priced against the 8 TB/s HBM peak as the contract asks,
                # but an output that fits the Infinity Cache is absorbed there -- the DRAM-streaming figures are emit_sweep's >= 100 k rows
                "regime": ("within the 256 MiB Infinity Cache (output %.1f MB): cache-assisted, not a DRAM-streaming figure" if 16 * (K + 7) * n_obs < 256 * 2 ** 20
                           else "beyond the 256 MiB Infinity Cache (output %.1f MB)") % (16 * (K + 7) * n_obs / 1e6),
                "launches_per_step": 1 if single_launch else 2,
                "algorithmic_bytes_per_launch": bytes_per_obs * n_obs, "bytes_per_obs": bytes_per_obs,
                "avg_launch_ms": emit_ms, "event_pair_per_launch_ms": float(np.mean(per_launch_ms)),
                "event_pair_median_ms": float(np.median(per_launch_ms))}
    out["roofline"] = roofline

    # ---- PCIe-inclusive rate of the same step when the caller wants the rows in HOST memory (the Ceres
    # EvaluationCallback route): kernels + D2H of residuals and all Jacobian blocks into
    # pinned buffers.  Reported for DESIGN.md; it is never `value`.
    pcie = None
    try:
        if a.headline_kernels_only:
            raise RuntimeError("skipped (--headline-kernels-only)")
        from visgeom_amd import capi as _capi
        import ctypes as _ct

        host_bytes = (16 + 16 * (K + 6)) * n_obs
        reps = 24

        def to_host_times(h_res, h_ji, h_jm):
            p.prepare()
            p.evaluate_dataset_to_host(ds, h_res, h_ji, h_jm)     # first call: staging blocks allocated and touched
            ts = []
            for _ in range(reps):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                p.prepare()
                p.evaluate_dataset_to_host(ds, h_res, h_ji, h_jm)
                ts.append(time.perf_counter() - t0)
            return np.array(ts)

        def stats(ts):
            return {"min_ms": float(ts.min() * 1e3), "median_ms": float(np.median(ts) * 1e3), "max_ms": float(ts.max() * 1e3),
                    "median_host_GBps": host_bytes / float(np.median(ts)) / 1e9, "evals_per_s_median": n_obs / float(np.median(ts))}

        # (a) the bus on this box: plain blocking hipMemcpy of the same byte count into hipHostMalloc memory
        sec = np.zeros(reps)
        _capi.check(_capi.load().vg_calib_d2h_copies(local_rank, host_bytes, reps, sec.ctypes.data_as(_ct.POINTER(_ct.c_double))))
        ceiling = {"bytes": host_bytes, "min_ms": float(sec.min() * 1e3), "median_ms": float(np.median(sec) * 1e3), "max_ms": float(sec.max() * 1e3),
                   "median_GBps": host_bytes / float(np.median(sec)) / 1e9}
        # (b) pinned destinations (torch.pin_memory = hipHostMalloc): chunks straight from the copy engine
        h_res = torch.empty(res.shape, dtype=torch.float64).pin_memory()
        h_ji = torch.empty(ji.shape, dtype=torch.float64).pin_memory()
        h_jm = [torch.empty(t.shape, dtype=torch.float64).pin_memory() for t in jm]
        t_pinned = to_host_times(h_res, h_ji, h_jm)
        assert torch.equal(h_res, res.cpu()) and torch.equal(h_jm[0], jm[0].cpu())
        del h_res, h_ji, h_jm
        # (c) ordinary (pageable) destinations, what Ceres allocates: library-owned pinned staging + the host's threads
        n_res, n_ji = np.empty(tuple(res.shape)), np.empty(tuple(ji.shape))
        n_jm = [np.empty(tuple(t.shape)) for t in jm]
        n_res[:] = 0
        n_ji[:] = 0
        for t in n_jm:
            t[:] = 0
        t_pageable = to_host_times(n_res, n_ji, n_jm)
        assert np.array_equal(n_res, res.cpu().numpy()) and np.array_equal(n_ji, ji.cpu().numpy()) and np.array_equal(n_jm[0], jm[0].cpu().numpy())
        del n_res, n_ji, n_jm
        pcie = {"host_bytes_per_step": host_bytes, "repetitions": reps, "d2h_ceiling_same_box": ceiling,
                "pinned_destination": stats(t_pinned), "pageable_destination": stats(t_pageable),
                "pinned_over_ceiling": float(np.median(t_pinned) / np.median(sec)), "pageable_over_ceiling": float(np.median(t_pageable) / np.median(sec)),
                # kept from earlier rounds' lines: the pinned route's median
                "ms_per_step": float(np.median(t_pinned) * 1e3), "evals_per_s": n_obs / float(np.median(t_pinned)),
                "host_GBps": host_bytes / float(np.median(t_pinned)) / 1e9}
    except Exception as e:
        pcie = {"error": repr(e)}
    out["pcie_inclusive"] = pcie

    # ---- measured streaming rates on this box, same 16 B/lane pattern (context for the fraction) ----
    from visgeom_amd import capi
    import ctypes

    L = capi.load()
    nd = 64 * 1024 * 1024  # 512 MiB of doubles: past the 256 MiB Infinity Cache
    buf = torch.empty(nd, dtype=torch.float64, device="cuda")
    buf2 = torch.empty(nd, dtype=torch.float64, device="cuda")
    sp = ctypes.c_void_p(stream.cuda_stream)

    def rate(fn, nbytes, reps=10):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
        torch.cuda.synchronize()
        return nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9

    write_gbs = rate(lambda: capi.check(L.vg_calib_stream_write(sp, ctypes.c_void_p(buf.data_ptr()), nd, 1.0)), nd * 8)
    copy_gbs = rate(lambda: capi.check(L.vg_calib_stream_copy(sp, ctypes.c_void_p(buf2.data_ptr()),
                                                              ctypes.c_void_p(buf.data_ptr()), nd)), nd * 16)
    del buf, buf2
    roofline["measured_stream_write_GBps"] = write_gbs
    roofline["measured_stream_copy_GBps"] = copy_gbs

    # ---- J^T J / J^T r build, ms per iteration (second half of BASELINE.json's metric) ----
    # fused:    chain prep + evaluate-and-contract kernel (J stays on chip, FP64 vector pipe) + fixed-order sum
    # two-pass: chain prep + emit (J to HBM) + second pass over the materialised rows + sum
    # N > 1: every iteration ends with ONE all-reduce (RCCL) of the W x W summed block [J^T J | J^T r | r^T r].
    gram, gsum = p.alloc_gram(ds)

    def finish():
        p.gram_sum(ds, gram, gsum)
        sum_over_ranks_(gsum)

    def it_fused():
        # narrow row blocks (EUCM / UCM mono): chain walk + evaluate + Gram in ONE launch on the FP64 vector pipe, the
        # workgroup partial sums added by ONE more launch; then the all-reduce of the W x W block
        p.prepare()
        p.gram_fused_sum(ds, gram, gsum)
        sum_over_ranks_(gsum)

    def it_two_pass():
        p.prepare()
        p.evaluate_dataset(ds, res, ji, jm)
        p.gram_from_rows(ds, res, ji, jm, gram)
        finish()

    def it_second_pass_only():
        p.gram_from_rows(ds, res, ji, jm, gram)
        finish()

    def wall_ms(fn, n):
        n = max(n, 200)  # secondary sections: enough iterations that the bracketing itself is noise whatever K was asked
        for _ in range(max(3, n // 10)):
            fn()
        fence()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        fence()
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device="cuda")
            all_reduce_(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el / n * 1e3

    W = p.gram_width(ds)
    jtj = {"unit": "ms/iter", "gram_width": W, "images_per_gpu": n_img, "allreduce": dist is not None,
           "rccl_world_size": comm.n_ranks if comm is not None else None,
           "collective": None if dist is None else ("vg_comm_allreduce_sum (RCCL, in place on the device block)" if comm is not None
                                                    else "torch.distributed " + backend),
           # secondary legs: best of three timed runs (a single run picked up a host hiccup once: 0.12 vs 0.05 ms);
           # the headline `value` above stays ONE timed run of exactly K steps, as the contract says
           "fused_ms_per_iter": min(wall_ms(it_fused, a.steps) for _ in range(3)),
           "two_pass_ms_per_iter": min(wall_ms(it_two_pass, a.steps) for _ in range(3)),
           "second_pass_only_ms": min(wall_ms(it_second_pass_only, a.steps) for _ in range(3)),
           "fused_algorithmic_bytes_per_obs": 16 + 8.0 * W * W / N,
           "two_pass_read_bytes_per_obs": 16 + 16 * (K + 6)}
    # roofline of the fused evaluate + Gram kernel: FP64 vector peak ("report it against the FP64 vector
    # peak").  Algorithmic flops per observation = Gram 2 (P+1)(P+2) (two rows, upper triangle incl. diagonal, one
    # multiply-add each; P = K + 6) + one evaluation of the restatement (EVAL_FLOPS, counted operation by operation in
    # DESIGN.md section 5.3; divisions and square roots count 1).  Kernel time: HIP events around back-to-back launches.
    p.prepare()
    n_gram = max(a.steps, 300)
    for _ in range(100):
        p.gram_fused(ds, gram)
    g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g0.record(stream)
    for _ in range(n_gram):
        p.gram_fused(ds, gram)
    g1.record(stream)
    torch.cuda.synchronize()
    gram_ms = g0.elapsed_time(g1) / n_gram
    flops_per_obs = 2 * (K + 7) * (K + 8) + EVAL_FLOPS[a.model]
    FP64_PEAK = 78.6  # TFLOP/s, MI355X FP64 vector = FP64 matrix
    ach = flops_per_obs * n_obs / (gram_ms * 1e-3) / 1e12
    jtj["roofline"] = {"bound": "fp64", "kernel": "fused evaluate + Gram (vg_dataset_gram_fused)", "flops_per_obs": flops_per_obs,
                       "gram_flops_per_obs": 2 * (K + 7) * (K + 8), "eval_flops_per_obs": EVAL_FLOPS[a.model],
                       "avg_launch_ms": gram_ms, "achieved": ach, "peak": FP64_PEAK, "unit": "TFLOP/s",
                       "frac": ach / FP64_PEAK}
    # what the FP64 pipe delivers on THIS box at the Gram kernels' occupancy (dependent-FMA chains, no memory): the same role as
    # measured_stream_write_GBps beside the HBM peak -- `frac` stays priced against the guide's 78.6 TFLOP/s
    try:
        fl = ctypes.c_int64(0)
        scratch = torch.zeros(4096, dtype=torch.float64, device="cuda")
        fma = lambda: capi.check(L.vg_calib_fp64_fma(sp, ctypes.c_void_p(scratch.data_ptr()), 20000, ctypes.byref(fl)))
        for _ in range(3):
            fma()
        torch.cuda.synchronize()
        f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        f0.record(stream)
        for _ in range(10):
            fma()
        f1.record(stream)
        torch.cuda.synchronize()
        fma_tf = fl.value * 10 / (f0.elapsed_time(f1) * 1e-3) / 1e12
        jtj["roofline"]["measured_fp64_fma_TFLOPs"] = fma_tf
        jtj["roofline"]["frac_of_measured_fma_rate"] = ach / fma_tf
    except Exception as e:  # a measurement helper must not take the line down
        jtj["roofline"]["measured_fp64_fma_TFLOPs"] = None
        jtj["roofline"]["measured_fp64_fma_error"] = str(e)[:200]

    # ---- BASELINE.json config 4, the north-star multi-GPU case: Mei (K = 10), 10 000 images x 96 corners IN TOTAL,
    # images sharded over the ranks (strong scaling).  One LM-style iteration = chain prep + fused evaluate + Gram of
    # this rank's images + fixed-order sum + ONE all-reduce of H (W x W: J^T J, J^T r, the cost and with it the
    # number of failed corners) on the device buffer.  Time = max over ranks between fences.
    sharded = None
    try:
        from visgeom_amd import distributed as vdist

        n_total = a.sharded_images
        lo, hi = vdist.shard_range(n_total, rank, world)
        dm = synthetic.make_mono("mei", hi - lo, 4, first_image=lo)
        pm = CalibrationProblem(local_rank)
        cm = pm.add_camera("mei", dm["init_intrinsics"])
        sm = pm.add_transform(False, dm["init_poses"])
        dsm = pm.add_dataset(cm, [(sm, 0)], dm["board"], dm["corners"])
        pm.finalize()
        gm, _ = pm.alloc_gram(dsm)
        Wm = pm.gram_width(dsm)
        # [H]: J^T J, J^T r and r^T r; a failed projection adds 2e30 to r^T r (two residuals of 1e15), so the number of
        # failed corners of ALL ranks is read off the reduced r^T r -- it travels in the same W*W doubles
        pack = torch.zeros(Wm * Wm, dtype=torch.float64, device="cuda")

        def it_sharded():
            pm.prepare()
            pm.gram_fused_sum(dsm, gm, pack)
            sum_over_ranks_(pack)

        def it_sharded_compute_only():
            pm.prepare()
            pm.gram_fused_sum(dsm, gm, pack)

        ms = min(wall_ms(it_sharded, a.steps) for _ in range(3))
        ms_compute = min(wall_ms(it_sharded_compute_only, a.steps) for _ in range(3))
        lat = allreduce_latency_us([Wm * Wm])
        sharded = {"workload": "Mei mono, %d images x %d corners in total, images sharded over %d rank(s)" % (n_total, N, world),
                   "scaling": "strong", "images_total": n_total, "images_this_rank": hi - lo, "n_ranks": world,
                   "rccl_world_size": comm.n_ranks if comm is not None else None,
                   "collective": "none (one rank)" if dist is None else
                                 ("one vg_comm_allreduce_sum per iteration, %d doubles, in place on the device buffer" % pack.numel()
                                  if comm is not None else "torch.distributed " + backend),
                   "ms_per_iter": ms, "evals_per_s": n_total * N / (ms * 1e-3), "gram_width": Wm,
                   # where an iteration goes: the same loop without the collective (max over ranks), the rest, and the isolated latency of
                   # an all-reduce of exactly this message
                   "per_iteration": {"compute_ms": ms_compute, "collective_ms": max(ms - ms_compute, 0.0), "message_doubles": Wm * Wm},
                   "allreduce_us": lat,
                   "fused_gram_flops_per_obs": 2 * Wm * (Wm + 1) + EVAL_FLOPS["mei"],
                   "fused_gram_TFLOPs_incl_sum_and_collective": (2 * Wm * (Wm + 1) + EVAL_FLOPS["mei"]) * n_total * N / (ms * 1e-3) / 1e12,
                   "cost": float(pack[Wm * Wm - 1].item()) * 0.5,
                   "n_failed": float(torch.floor(pack[Wm * Wm - 1] / 2e30 + 0.5).item())}
        pm.close()
    except Exception as e:  # never take the headline down
        sharded = {"error": repr(e)}
    out["jtj"] = jtj
    out["sharded_mei"] = sharded

    # ---- full LM loop on the same set (GPU Gram + Schur, host Cholesky of the 6 x 6 reduced system); with N > 1
    # the images stay sharded and every iteration sums the small normal-equation blocks over ranks (RCCL) ----
    solve = None
    try:
        from visgeom_amd import distributed as vdist

        # the first solve of a process loads the solver's kernels and allocates the library's cached work blocks; a
        # calibration service solves again and again, so the warm figure is reported and the cold one kept beside it
        runs = []
        for rep in range(3):
            if rep:
                ps.close()
            ps = CalibrationProblem(local_rank)
            cs = ps.add_camera(a.model, d["init_intrinsics"])
            ss = ps.add_transform(False, d["init_poses"])
            ps.add_dataset(cs, [(ss, 0)], d["board"], d["corners"])
            ps.finalize()
            fence()
            summ = ps.solve(comm=comm, allreduce=vdist.make_allreduce() if (dist is not None and comm is None) else None,
                            max_num_iterations=50)
            fence()
            runs.append(summ["total_seconds"] * 1e3)
        xs = ps.get_parameters()
        n_eval_w, n_schur_w = solver_message_sizes([K + 7], K)
        lat_w = allreduce_latency_us([n_eval_w, n_schur_w], calls=500)
        solve = {"rccl_world_size": comm.n_ranks if comm is not None else None, "allreduce_us": lat_w,
                 "messages_doubles": {"evaluation": n_eval_w, "schur": n_schur_w},
                 "iterations": summ["num_iterations"], "successful_steps": summ["num_successful_steps"],
                 "termination": summ["termination"], "initial_cost": summ["initial_cost"], "final_cost": summ["final_cost"],
                 "total_ms": summ["total_seconds"] * 1e3, "first_solve_of_the_process_ms": runs[0], "runs_ms": runs,
                 "ms_per_iteration": summ["total_seconds"] * 1e3 / max(1, summ["num_iterations"]),
                 "setup_ms": summ["host_seconds"] * 1e3, "iterations_ms": summ["evaluate_seconds"] * 1e3, "global_columns": summ["num_global_columns"],
                 "pose_blocks_per_gpu": summ["num_pose_blocks"],
                 "max_rel_intrinsics_error_vs_generating": float(np.max(np.abs(xs[:K] - d["gt_intrinsics"]) /
                                                                        np.maximum(np.abs(d["gt_intrinsics"]), 1.0)))}
        ps.close()
    except Exception as e:  # the solve leg must never take the headline measurement down
        solve = {"error": repr(e)}
    out["solve"] = solve

    # ---- full LM solves of problems whose images are SHARDED over the ranks (strong scaling): the collective of the path
    # -- one in-place all-reduce of the summed normal-equation blocks per evaluation, one of the Schur complement per linear
    # solve -- is on the critical path of every iteration here.  (a) BASELINE.json config 4, Mei 10 000 images in total;
    # (b) EUCM, 100 000 images in total: the size from which sharding is expected to pay.  With one
    # rank these are plain solves; the communicator is passed whenever one exists.
    sharded_solve = {}
    stream_100k = None
    emit_sweep = None

    def beyond_l3_stream(dd, model_b, n_b):
        """The emit step on a working set far beyond the 256 MiB Infinity Cache (100 000 images: 2.15 GB of output per step):
        chain prep + emit kernel on prepared frames, HIP events on the launch stream."""
        from visgeom_amd.benchlib import emit_bytes_per_obs, timed as timed_b

        pb = CalibrationProblem(local_rank)
        cb = pb.add_camera(model_b, dd["init_intrinsics"])
        sb = pb.add_transform(False, dd["init_poses"])
        db = pb.add_dataset(cb, [(sb, 0)], dd["board"], dd["corners"])
        pb.finalize()
        rb, jib, jmb = pb.alloc_outputs(db)

        def step_b():
            pb.prepare()
            pb.evaluate_dataset(db, rb, jib, jmb)

        def emit_b():
            pb.evaluate_dataset(db, rb, jib, jmb)

        reps_b = 40
        t_step, t_emit = timed_b(step_b, reps_b), timed_b(emit_b, reps_b)
        nb = n_b * N * emit_bytes_per_obs(model_b, 1)
        one = capi.load().vg_dataset_single_launch(pb._h, db) == 1
        pb.close()
        del rb, jib, jmb
        traffic_b = None   # HBM bytes per launch from the committed PMC passes of this same command (tools/prof_summary.py)
        try:
            ent = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get("%s_%d_stream" % (model_b, n_b))
            traffic_b = ent["hbm_bytes_per_launch"] if ent else None
        except Exception:
            traffic_b = None
        return {"workload": "%s mono, %d images x %d corners on this GPU: %.2f GB of output per step (beyond the 256 MiB Infinity Cache)" % (model_b.upper(), n_b, N, nb / 1e9),
                "step_ms": t_step * 1e3, "evals_per_s": n_b * N / t_step,
                "launches": "one launch" if one else "vg_chain_prep_multi_kernel + vg_emit_kernel on prepared frames",
                "roofline": {"bound": "hbm", "kernel": "vg_emit_kernel<%s,jac,frames-in-LDS%s>" % (model_b, ",inline-chain" if one else ""),
                             "achieved": nb / t_emit / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nb / t_emit / 1e9 / HBM_PEAK_GBS,
                             "frac_whole_step": nb / t_step / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": nb,
                             "avg_launch_ms": t_emit * 1e3, "traffic": traffic_b}}

    try:
        from visgeom_amd import distributed as vdist

        for key, model_s, n_total, cfg_s in (("mei_10k", "mei", a.sharded_images, 4), ("eucm_100k", "eucm", a.sharded_solve_images, 1),
                                             ("mei_100k", "mei", a.sharded_solve_images, 4)):
            if n_total <= 0:
                continue
            lo, hi = vdist.shard_range(n_total, rank, world)
            dsh = synthetic.make_mono(model_s, hi - lo, cfg_s, first_image=lo)
            runs, summ = [], None
            for rep in range(2):
                psh = CalibrationProblem(local_rank)
                csh = psh.add_camera(model_s, dsh["init_intrinsics"])
                ssh = psh.add_transform(False, dsh["init_poses"])
                psh.add_dataset(csh, [(ssh, 0)], dsh["board"], dsh["corners"])
                psh.finalize()
                fence()
                t0 = time.perf_counter()
                summ = psh.solve(comm=comm, allreduce=vdist.make_allreduce() if (dist is not None and comm is None) else None,
                                 max_num_iterations=100)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                fence()
                if dist is not None:
                    tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
                    all_reduce_(tt, op=dist.ReduceOp.MAX)
                    dt = float(tt.item())
                runs.append(dt * 1e3)
                xsh = psh.get_parameters()
                psh.close()
            if key == "eucm_100k" and not a.no_secondary_configs:
                stream_100k = beyond_l3_stream(dsh, model_s, hi - lo)
                if rank == 0 and not a.headline_kernels_only:   # the Infinity-Cache knee between the headline size and this one, on slices of the same set
                    try:
                        from visgeom_amd import benchlib as _bl

                        big = (150000, 200000, 400000, 1000000) if hi - lo >= 100000 else ()   # the set repeated: steady DRAM streaming
                        emit_sweep = _bl.emit_sweep(dsh, model_s, sorted(set([s_ for s_ in (2500, 5000, 10000, 12500, 15000, 20000, 25000, 50000, 60000, 75000, 100000)
                                                                                if s_ <= hi - lo] + [hi - lo] + list(big))), device=local_rank)
                    except Exception as e:
                        emit_sweep = {"error": repr(e)}
            Ksh = dsh["init_intrinsics"].size
            n_eval, n_schur = solver_message_sizes([Ksh + 7], Ksh)
            lat_s = allreduce_latency_us([n_eval, n_schur], calls=500)
            ms_it = runs[-1] / max(1, summ["num_iterations"])
            coll_ms = None if lat_s is None else (lat_s[str(n_eval)]["median_us"] + lat_s[str(n_schur)]["median_us"]) * 1e-3
            sharded_solve[key] = {
                "rccl_world_size": comm.n_ranks if comm is not None else None,
                # two collectives per iteration on the critical path: their isolated latencies at the exact message sizes, and what is
                # left of the measured iteration once they are taken out (an estimate: the solver queues them between its kernels)
                "allreduce_us": lat_s, "messages_doubles": {"evaluation": n_eval, "schur": n_schur},
                "per_iteration": {"total_ms": ms_it, "collective_ms": coll_ms, "compute_ms": None if coll_ms is None else max(ms_it - coll_ms, 0.0)},
                "workload": "%s mono, %d images x %d corners in total over %d rank(s), full LM solve" % (model_s.upper(), n_total, N, world),
                "scaling": "strong", "images_total": n_total, "images_this_rank": hi - lo, "n_ranks": world,
                "collectives_per_iteration": 0 if world == 1 else 2,
                "iterations": summ["num_iterations"], "termination": summ["termination"], "final_cost": summ["final_cost"],
                "solve_ms_max_over_ranks": runs[-1], "first_solve_ms": runs[0],
                "ms_per_iteration": runs[-1] / max(1, summ["num_iterations"]),
                "max_rel_intrinsics_error_vs_generating": float(np.max(np.abs(xsh[:Ksh] - dsh["gt_intrinsics"]) /
                                                                       np.maximum(np.abs(dsh["gt_intrinsics"]), 1.0)))}
    except Exception as e:  # never take the headline down
        sharded_solve["error"] = repr(e)

    out["sharded_solve"] = sharded_solve
    # ---- BASELINE.json configs 3 (stereo pair, two-member chain) and 5 (four-camera rig, full LM loop) and the beyond-L3
    # stream, measured in the driver's own run: emit step, merged Gram iteration, LM solve; every emit /
    # Gram pass carries its own `roofline` and the kernel name the rocprofv3 trace of tools/prof_configs.sh shows.  Every rank
    # measures its own replica of the configuration on its GPU (no collective: these sizes are one-GPU problems); rank 0 reports.
    if not a.no_secondary_configs:
        from visgeom_amd import benchlib

        small = int(os.environ.get("VG_BENCH_CONFIG_IMAGES", "0")) or None    # rehearsals shrink the configurations
        for key, cfg_c in (("config3_stereo", 3), ("config5_rig", 5)):
            try:
                out[key] = benchlib.section(cfg_c, reps=100 if small is None else 5, device=local_rank, images=small)
            except Exception as e:  # never take the headline down
                out[key] = {"error": repr(e)}
        out["eucm_100k"] = stream_100k if stream_100k is not None else {"skipped": "no 100 k-image set in this run (--sharded-solve-images 0)"}
        if emit_sweep is not None:
            out["emit_sweep"] = emit_sweep
        # ---- the product entry point end to end: `calib a.json` (test/calibration/generic_calibration.cpp:32-44)
        # on a generated calibration file of the headline size -- JSON text in, poses from scratch (estimateInitialGrid,
        # unified_calibration.cpp:1066-1158: 4-corner construction + one independent LM per image), global solve, report and
        # image_error files out -- with the library's per-phase clock; and the pose-initialisation kernel (f2) on its own.
        # Host-heavy: rank 0 only.
        if rank == 0 and not a.headline_kernels_only:
            try:
                e2e_images = small or n_img
                out["calib_e2e"] = benchlib.calib_e2e("mono_%s" % a.model, e2e_images, runs=2, cli=True, device=local_rank)
            except Exception as e:
                out["calib_e2e"] = {"error": repr(e)}
            try:
                pi = benchlib.pose_init(a.model, small or n_img, reps=5, device=local_rank, keep_inputs=True)
                inputs = pi.pop("_inputs")
                if not a.no_cpu_baseline:
                    pi["cpu_baseline"] = pose_init_cpu_baseline(inputs)
                    pi["images_per_s"] = (small or n_img) / (pi["refine_call_ms"] * 1e-3)
                out["pose_init"] = pi
            except Exception as e:
                out["pose_init"] = {"error": repr(e)}
    for key in ("config3_stereo", "config5_rig", "eucm_100k", "calib_e2e", "pose_init"):   # replicas per rank: no collective inside them
        if isinstance(out.get(key), dict):
            out[key]["rccl_world_size"] = comm.n_ranks if comm is not None else None
            out[key]["collective"] = "none (every rank measures its own replica; rank 0 reports)"
    # key order of the line as before: headline fields, then the sections
    out = {k: out[k] for k in list(out)[:list(out).index("config") + 1] + ["roofline", "jtj", "sharded_mei", "sharded_solve", "solve", "pcie_inclusive"] +
           [k for k in ("config3_stereo", "config5_rig", "eucm_100k", "emit_sweep", "calib_e2e", "pose_init") if k in out]}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(d, a.model, a.cpu_seconds)
        out["gpu_over_cpu_allcores"] = value / out["cpu_baseline"]["value"]   # against the threads the box's quota grants (cpu_baseline.cores)
        ex = out["cpu_baseline"].get("extrapolated_all_physical_cores")
        if ex:   # the same ratio against every physical core of the host, by linear extrapolation of the measured per-thread rate
            out["gpu_over_cpu_extrapolated_all_physical_cores"] = value / ex["value"]
    p.close()
    if comm is not None:
        comm.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    secondary_done.set()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
