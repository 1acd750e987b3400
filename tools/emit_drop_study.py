"""Why does the emit kernel lose a fifth of its rate between 1 GB and 2 GB of output?  One process, one box.

Sections (each prints a table; `python tools/emit_drop_study.py [section ...]`, default all):
  stream   the library's pure streaming-write kernel (vg_calib_stream_write: 16 B per lane, no reads, no arithmetic) over the same
           byte counts: back-to-back launches into ONE buffer, and rotating over enough distinct buffers that no launch finds
           a line of its own previous pass in the 256 MiB Infinity Cache.  If the write kernel has the same hump the memory system
           is the limiter, not the emit kernel.
  sweep    the emit kernel (EUCM, prep + emit route, non-temporal stores) at 25 k .. 1 M images: (a) separate allocations per
           size, (b) slices of ONE allocation made for the largest size, (c) the three output arrays carved from ONE allocation,
           (d) rotating over distinct output sets, (e) one launch at a time with an idle gap in front (events around each launch)
  map      the tile map at >= 1 GB: XCD-contiguous eighths (default) against windows of 8 W tiles (W = 1 is the linear map)
  ramp     per-launch time series after a fresh set-up, after host idles of 10 ms .. 3 s, after a warm-up elsewhere, on new arrays
  place    steady-state series per tile map on fresh arrays, after allocator churn, at shifted addresses; and right behind a
           hipFree of 8 GB (does the driver's wipe of freed VRAM compete with the launches that follow?)
  alloc    addresses / alignment of the output arrays (2 MiB fragments?)
"""
import os
import sys
import time

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from visgeom_amd import CalibrationProblem, capi, synthetic  # noqa: E402
from visgeom_amd.benchlib import HBM_PEAK, N_CORNERS, emit_bytes_per_obs, timed  # noqa: E402

import ctypes  # noqa: E402

L = capi.load()
dev = torch.device("cuda", 0)
sections = sys.argv[1:] or ["alloc", "stream", "sweep", "map"]
MAXN = int(os.environ.get("STUDY_MAX_IMAGES", "1000000"))
SIZES = [n for n in (25000, 50000, 60000, 75000, 100000, 150000, 200000, 400000, 1000000) if n <= MAXN]


def reps_for(nbytes):
    return max(8, min(60, int(40e9 / nbytes)))


def make_problem(d, n, first=0):
    p = CalibrationProblem(0)
    cam = p.add_camera("eucm", d["init_intrinsics"])
    seq = p.add_transform(False, d["init_poses"][first:first + n])
    ds = p.add_dataset(cam, [(seq, 0)], d["board"], d["corners"][first:first + n])
    p.finalize()
    p.force_prepared_frames(True)   # prep + emit at every size (the route of the DRAM-streaming launches)
    p.prepare()
    return p, ds


def line(tag, n, t):
    nbytes = n * N_CORNERS * emit_bytes_per_obs("eucm", 1)
    print("%-44s %8d images %8.1f MB  %9.2f us  %6.3f TB/s  frac %.3f" % (tag, n, nbytes / 1e6, t * 1e6, nbytes / t / 1e12, nbytes / t / HBM_PEAK), flush=True)


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


if "alloc" in sections:
    print("== alloc: device addresses of torch allocations (caching allocator; large blocks come straight from hipMalloc)")
    for mb in (153.6, 460.8, 921.6, 1843.2):
        t = torch.empty(int(mb * 1e6) // 8, dtype=torch.float64, device=dev)
        a = t.data_ptr()
        print("  %8.1f MB at 0x%x  mod 2 MiB = %d KiB, mod 1 GiB = %.1f MiB" % (mb, a, (a % (2 << 20)) >> 10, (a % (1 << 30)) / 2 ** 20))
        del t
    free, total = torch.cuda.mem_get_info()
    print("  free %.1f GB of %.1f GB" % (free / 1e9, total / 1e9))

if "stream" in sections:
    print("== stream: vg_calib_stream_write, per launch")
    big = torch.empty(int(24.0e9) // 8, dtype=torch.float64, device=dev)
    for mb in (215, 250, 500, 750, 1000, 1250, 1500, 2000, 2150, 3000, 4300, 8600, 21500):
        n8 = int(mb * 1e6) // 8
        nsets = max(1, min(int(24.0e9 // (mb * 1e6)), 8))
        k = [0]

        def same():
            L.vg_calib_stream_write(stream_ptr(), ctypes.c_void_p(big.data_ptr()), n8, 1.0)

        def rot():
            L.vg_calib_stream_write(stream_ptr(), ctypes.c_void_p(big.data_ptr() + (k[0] % nsets) * n8 * 8), n8, 1.0)
            k[0] += 1

        r = reps_for(mb * 1e6)
        ts, tr = timed(same, r), timed(rot, r)
        print("  %8.0f MB  same buffer %9.2f us %6.3f TB/s   rotating over %d buffers %9.2f us %6.3f TB/s" % (
            mb, ts * 1e6, mb * 1e6 / ts / 1e12, nsets, tr * 1e6, mb * 1e6 / tr / 1e12), flush=True)
    del big
    torch.cuda.empty_cache()

d = None
if "sweep" in sections or "map" in sections:
    t0 = time.time()
    d = synthetic.make_mono("eucm", max(SIZES), 1)
    print("generated %d images in %.1f s" % (max(SIZES), time.time() - t0), flush=True)


def outputs_separate(p, ds):
    return p.alloc_outputs(ds)


def outputs_from(buf, off, n):
    """res, ji, jm carved from the flat float64 tensor `buf` starting at element `off` (2 MiB aligned pieces)"""
    al = (2 << 20) // 8
    pieces = []
    for cols in (1, 6, 6):
        cnt = n * 2 * N_CORNERS * cols
        pieces.append(buf[off:off + cnt])
        off = (off + cnt + al - 1) // al * al
    res = pieces[0].view(n, 2 * N_CORNERS)
    ji = pieces[1].view(n, 2 * N_CORNERS, 6)
    jm = [pieces[2].view(n, 2 * N_CORNERS, 6)]
    return (res, ji, jm), off


if "sweep" in sections:
    print("== sweep (a): separate allocations per size, back-to-back launches")
    for n in SIZES:
        p, ds = make_problem(d, n)
        res, ji, jm = outputs_separate(p, ds)
        nb = n * N_CORNERS * emit_bytes_per_obs("eucm", 1)
        line("separate allocations", n, timed(lambda: p.evaluate_dataset(ds, res, ji, jm), reps_for(nb)))
        p.close()
        del res, ji, jm
        torch.cuda.empty_cache()
    print("== sweep (b): slices of the arrays allocated for the LARGEST size")
    nmax = max(SIZES)
    pbig, dsbig = make_problem(d, nmax)
    RES, JI, JM = outputs_separate(pbig, dsbig)
    pbig.close()
    for n in SIZES:
        p, ds = make_problem(d, n)
        nb = n * N_CORNERS * emit_bytes_per_obs("eucm", 1)
        line("slices of one set of arrays", n, timed(lambda: p.evaluate_dataset(ds, RES[:n], JI[:n], [JM[0][:n]]), reps_for(nb)))
        p.close()
    del RES, JI, JM
    torch.cuda.empty_cache()
    print("== sweep (c): res | jac_intr | jac_pose carved from ONE allocation")
    for n in SIZES:
        p, ds = make_problem(d, n)
        buf = torch.empty(n * 2 * N_CORNERS * 13 + 3 * (2 << 20) // 8, dtype=torch.float64, device=dev)
        (res, ji, jm), _ = outputs_from(buf, 0, n)
        nb = n * N_CORNERS * emit_bytes_per_obs("eucm", 1)
        line("one allocation, three arrays", n, timed(lambda: p.evaluate_dataset(ds, res, ji, jm), reps_for(nb)))
        p.close()
        del buf, res, ji, jm
        torch.cuda.empty_cache()
    print("== sweep (d): rotating over distinct output sets (no launch overwrites lines of the previous one)")
    for n in [s for s in SIZES if s <= 400000]:
        p, ds = make_problem(d, n)
        nb = n * N_CORNERS * emit_bytes_per_obs("eucm", 1)
        nsets = max(2, min(6, int(30e9 // nb)))
        sets = [outputs_separate(p, ds) for _ in range(nsets)]
        k = [0]

        def rot():
            res, ji, jm = sets[k[0] % nsets]
            k[0] += 1
            p.evaluate_dataset(ds, res, ji, jm)

        line("rotating over %d output sets" % nsets, n, timed(rot, reps_for(nb)))
        p.close()
        del sets
        torch.cuda.empty_cache()
    print("== sweep (e): one launch at a time, device idle in front (events around each launch, median of 15)")
    for n in [s for s in SIZES if s <= 400000]:
        p, ds = make_problem(d, n)
        res, ji, jm = outputs_separate(p, ds)
        ts = []
        for _ in range(18):
            torch.cuda.synchronize()
            time.sleep(0.002)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            p.evaluate_dataset(ds, res, ji, jm)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e-3)
        line("single launches, idle in front", n, float(np.median(ts[3:])))
        p.close()
        del res, ji, jm
        torch.cuda.empty_cache()

if "map" in sections:
    print("== map: tile map of the emit launch (hook emit_map_window: W tiles of 256 observations per XCD run; 0 = contiguous eighths)")
    for n in [s for s in (50000, 100000, 200000, 1000000) if s <= MAXN]:
        p, ds = make_problem(d, n)
        res, ji, jm = outputs_separate(p, ds)
        nb = n * N_CORNERS * emit_bytes_per_obs("eucm", 1)
        for rep in range(2):
            for W in (0, 1, 4, 16, 64, 256, 1024, 4096):
                capi.debug_set("emit_map_window", W)
                line("W = %d%s" % (W, " (contiguous eighths)" if W == 0 else " (linear)" if W == 1 else " (%.1f MiB runs)" % (W * 24 / 1024.)), n,
                     timed(lambda: p.evaluate_dataset(ds, res, ji, jm), reps_for(nb)))
        capi.debug_set("emit_map_window", 0)
        for rep in range(2):
            for nt, name in ((10 ** 12, "plain stores"), (1, "non-temporal stores")):
                capi.debug_set("emit_nt_min_bytes", nt)
                line(name, n, timed(lambda: p.evaluate_dataset(ds, res, ji, jm), reps_for(nb)))
        capi.debug_set("emit_nt_min_bytes", 0)
        p.close()
        del res, ji, jm
        torch.cuda.empty_cache()


def dpm_state():
    """the starred (current) level of the clock tables amdgpu exposes, when the box has them"""
    import glob

    out = []
    for f in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_*clk"))[:0]:   # eight cards are visible, none identifiable: off
        try:
            cur = [ln.strip() for ln in open(f) if "*" in ln]
            out.append("%s=%s" % (os.path.basename(f)[7:], cur[0].split(":")[1].strip(" *") if cur else "?"))
        except OSError:
            pass
    return " ".join(out) or "(no pp_dpm tables)"


def series(fn, count):
    """`count` back-to-back launches, an event between each pair: per-launch seconds"""
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(count + 1)]
    ev[0].record()
    for i in range(count):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return np.array([ev[i].elapsed_time(ev[i + 1]) * 1e-3 for i in range(count)])


def show(tag, ts, nbytes):
    t = np.cumsum(ts)
    marks = [0, 1, 2, 4, 8, 16, 32, 64, 96, 128, 192, 255, 383, 511]
    print("  %-58s" % tag + "  ".join("%d:%.0f" % (i, ts[i] * 1e6) for i in marks if i < len(ts)))
    half = len(ts) // 2
    print("  %-58s first 8 launches %.3f of peak, last half %.3f of peak; cumulative %.1f ms; %s" % (
        "", nbytes / ts[:8].mean() / HBM_PEAK, nbytes / ts[half:].mean() / HBM_PEAK, t[-1] * 1e3, dpm_state()), flush=True)


if "ramp" in sections:
    print("== ramp: per-launch time series of the emit launch (launch index : microseconds), EUCM prep + emit route")
    if d is None:
        d = synthetic.make_mono("eucm", 200000, 1)
    for n in (100000, 50000, 200000):
        nb = n * N_CORNERS * emit_bytes_per_obs("eucm", 1)
        cnt = 512 if n <= 100000 else 256
        p, ds = make_problem(d, n)
        torch.cuda.synchronize()
        t_host = time.time()
        res, ji, jm = outputs_separate(p, ds)

        def emit():
            p.evaluate_dataset(ds, res, ji, jm)

        print(" %d images, %.1f MB per launch; before: %s" % (n, nb / 1e6, dpm_state()))
        show("A fresh problem, fresh output arrays", series(emit, cnt), nb)
        show("B again at once (same arrays)", series(emit, cnt), nb)
        for idle in (0.01, 0.05, 0.2, 1.0, 3.0):
            torch.cuda.synchronize()
            time.sleep(idle)
            show("C after %.2f s of host sleep (device idle)" % idle, series(emit, cnt), nb)
        torch.cuda.synchronize()
        time.sleep(3.0)
        other = torch.empty(int(1.0e9) // 8, dtype=torch.float64, device=dev)
        t0 = time.time()
        while time.time() - t0 < 0.08:
            for _ in range(20):
                L.vg_calib_stream_write(stream_ptr(), ctypes.c_void_p(other.data_ptr()), other.numel(), 1.0)
            torch.cuda.synchronize()
        show("D 3 s idle, then 80 ms of stream writes elsewhere, then", series(emit, cnt), nb)
        torch.cuda.synchronize()
        time.sleep(3.0)
        fresh = outputs_separate(p, ds)
        show("E 3 s idle, NEW output arrays (never touched)", series(lambda: p.evaluate_dataset(ds, fresh[0], fresh[1], fresh[2]), cnt), nb)
        show("F the new arrays again at once", series(lambda: p.evaluate_dataset(ds, fresh[0], fresh[1], fresh[2]), cnt), nb)
        show("G the FIRST arrays again at once", series(emit, cnt), nb)
        p.close()
        del res, ji, jm, fresh, other
        torch.cuda.empty_cache()


if "place" in sections:
    print("== place: steady launch time (mean of 100 back-to-back launches after 30) per tile map, W = 0 contiguous eighths / W tiles per run")
    if d is None:
        d = synthetic.make_mono("eucm", 200000, 1)

    def steady(fn, count=100):
        series(fn, 30)
        return series(fn, count).mean()

    def table(tag, p, ds, outs, n):
        nb = n * N_CORNERS * emit_bytes_per_obs("eucm", 1)
        res, ji, jm = outs
        cells = []
        for W in (0, 16, 64, 1, 0, 16, 64, 1):
            capi.debug_set("emit_map_window", W)
            t = steady(lambda: p.evaluate_dataset(ds, res, ji, jm))
            cells.append("W=%d %.0f us (%.3f)" % (W, t * 1e6, nb / t / HBM_PEAK))
        capi.debug_set("emit_map_window", 0)
        print("  %-52s res 0x%x ji 0x%x jm 0x%x" % (tag, res.data_ptr(), ji.data_ptr(), jm[0].data_ptr()))
        print("      " + "  ".join(cells), flush=True)

    for n in (100000, 200000):
        print(" %d images" % n)
        p, ds = make_problem(d, n)
        outs = outputs_separate(p, ds)
        time.sleep(0.5)
        table("fresh process / fresh arrays", p, ds, outs, n)
        del outs
        torch.cuda.empty_cache()
        churn = [torch.empty(int(g * 1e9) // 8, dtype=torch.float64, device=dev) for g in (3.1, 0.7, 5.3, 1.9, 9.7)]
        for t in churn:
            t.fill_(1.0)
        del churn[1], churn[2]
        keep = churn
        outs = outputs_separate(p, ds)
        time.sleep(0.5)
        table("after allocator churn (holes of 0.7 and 1.9 GB)", p, ds, outs, n)
        del outs, keep, churn
        torch.cuda.empty_cache()
        time.sleep(0.5)
        for shift_mb in (0, 1, 3, 17):
            buf = torch.empty(n * 2 * N_CORNERS * 13 + 64 * (2 << 20) // 8, dtype=torch.float64, device=dev)
            al = (2 << 20) // 8
            off = 0
            pieces = []
            for k, cols in enumerate((1, 6, 6)):
                cnt = n * 2 * N_CORNERS * cols
                pieces.append(buf[off:off + cnt])
                off = (off + cnt + al - 1) // al * al + k * shift_mb * (1 << 20) // 8 + shift_mb * (1 << 20) // 8
            outs = (pieces[0].view(n, 2 * N_CORNERS), pieces[1].view(n, 2 * N_CORNERS, 6), [pieces[2].view(n, 2 * N_CORNERS, 6)])
            table("one allocation, arrays shifted by %d / %d MiB" % (shift_mb, 2 * shift_mb), p, ds, outs, n)
            del outs, pieces, buf
            torch.cuda.empty_cache()
        # the wipe of freed VRAM
        outs = outputs_separate(p, ds)
        res, ji, jm = outs
        nb = n * N_CORNERS * emit_bytes_per_obs("eucm", 1)
        fn = lambda: p.evaluate_dataset(ds, res, ji, jm)  # noqa: E731
        series(fn, 30)
        show("steady, nothing freed", series(fn, 128), nb)
        for gb in (2.0, 8.0):
            scratch = torch.empty(int(gb * 1e9) // 8, dtype=torch.float64, device=dev)
            scratch.fill_(2.0)
            series(fn, 30)
            torch.cuda.synchronize()
            del scratch
            torch.cuda.empty_cache()   # hipFree
            show("right behind hipFree of %.0f GB" % gb, series(fn, 128), nb)
            show("  ... and the next 128 launches", series(fn, 128), nb)
        p.close()
        del outs, res, ji, jm
        torch.cuda.empty_cache()

if "map2" in sections:
    print("== map2: windowed tile map against contiguous eighths at every size, library's own route and store policy, steady state (100 launches after 30), twice")
    for model in ("eucm", "mei"):
        sizes = [2500, 5000, 10000, 12500, 15000, 25000, 50000, 100000, 200000] if model == "eucm" else [5000, 10000, 50000, 100000]
        dm = synthetic.make_mono(model, max(sizes), 1)
        for n in sizes:
            p = CalibrationProblem(0)
            cam = p.add_camera(model, dm["init_intrinsics"])
            seq = p.add_transform(False, dm["init_poses"][:n])
            ds = p.add_dataset(cam, [(seq, 0)], dm["board"], dm["corners"][:n])
            p.finalize()
            p.prepare()
            res, ji, jm = p.alloc_outputs(ds)
            nb = n * N_CORNERS * emit_bytes_per_obs(model, 1)
            cells = []
            for W in (0, 4, 16, 64, 0, 4, 16, 64):
                capi.debug_set("emit_map_window", W)
                series(lambda: p.evaluate_dataset(ds, res, ji, jm), 30)
                t = series(lambda: p.evaluate_dataset(ds, res, ji, jm), 100).mean()
                cells.append("W=%d %.1f us (%.3f)" % (W, t * 1e6, nb / t / HBM_PEAK))
            capi.debug_set("emit_map_window", 0)
            print("  %-5s %7d images %7.1f MB %s:  %s" % (model, n, nb / 1e6, "inline" if L.vg_dataset_single_launch(p._h, ds) == 1 else "prep  ", "  ".join(cells)), flush=True)
            p.close()
            del res, ji, jm
            torch.cuda.empty_cache()

if "multi" in sections:
    print("== multi: merged emit launch of configs 3 / 5 (and 4 x their size), contiguous eighths per dataset (hook -1) against the default windows")
    from visgeom_amd import benchlib

    for cfg, images in ((3, None), (5, None), (3, 20000), (5, 20000)):
        p, dss, gt, name = benchlib.build(cfg, 0, images)
        f = benchlib.passes(p, dss)
        p.prepare()
        nb = sum(n * N_CORNERS * emit_bytes_per_obs(m, Lc) for _, m, Lc, n in dss)
        cells = []
        for hook in (-1, 0, -1, 0, -1, 0):
            capi.debug_set("emit_map_window", hook)
            series(f["emit_only"], 30)
            t = series(f["emit_only"], 100).mean()
            cells.append("%s %.1f us (%.3f)" % ("eighths" if hook else "windows", t * 1e6, nb / t / HBM_PEAK))
        capi.debug_set("emit_map_window", 0)
        print("  %-46s %7.1f MB: %s" % (name, nb / 1e6, "  ".join(cells)), flush=True)
        p.close()
        del f
        torch.cuda.empty_cache()

if "headline" in sections:
    print("== headline: 10 k images (inside the Infinity Cache), contiguous eighths (hook -1) against windows W = 16 / 4 / 64, 400 launches each, alternating")
    for model in ("eucm", "mei"):
        dm = synthetic.make_mono(model, 10000, 1)
        p = CalibrationProblem(0)
        cam = p.add_camera(model, dm["init_intrinsics"])
        seq = p.add_transform(False, dm["init_poses"])
        ds = p.add_dataset(cam, [(seq, 0)], dm["board"], dm["corners"])
        p.finalize()
        p.prepare()
        res, ji, jm = p.alloc_outputs(ds)
        nb = 10000 * N_CORNERS * emit_bytes_per_obs(model, 1)
        for rep in range(5):
            cells = []
            for W in (-1, 16, 4, 64):
                capi.debug_set("emit_map_window", W)
                series(lambda: p.evaluate_dataset(ds, res, ji, jm), 50)
                t = timed(lambda: p.evaluate_dataset(ds, res, ji, jm), 400)
                cells.append("%s %.2f us (%.3f)" % ("eighths" if W < 0 else "W=%d" % W, t * 1e6, nb / t / HBM_PEAK))
            print("  %s: %s" % (model, "  ".join(cells)), flush=True)
        capi.debug_set("emit_map_window", 0)
        p.close()
