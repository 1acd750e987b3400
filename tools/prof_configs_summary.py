#!/usr/bin/env python3
"""gpurun_out/prof_configs_<tag>/ (tools/prof_configs.sh) -> tracked evidence under profiles/:

  <tag>_configs_kernel_stats.csv   the rocprofv3 --stats rows of every (config, kind of pass) run, verbatim, with two
                                   leading columns config / pass
  <tag>_configs_pmc.md             per (config, pass, kernel): calls, average ns (trace), FETCH_SIZE / WRITE_SIZE raw KiB,
                                   corrected HBM bytes per launch (factors calibrated on the 512 MiB streams of bench.py,
                                   as tools/prof_summary.py does), algorithmic bytes or flops of that launch and the
                                   fraction of the roofline that bounds it
  <tag>_configs.md                 the table tools/bench_configs.py printed in the same session (HIP-event timings)

usage: python tools/prof_configs_summary.py <tag>
"""
import collections
import csv
import os
import sys

tag = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "gpurun_out", "prof_configs_" + tag)
OUT = os.path.join(ROOT, "profiles")
CAL_BYTES = 64 * 1024 * 1024 * 8
HBM_PEAK, FP64_PEAK = 8.0e12, 78.6e12
KOF = {"eucm": 6, "ucm": 5, "mei": 10}
EVAL_FLOPS = {"eucm": 200, "ucm": 197, "mei": 346}
# (model, chain length, images) of every dataset of a config -- tools/bench_configs.py::build
DATASETS = {2: [("eucm", 1, 1000)], 3: [("eucm", 1, 2000), ("eucm", 2, 2000)], 4: [("mei", 1, 10000)],
            5: [("ucm", 1, 5000), ("eucm", 2, 5000), ("eucm", 2, 5000), ("mei", 2, 5000)]}


def short(name):
    return name.split("(")[0].replace("void ", "")


def stats_rows(d):
    f = os.path.join(P, d, "t_kernel_stats.csv")
    return list(csv.DictReader(open(f))) if os.path.exists(f) else []


def counters(d):
    agg = collections.defaultdict(list)
    f = os.path.join(P, d, "t_counter_collection.csv")
    if os.path.exists(f):
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def emit_bytes(cfg):
    return sum(n * 96 * (32 + 16 * (KOF[m] + 6 * L)) for m, L, n in DATASETS[cfg])


def gram_flops(cfg):
    return sum(n * 96 * (EVAL_FLOPS[m] + 48 * (L - 1) + 2 * (KOF[m] + 6 * L + 1) * (KOF[m] + 6 * L + 2)) for m, L, n in DATASETS[cfg])


cal_f, cal_w = counters("cal_FETCH_SIZE"), counters("cal_WRITE_SIZE")
f_cal = CAL_BYTES / (cal_f["vg::vg_stream_copy_kernel"] * 1024) if "vg::vg_stream_copy_kernel" in cal_f else float("nan")
w_cal = CAL_BYTES / (cal_w["vg::vg_stream_write_kernel"] * 1024) if "vg::vg_stream_write_kernel" in cal_w else float("nan")

all_rows, fields = [], None
traffic_json = {}
md = ["# rocprofv3 per config, round tag `%s`" % tag, "",
      "Source: `tools/prof_configs.sh %s` on one MI355X: per config and kind of pass one `--kernel-trace --stats` run of" % tag,
      "`tools/bench_configs.py --config N --only emit|jtj|solve`, and separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs.",
      "Counter factors calibrated in the same session on bench.py's 512 MiB streams: FETCH_SIZE x **%.4f**, WRITE_SIZE x **%.4f**" % (f_cal, w_cal),
      "(true bytes / (raw KiB x 1024)).  Roofline of an emit kernel: HBM, 8 TB/s, algorithmic bytes 16 + 16 + 16 (K + 6 L) per",
      "observation; of a fused Gram kernel: FP64 vector pipe, 78.6 TFLOP/s, flops per observation = evaluate (200 / 197 / 346 +",
      "48 per extra chain member) + 2 (P + 1)(P + 2).  A launch that serves several datasets is priced with their sum.", ""]
for cfg in sorted(DATASETS):
    for kind in ("emit", "jtj", "solve"):
        rows = stats_rows("c%d_%s" % (cfg, kind))
        if not rows:
            continue
        fields = fields or list(rows[0].keys())
        for r in rows:
            all_rows.append(dict(r, config=cfg, **{"pass": kind}))
        if kind == "solve":
            md += ["## config %d, full LM solve (kernel-trace only)" % cfg, "", "| kernel | calls | avg ns | total % |", "|---|---|---|---|"]
            for r in rows:
                if short(r["Name"]).startswith("vg::"):
                    md.append("| %s | %s | %s | %s |" % (short(r["Name"]), r["Calls"], r["AverageNs"], r.get("Percentage", "-")))
            md.append("")
            continue
        fe, wr = counters("c%d_%s_FETCH_SIZE" % (cfg, kind)), counters("c%d_%s_WRITE_SIZE" % (cfg, kind))
        md += ["## config %d, %s pass" % (cfg, "emit (chain prep + emit launches)" if kind == "emit" else "normal-equation build (fused Gram + sums)"), "",
               "| kernel | calls | avg ns (trace) | FETCH KiB raw | WRITE KiB raw | HBM bytes / launch (corrected) | algorithmic work / launch | fraction of its roofline |",
               "|---|---|---|---|---|---|---|---|"]
        for r in rows:
            k = short(r["Name"])
            if not k.startswith("vg::"):
                continue
            ns = float(r["AverageNs"])
            traffic = fe.get(k, 0.) * 1024 * f_cal + wr.get(k, 0.) * 1024 * w_cal
            work, frac = "-", "-"
            if "vg_emit" in k:
                b = emit_bytes(cfg)
                work, frac = "%.4g B" % b, "%.3f of 8 TB/s (traffic / algorithmic = %.3f)" % (b / (ns * 1e-9) / HBM_PEAK, traffic / b if b else 0)
                if traffic > 0:   # what bench.py's per-config sections report as roofline.traffic (visgeom_amd/benchlib.py)
                    traffic_json["config%d_emit" % cfg] = {"hbm_bytes_per_launch": traffic, "kernel": k, "tag": tag, "trace_avg_ns": ns,
                                                            "fetch_factor": f_cal, "write_factor": w_cal, "algorithmic_bytes": b}
            elif "vg_gram_valu" in k or "vg_gram_fused" in k:
                fl = gram_flops(cfg)
                work, frac = "%.4g flop" % fl, "%.3f of 78.6 TFLOP/s" % (fl / (ns * 1e-9) / FP64_PEAK)
            md.append("| %s | %s | %.0f | %.1f | %.1f | %.4g | %s | %s |" % (k, r["Calls"], ns, fe.get(k, 0.), wr.get(k, 0.), traffic, work, frac))
        md.append("")
if all_rows:
    with open(os.path.join(OUT, tag + "_configs_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=["config", "pass"] + fields)
        w.writeheader()
        for r in all_rows:
            w.writerow(r)
open(os.path.join(OUT, tag + "_configs_pmc.md"), "w").write("\n".join(md) + "\n")
if traffic_json:
    import json

    tj = os.path.join(OUT, "pmc_traffic.json")
    cur = json.load(open(tj)) if os.path.exists(tj) else {}
    cur.update(traffic_json)
    json.dump(cur, open(tj, "w"), indent=1, sort_keys=True)
src = os.path.join(ROOT, "gpurun_out", "bench_configs_%s.txt" % tag)
if os.path.exists(src):
    text = open(src).read()
    open(os.path.join(OUT, tag + "_configs.md"), "w").write(
        "# tools/bench_configs.py, round tag `%s` (one MI355X, HIP-event timings, 200 repetitions)\n\n" % tag +
        "One JSON line per config (with `roofline_emit` / `roofline_jtj`), then the table.\n\n```\n" + text + "```\n")
print("\n".join(md))
