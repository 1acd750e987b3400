#!/usr/bin/env python3
"""gpurun_out/pmc_sq_<tag>.csv (tools/pmc_sq.sh + tools/pmc_aggregate.py, aggregated on the GPU box) ->
profiles/<tag>_pmc_sq.md: per kernel the SQ counters (means per dispatch) and the ratios read off them.

Units (/opt/skills/guides/MI355X_MICROARCH.md, "rocprofv3 PMC slots"): SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count
quad-cycles summed over waves; WAIT_ANY (parked at s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY
~ WAVE_CYCLES.  Ratios:
  valu/wave      SQ_INSTS_VALU / SQ_WAVES                      instructions a wave issues to the vector ALU
  active_valu    SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES          share of a wave's life spent issuing VALU work
  wait_any       SQ_WAIT_ANY / SQ_WAVE_CYCLES                  share parked on s_waitcnt / barriers
  wait_inst      SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES             share stalled at issue (pipe busy: the other wave of the SIMD)
  simd_valu_util 4 SQ_ACTIVE_INST_VALU / (SQ_BUSY_CYCLES/SE-normalised) is NOT derivable without the SE count of the
                 counter; instead:  valu_issue_share = SQ_ACTIVE_INST_VALU / (SQ_ACTIVE_INST_VALU + idle), reported as
                 ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE) -- the fraction of all SIMD cycles of the launch in which
                 a VALU instruction was being issued (GRBM_GUI_ACTIVE = cycles the GPU was busy with the dispatch).
Fails loudly (exit 3) on an empty table.

usage: python tools/pmc_sq_summary.py <tag>
"""
import collections
import csv
import os
import sys

tag = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "pmc_sq_%s.csv" % tag)
OUT = os.path.join(ROOT, "profiles", "%s_pmc_sq.md" % tag)
N_SIMD = 1024

if not os.path.exists(SRC):
    print("missing %s" % SRC, file=sys.stderr)
    sys.exit(3)
data = collections.defaultdict(dict)   # (workload, kernel) -> counter -> (mean, n)
for r in csv.DictReader(open(SRC)):
    wl = r["run"].rsplit("_", 1)[0]
    data[(wl, r["kernel"])][r["counter"]] = (float(r["mean"]), int(r["dispatches"]))
if not data:
    print("EMPTY counter table in %s" % SRC, file=sys.stderr)
    sys.exit(3)

WANT = ("gram_valu", "emit_kernel", "emit_multi", "reproject", "camera_jacobian", "chain_prep", "schur_rows", "backsub", "pose_lm",
        "lm_", "gram_rows")
lines = ["# SQ counters per kernel, round tag `%s`" % tag, "",
         "Source: `tools/pmc_sq.sh %s` on one MI355X: separate `rocprofv3 --pmc ... --kernel-trace` runs (two passes of 8 counters)," % tag,
         "aggregated on the box by `tools/pmc_aggregate.py` (means per dispatch).  Quad-cycle counters as the guide states.", "",
         "| workload | kernel | dispatches | waves | VALU / wave | wave quad-cycles / wave | active VALU | active any | wait any (s_waitcnt) | wait inst (issue stall) | VALU issue share of all SIMD cycles | LDS insts / wave | LDS bank-conflict share |",
         "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
n_rows = 0
for (wl, k), c in sorted(data.items()):
    if not any(w in k for w in WANT):
        continue
    g = lambda name: c.get(name, (float("nan"), 0))[0]
    waves, wc = g("SQ_WAVES"), g("SQ_WAVE_CYCLES")
    if not waves or waves != waves:
        continue
    util = 4.0 * g("SQ_ACTIVE_INST_VALU") / (N_SIMD * g("GRBM_GUI_ACTIVE")) if g("GRBM_GUI_ACTIVE") == g("GRBM_GUI_ACTIVE") else float("nan")
    lds_conf = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE") if g("SQ_LDS_IDX_ACTIVE") else float("nan")
    lines.append("| %s | %s | %d | %.0f | %.0f | %.0f | %.3f | %.3f | %.3f | %.3f | %.3f | %.1f | %.3f |" % (
        wl, k.replace("vg::", ""), c["SQ_WAVES"][1], waves, g("SQ_INSTS_VALU") / waves, wc / waves, g("SQ_ACTIVE_INST_VALU") / wc,
        g("SQ_ACTIVE_INST_ANY") / wc, g("SQ_WAIT_ANY") / wc, g("SQ_WAIT_INST_ANY") / wc, util, g("SQ_INSTS_LDS") / waves, lds_conf))
    n_rows += 1
lines += ["", "Raw means per dispatch:", "", "| workload | kernel | counter | mean | dispatches |", "|---|---|---|---|---|"]
for (wl, k), c in sorted(data.items()):
    if not any(w in k for w in WANT):
        continue
    for cn, (m, n) in sorted(c.items()):
        lines.append("| %s | %s | %s | %.6g | %d |" % (wl, k.replace("vg::", ""), cn, m, n))
if n_rows == 0:
    print("no kernel of interest in %s" % SRC, file=sys.stderr)
    sys.exit(3)
open(OUT, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:8 + n_rows]))
