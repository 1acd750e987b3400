"""Gaps between consecutive dispatches of a rocprofv3 kernel trace (start of a kernel - end of the one before it), grouped by the
pair of kernel names: where an LM iteration spends the time its kernels do not account for.
usage: python tools/exp/trace_gaps.py <t_kernel_trace.csv>"""
import collections
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.split("(")[0].replace("void ", "").replace("vg::", "")[:38]
gaps, durs = collections.defaultdict(list), collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    if g < 200000:   # same burst of work
        gaps[(short(a["Kernel_Name"]), short(b["Kernel_Name"]))].append(g)
for r in rows:
    durs[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("%-40s -> %-40s %6s %9s %9s" % ("kernel", "next kernel", "n", "gap us", "median"))
for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1])):
    if len(v) >= 5:
        v2 = sorted(v)
        print("%-40s -> %-40s %6d %9.2f %9.2f" % (k[0], k[1], len(v), sum(v) / len(v) / 1e3, v2[len(v2) // 2] / 1e3))
print()
for k, v in sorted(durs.items(), key=lambda kv: -sum(kv[1])):
    if len(v) >= 5:
        print("%-40s n %5d  mean %8.2f us" % (k, len(v), sum(v) / len(v) / 1e3))
