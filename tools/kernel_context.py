#!/usr/bin/env python3
"""Durations of one kernel in a rocprofv3 --kernel-trace CSV, grouped by the kernel that ran right before it.
rocprofv3 timestamps of consecutive dispatches are contiguous (end of one = start of the next), so the dispatch gap after
a short kernel is booked on the following one; back-to-back launches of the same kernel show its own duration.
usage: python tools/kernel_context.py <t_kernel_trace.csv> <kernel name substring>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
key = sys.argv[2]
seq = sorted(((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows), key=lambda x: x[1])
by = collections.defaultdict(list)
gaps = collections.defaultdict(list)
for i, (n, s, e) in enumerate(seq):
    if key in n and i:
        by[seq[i - 1][0][:60]].append(e - s)
        gaps[seq[i - 1][0][:60]].append(s - seq[i - 1][2])
print("kernel: *%s*   (ns)" % key)
print("%-62s %6s %8s %8s %8s %10s" % ("preceded by", "calls", "min", "median", "max", "median gap"))
for k, v in sorted(by.items(), key=lambda kv: -len(kv[1])):
    v = sorted(v)
    g = sorted(gaps[k])
    print("%-62s %6d %8d %8d %8d %10d" % (k, len(v), v[0], v[len(v) // 2], v[-1], g[len(g) // 2]))
