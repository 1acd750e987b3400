#!/usr/bin/env python3
"""Runs ON THE GPU BOX at the end of tools/pmc_sq.sh / tools/gpu_check.sh: reduce the per-dispatch rocprofv3 counter CSVs
(tens of MiB: one row per dispatch and counter) to one small CSV  run,kernel,counter,dispatches,mean,min,max  that fits the
64 MiB gpurun_out/ merge.  Exits non-zero when no counter row was found at all (an empty table must not pass silently).

usage: python tools/pmc_aggregate.py <directory with run sub-directories> <out.csv>
"""
import collections
import csv
import glob
import os
import sys


def short(name):
    return name.split("(")[0].replace("void ", "")


def main():
    src, dst = sys.argv[1], sys.argv[2]
    rows = 0
    with open(dst, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["run", "kernel", "counter", "dispatches", "mean", "min", "max"])
        for f in sorted(glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)):
            run = os.path.relpath(f, src).split(os.sep)[0]
            acc = collections.defaultdict(list)
            with open(f) as g:
                for r in csv.DictReader(g):
                    acc[(short(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
            for (k, c), v in sorted(acc.items()):
                if not k.startswith("vg::"):
                    continue
                w.writerow([run, k, c, len(v), "%.6g" % (sum(v) / len(v)), "%.6g" % min(v), "%.6g" % max(v)])
                rows += 1
    print("pmc_aggregate: %d (run, kernel, counter) rows -> %s" % (rows, dst))
    if rows == 0:
        print("pmc_aggregate: NO counter rows found under %s" % src, file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
