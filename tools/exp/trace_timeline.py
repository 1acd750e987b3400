"""Timeline of the LAST burst of dispatches in a rocprofv3 kernel trace (one warm solve): start offset, duration, gap to the
previous dispatch.  usage: python tools/exp/trace_timeline.py <t_kernel_trace.csv> [max gap us that still belongs to the burst]"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
lim = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 300e3
i = len(rows) - 1
while i > 0 and int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]) < lim:
    i -= 1
t0 = int(rows[i]["Start_Timestamp"])
prev = None
for r in rows[i:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("vg::", "")[:44]
    print("%9.2f us  %-46s %7.2f us   gap %6.2f" % ((s - t0) / 1e3, name, (e - s) / 1e3, 0. if prev is None else (s - prev) / 1e3))
    prev = e
print("burst: %.2f us" % ((prev - t0) / 1e3))
