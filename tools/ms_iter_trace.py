#!/usr/bin/env python3
"""Durations of the successive ms_iter launches of one heavy-tailed batch, from a rocprofv3 kernel trace csv:
python tools/ms_iter_trace.py <kernel_trace.csv>"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "ms_iter" in r["Kernel_Name"] or "ms_compact" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last fit_batch call = the launches after the last gap > 2 ms
cut = 0
for i in range(1, len(rows)):
    if int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]) > 2_000_000:
        cut = i
rows = rows[cut:]
t0 = int(rows[0]["Start_Timestamp"])
it = 0
for r in rows:
    name = "iter" if "ms_iter" in r["Kernel_Name"] else "compact"
    if name == "iter":
        it += 1
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if it <= 12 or it % 16 == 0 or name == "compact" and it % 16 == 1:
        print("%-7s t=%3d  start %8.1f us  dur %7.1f us" % (name, it, (int(r["Start_Timestamp"]) - t0) / 1e3, dur))
print("launches", len(rows), "span %.2f ms" % ((int(rows[-1]["End_Timestamp"]) - t0) / 1e6),
      "sum of kernel time %.2f ms" % (sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows) / 1e6))
