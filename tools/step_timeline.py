#!/usr/bin/env python
"""Timeline of ONE pipelined bench step from a rocprofv3 kernel trace (csv): per kernel start offset, duration, queue, and
how much of it ran beside kernels of OTHER queues.  usage: step_timeline.py kernel_trace.csv [min_us]"""
import csv, re, sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
# steps are delimited by the level-0 FPS launches (one per step, the longest fps_cells launch)
fps = [r for r in rows if "fps_cells_kernel" in r["Kernel_Name"] and (r["e"] - r["s"]) > 1_000_000]
if len(fps) < 3:
    print("not enough steps in the trace"); sys.exit(0)
t0, t1 = fps[-2]["s"], fps[-1]["s"]
step = [r for r in rows if t0 <= r["s"] < t1]
print("step span %.3f ms, %d launches" % ((t1 - t0) / 1e6, len(step)))
qs = sorted({r["Queue_Id"] for r in step})
print("queues:", qs)
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    return n[:58]
# busy time per queue and union
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = None, None
    for s, e in iv:
        if cs is None: cs, ce = s, e
        elif s <= ce: ce = max(ce, e)
        else: tot += ce - cs; cs, ce = s, e
    if cs is not None: tot += ce - cs
    return tot
for q in qs:
    iv = [(r["s"], r["e"]) for r in step if r["Queue_Id"] == q]
    print("queue %s: %d launches, busy %.3f ms" % (q, len(iv), union(iv) / 1e6))
print("any queue busy: %.3f ms" % (union([(r["s"], r["e"]) for r in step]) / 1e6))
by_q = defaultdict(list)
for r in step: by_q[r["Queue_Id"]].append((r["s"], r["e"]))
def overlap_with_others(r):
    tot = 0
    for q, iv in by_q.items():
        if q == r["Queue_Id"]: continue
        for s, e in iv:
            lo, hi = max(s, r["s"]), min(e, r["e"])
            if hi > lo: tot += hi - lo
    return tot
print("%9s %9s %5s %7s  kernel" % ("start us", "dur us", "q", "ovl %"))
for r in step:
    d = r["e"] - r["s"]
    if d / 1e3 < min_us: continue
    print("%9.1f %9.1f %5s %7.0f  %s" % ((r["s"] - t0) / 1e3, d / 1e3, r["Queue_Id"][-3:], 100.0 * min(1.0, overlap_with_others(r) / d), short(r["Kernel_Name"])))
