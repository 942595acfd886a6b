#!/usr/bin/env python
"""Timeline of ONE pipelined bench step from a rocprofv3 kernel trace (csv or csv.gz): per kernel start offset, duration,
HIP stream, and how much of it ran beside kernels of OTHER streams.
usage: step_timeline.py kernel_trace.csv[.gz] [min_us]      (trace of `bench.py --steps 4 --warmup 2 ...`)"""
import csv, gzip, io, re, sys
from collections import defaultdict

path = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
f = io.TextIOWrapper(gzip.open(path)) if path.endswith(".gz") else open(path)
rows = list(csv.DictReader(f))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
# the pipelined steps are the ones whose level-0 FPS runs on a side stream; one such launch per step
fps = [r for r in rows if "fps_cells_kernel" in r["Kernel_Name"] and r["e"] - r["s"] > 1_000_000 and r["Stream_Id"] != "0"]
if len(fps) < 3:
    print("not enough pipelined steps in the trace"); sys.exit(0)
t0, t1 = fps[-3]["s"], fps[-2]["s"]
step = [r for r in rows if t0 <= r["s"] < t1]
print("step span %.3f ms, %d launches" % ((t1 - t0) / 1e6, len(step)))


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    return n[:50]


def union(iv):
    iv = sorted(iv); tot = 0; cs = ce = None
    for s, e in iv:
        if cs is None: cs, ce = s, e
        elif s <= ce: ce = max(ce, e)
        else: tot += ce - cs; cs, ce = s, e
    if cs is not None: tot += ce - cs
    return tot


by_q = defaultdict(list)
for r in step:
    by_q[r["Stream_Id"]].append((r["s"], r["e"]))
for q, iv in sorted(by_q.items()):
    print("stream %s: %d launches, busy %.3f ms" % (q, len(iv), union(iv) / 1e6))
print("any stream busy: %.3f ms" % (union([(r["s"], r["e"]) for r in step]) / 1e6))


def overlap_with_others(r):
    tot = 0
    for q, iv in by_q.items():
        if q == r["Stream_Id"]: continue
        for s, e in iv:
            lo, hi = max(s, r["s"]), min(e, r["e"])
            if hi > lo: tot += hi - lo
    return tot


print("%9s %9s %6s %6s  kernel (launches >= %.0f us)" % ("start us", "dur us", "stream", "ovl %", min_us))
for r in step:
    d = r["e"] - r["s"]
    if d / 1e3 < min_us: continue
    print("%9.1f %9.1f %6s %6.0f  %s" % ((r["s"] - t0) / 1e3, d / 1e3, r["Stream_Id"], 100.0 * min(1.0, overlap_with_others(r) / d),
                                        short(r["Kernel_Name"])))
