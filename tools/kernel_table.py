#!/usr/bin/env python3
"""Per-kernel table from a rocprofv3 kernel_trace.csv: calls, mean / min duration, share, grouped by
(short kernel name, grid, LDS bytes).  Usage: python tools/kernel_table.py trace.csv [filter] [--last-ms T]
(--last-ms: only dispatches that started within the last T ms of the trace, e.g. the timed steps after a warm-up)"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    depth, out = 0, []
    for ch in name:          # cut the argument list, keep template arguments
        if ch == "(" and depth == 0:
            break
        if ch == "<":
            depth += 1
        if ch == ">":
            depth -= 1
        out.append(ch)
    return "".join(out)


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    args = sys.argv[2:]
    last_ms = None
    if "--last-ms" in args:
        i = args.index("--last-ms")
        last_ms = float(args[i + 1])
        del args[i:i + 2]
    flt = args[0] if args else None
    if last_ms is not None:
        t_end = max(int(r["End_Timestamp"]) for r in rows)
        rows = [r for r in rows if int(r["Start_Timestamp"]) >= t_end - last_ms * 1e6]
    groups = defaultdict(list)
    for r in rows:
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        key = (short(r["Kernel_Name"])[:70], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""),
               r.get("LDS_Block_Size", ""), r.get("VGPR_Count", ""), r.get("Scratch_Size", ""))
        if flt and flt not in key[0]:
            continue
        groups[key].append(us)
    tot = sum(sum(v) for v in groups.values())
    print("%-70s %9s %6s %7s %5s %6s %6s %10s %10s %6s" % ("kernel", "grid.x", "grid.y", "lds", "vgpr", "scr", "calls", "mean us", "min us", "%"))
    for k, v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        print("%-70s %9s %6s %7s %5s %6s %6d %10.1f %10.1f %6.2f" % (k + (len(v), sum(v) / len(v), min(v), 100 * sum(v) / tot)))


if __name__ == "__main__":
    main()
