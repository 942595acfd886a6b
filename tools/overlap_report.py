#!/usr/bin/env python3
"""How much do the two islands of the bench step overlap?  From a rocprofv3 kernel trace of the DEFAULT
(multi-stream) bench run: for the last `--last-ms` of the trace, the union of busy time per kernel class
(mfma = fused MLP kernels, valu = MeanShift / vote kernels, fps, other), their pairwise overlap, and the
mean duration of each MLP / ms_iter kernel (compare with the serial trace).
usage: python tools/overlap_report.py trace.csv [--last-ms 120]"""
import csv
import sys


def cls(name):
    if "mlp_chain" in name:
        return "mfma"
    if "ms_" in name or "vote_compact" in name or "best_fit" in name:
        return "valu"
    if "fps_" in name:
        return "fps"
    return "other"


def union(iv):
    iv = sorted(iv)
    out, cur = [], None
    for a, b in iv:
        if cur is None or a > cur[1]:
            if cur:
                out.append(cur)
            cur = [a, b]
        else:
            cur[1] = max(cur[1], b)
    if cur:
        out.append(cur)
    return out


def length(iv):
    return sum(b - a for a, b in iv)


def inter(x, y):
    i = j = 0
    out = []
    while i < len(x) and j < len(y):
        a, b = max(x[i][0], y[j][0]), min(x[i][1], y[j][1])
        if a < b:
            out.append((a, b))
        if x[i][1] < y[j][1]:
            i += 1
        else:
            j += 1
    return out


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    last_ms = float(sys.argv[sys.argv.index("--last-ms") + 1]) if "--last-ms" in sys.argv else 120.0
    t_end = max(int(r["End_Timestamp"]) for r in rows)
    rows = [r for r in rows if int(r["Start_Timestamp"]) >= t_end - last_ms * 1e6]
    t0 = min(int(r["Start_Timestamp"]) for r in rows)
    by = {}
    dur = {}
    for r in rows:
        c = cls(r["Kernel_Name"])
        a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        by.setdefault(c, []).append((a, b))
        key = r["Kernel_Name"].split("(")[0].replace("(anonymous namespace)::", "")[-40:] + " g=" + r.get("Grid_Size_X", "")
        if c in ("mfma",) or "ms_iter" in key:
            dur.setdefault(key, []).append((b - a) / 1e3)
    u = {k: union(v) for k, v in by.items()}
    span = (t_end - t0) / 1e6
    print("window %.1f ms" % span)
    for k, v in u.items():
        print("  %-6s busy %.2f ms (%.0f %% of the window)" % (k, length(v) / 1e6, 100 * length(v) / 1e6 / span))
    if "mfma" in u and "valu" in u:
        ov = length(inter(u["mfma"], u["valu"])) / 1e6
        print("  mfma AND valu kernels in flight together: %.2f ms (%.0f %% of the valu time)" % (ov, 100 * ov / (length(u["valu"]) / 1e6)))
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:16]:
        print("  %-60s n=%3d mean %8.1f us" % (k, len(v), sum(v) / len(v)))


if __name__ == "__main__":
    main()
