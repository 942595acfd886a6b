#!/bin/bash
# SQ counters of the kernels whose name matches $1, for `python tools/bench_ops.py --ops $2 --reps 1`.
# usage (GPU box, repo root): bash tools/pmc_kernel.sh ball_query_grid bq "CTR1 CTR2 ..."
R=${GRAFT_REPO_ROOT:-$(pwd)}
PAT=${1:-ball_query_grid}
OPS=${2:-bq}
CTRS=${3:-"SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS"}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_k
timeout 600 rocprofv3 --kernel-trace --pmc $CTRS -d /tmp/pmc_k -o k --output-format csv -- \
  python $R/tools/bench_ops.py --ops $OPS --reps 1 > /tmp/pmc_k.log 2>&1
python - "$PAT" <<PY
import csv, collections, glob, sys
pat = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob("/tmp/pmc_k/**/*counter_collection.csv", recursive=True)[0])))
by = collections.OrderedDict()
for r in rows:
    if pat not in r["Kernel_Name"]:
        continue
    k = (int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0][-40:] + " g=" + r.get("Grid_Size", ""))
    by.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
keys = sorted(by)
names = sorted({n for k in keys for n in by[k]})
print("kernel".ljust(60), " ".join(n[-16:].rjust(16) for n in names))
for k in keys[-8:]:
    print(k[1].ljust(60), " ".join(("%.4g" % by[k].get(n, float("nan"))).rjust(16) for n in names))
PY
