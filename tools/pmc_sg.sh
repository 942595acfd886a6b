#!/bin/bash
# SQ counters of the split-GEMM kernels for one launch shape of tools/sg_time.py (one rocprofv3 pass per counter group).
# usage (GPU box, repo root): bash tools/pmc_sg.sh "FP2 Y" "CTR1 CTR2 ..." ["CTR3 ..." ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
SHAPE=${1:-FP2 Y}
shift
cd /tmp && export TMPDIR=/tmp
for CTRS in "$@"; do
rm -rf /tmp/pmc_sg
timeout 300 rocprofv3 --kernel-trace --pmc $CTRS -d /tmp/pmc_sg -o k --output-format csv -- \
  python $R/tools/sg_time.py --reps 1 --only "$SHAPE" > /tmp/pmc_sg.log 2>&1
python - <<PY
import csv, collections, glob
rows = list(csv.DictReader(open(glob.glob("/tmp/pmc_sg/**/*counter_collection.csv", recursive=True)[0])))
by = collections.OrderedDict()
for r in rows:
    if "sg_gemm" not in r["Kernel_Name"]:
        continue
    n = r["Kernel_Name"]
    k = (int(r["Dispatch_Id"]), n[n.index("sg_gemm"):][:30] + " g=" + r.get("Grid_Size", ""))
    by.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
keys = sorted(by)
names = sorted({n for k in keys for n in by[k]})
print("kernel".ljust(36), " ".join(n[-18:].rjust(18) for n in names))
seen = {}
for k in keys:
    seen[k[1]] = k
for k in seen.values():
    print(k[1].ljust(36), " ".join(("%.4g" % by[k].get(n, float("nan"))).rjust(18) for n in names))
PY
done
