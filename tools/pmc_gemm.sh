#!/bin/bash
# SQ counters of the bf16 training GEMM launches of tools/gemm_shapes.py (one dispatch per shape is printed).
# usage (GPU box, repo root): bash tools/pmc_gemm.sh "CTR1 CTR2 ..."
R=${GRAFT_REPO_ROOT:-$(pwd)}
CTRS=${1:-"SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_g
timeout 600 rocprofv3 --kernel-trace --pmc $CTRS -d /tmp/pmc_g -o k --output-format csv -- python $R/tools/gemm_shapes.py > /tmp/pmc_g.log 2>&1
python - <<PY
import csv, collections, glob
rows = list(csv.DictReader(open(glob.glob("/tmp/pmc_g/**/*counter_collection.csv", recursive=True)[0])))
by = collections.OrderedDict()
for r in rows:
    if "mt_gemm_nt" not in r["Kernel_Name"]:
        continue
    k = (int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0][-28:] + " g=" + r.get("Grid_Size", ""))
    by.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
keys = sorted(by)
names = sorted({n for k in keys for n in by[k]})
print("kernel".ljust(44), " ".join(n[-14:].rjust(14) for n in names))
for k in keys[12::13]:          # 13 launches per shape: one of each
    print(k[1].ljust(44), " ".join(("%.4g" % by[k].get(n, float("nan"))).rjust(14) for n in names))
PY
