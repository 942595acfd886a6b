#!/bin/bash
# SQ counters per fused-MLP launch of one Pointnet2MSG forward.  usage: bash tools/pmc_msg.sh "CTR1 CTR2 ..."
R=${GRAFT_REPO_ROOT:-$(pwd)}
CTRS=${1:-"SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES"}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_msg
PVN3D_GEOMETRY_STREAM=0 timeout 600 rocprofv3 --kernel-trace --pmc $CTRS -d /tmp/pmc_msg -o m --output-format csv -- \
  python $R/tools/bench_ops.py --ops msg --reps 1 > /tmp/pmc_msg.log 2>&1
tail -2 /tmp/pmc_msg.log
python - <<PY
import csv, collections
rows = list(csv.DictReader(open("/tmp/pmc_msg/m_counter_collection.csv")))
by = collections.OrderedDict()
for r in rows:
    if "mlp_chain" not in r["Kernel_Name"]:
        continue
    k = (int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0][-34:] + " g=" + r.get("Grid_Size", ""))
    by.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
keys = sorted(by)[12:24]
names = sorted({n for k in keys for n in by[k]})
print("kernel".ljust(52), " ".join(n[-18:].rjust(18) for n in names))
for k in keys:
    print(k[1].ljust(52), " ".join(("%.3g" % by[k].get(n, float("nan"))).rjust(18) for n in names))
PY
