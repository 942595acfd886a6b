#!/bin/bash
# SQ counters of fps_cells_kernel for one 12288 -> 2048 run (tools/fps_time.py).  usage: bash tools/pmc_fps.sh "CTR ..." 
R=${GRAFT_REPO_ROOT:-$(pwd)}
CTRS=${1:-"SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_f
timeout 300 rocprofv3 --kernel-trace --pmc $CTRS -d /tmp/pmc_f -o k --output-format csv -- \
  python $R/tools/fps_time.py --shapes 12288:2048 --frames 1 --reps 1 > /tmp/pmc_f.log 2>&1
python - <<PY
import csv, collections, glob
rows = list(csv.DictReader(open(glob.glob("/tmp/pmc_f/**/*counter_collection.csv", recursive=True)[0])))
by = collections.OrderedDict()
for r in rows:
    if "fps_" not in r["Kernel_Name"]:
        continue
    k = (int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0][-34:])
    by.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
for k in sorted(by)[:4]:
    print(k[1], " ".join("%s=%.4g" % (n.replace("SQ_", ""), v) for n, v in sorted(by[k].items())))
PY
