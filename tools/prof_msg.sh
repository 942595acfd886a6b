#!/bin/bash
# Per-launch durations of the fused SA/FP MLP kernels in one Pointnet2MSG forward (64 frames).
# usage (GPU box, repo root): bash tools/prof_msg.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_msg
PVN3D_GEOMETRY_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_msg -o msg --output-format csv -- \
  python $R/tools/bench_ops.py --ops msg --reps 2 > /tmp/prof_msg.log 2>&1
tail -3 /tmp/prof_msg.log
find /tmp/prof_msg -name "*.csv" | head
mkdir -p $R/gpurun_out
cp $(find /tmp/prof_msg -name "*kernel_stats.csv" | head -1) $R/gpurun_out/msg_kernel_stats.csv
python - <<PY
import csv, glob, os
f = [p for p in glob.glob("/tmp/prof_msg/**/*kernel_trace.csv", recursive=True)][0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ml = [r for r in rows if "mlp_chain" in r["Kernel_Name"]]
# bench_ops runs the fused forward (1 warm + reps) before the unfused one: take the 2nd forward
seq = ml[12:24]
tot = 0
for r in seq:
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += us
    name = r["Kernel_Name"].split("(")[0][-48:]
    print("%-50s grid=%s/%s lds=%s  %9.1f us" % (name, r.get("Grid_Size_X"), r.get("Grid_Size_Y"), r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", "")), us))
print("sum", tot)
PY
