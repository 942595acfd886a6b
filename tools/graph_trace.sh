# Kernel timeline of the single-frame vote -> cluster -> pose call as a HIP-graph replay (run on the GPU box from the repo root):
# plain timing, then a rocprofv3 kernel trace of tools/graph_trace.py; gpurun_out/gt/timeline.txt lists the last launches.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/gt; mkdir -p $O
timeout 200 python tools/graph_trace.py > $O/plain.log 2>&1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o gt -- python $R/tools/graph_trace.py > $O/prof.log 2> $O/prof.err)
find $O/prof -name "*kernel_trace.csv" -exec cp {} $O/kt.csv \;
rm -rf $O/prof
python - <<'PY'
import csv, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/gt"
rows = list(csv.DictReader(open(O + "/kt.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 400 launches: print name, duration, gap to the previous end
out = []
prev = None
for r in rows[-700:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) if prev else 0
    out.append("%8.1f %8.1f  %s" % (gap / 1e3, (e - s) / 1e3, r["Kernel_Name"][:90]))
    prev = e
open(O + "/timeline.txt", "w").write("\n".join(out))
PY
cat $O/plain.log; tail -3 $O/prof.log
