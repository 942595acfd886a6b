# Per-stream timeline of one pipelined bench step (run on the GPU box from the repo root): rocprofv3 kernel trace of a short
# bench.py run, then tools/step_timeline.py -> gpurun_out/tl/timeline.txt (profiles/r04_step_timeline.txt).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tl; mkdir -p $O
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o tl -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra-configs > $O/bench.json 2> $O/prof.err)
find $O/prof -name "*kernel_trace.csv" -exec cp {} $O/kt.csv \;
rm -rf $O/prof
gzip -f $O/kt.csv; python tools/step_timeline.py $O/kt.csv.gz 40 > $O/timeline.txt 2>&1
head -c 400 $O/bench.json; echo; head -12 $O/timeline.txt
