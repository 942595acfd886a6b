#!/bin/bash
# One gpurun session: GPU tests, micro-benchmarks, the default bench line and a rocprofv3 kernel-stats pass.
# Usage (from the repo root on the GPU box): bash tools/gpu_session.sh TAG [extra steps...]
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $OUT/tests.log
echo "tests rc=$?" >> $OUT/tests.log
[ -x tools/microbench.bin ] && timeout 120 tools/microbench.bin > $OUT/microbench.log 2>&1
timeout 600 python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o serial -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-extra-configs --serial > $GRAFT_REPO_ROOT/$OUT/bench_serial_prof.json 2> $GRAFT_REPO_ROOT/$OUT/prof.err)
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/prof -name "*kernel_trace.csv" -exec cp {} $OUT/kernel_trace.csv \;
rm -rf $OUT/prof
tail -3 $OUT/tests.log; cat $OUT/microbench.log; head -c 600 $OUT/bench.json; echo; tail -3 $OUT/bench.err
