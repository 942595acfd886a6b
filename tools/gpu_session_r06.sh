#!/bin/bash
# Round-6 evidence session (run on the GPU box from the repo root, AFTER the last kernel commit): PMC passes first
# (bench.py reads their JSON), kernel traces, the per-stream step timeline, the co-execution / fault evidence, then the
# default bench line.  Everything lands under gpurun_out/r06e/ and the summaries are copied to profiles/r06_* by the caller.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06e
mkdir -p $O
timeout 500 python tools/pmc_mfma.py r06 > $O/pmc_mfma.log 2>&1
timeout 600 python tools/pmc_traffic.py r06 > $O/pmc_traffic.log 2>&1
timeout 400 python tools/pmc_train.py r06 > $O/pmc_train.log 2>&1
cp profiles/r06_mfma_busy.json profiles/r06_pmc_traffic.json profiles/latest_pmc_traffic.json profiles/r06_train_pmc_traffic.json profiles/latest_train_pmc_traffic.json $O/ 2>/dev/null
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o serial -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-extra-configs --serial > $O/bench_serial_under_rocprof.json 2> $O/prof.err)
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/bench_serial_kernel_stats.csv \;
find $O/prof -name "*kernel_trace.csv" -exec cp {} $O/bench_serial_kernel_trace.csv \;
rm -rf $O/prof
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/proft -o train -- python $R/tools/train_bench.py --frames 24 --steps 3 > $O/train_bench_prof.log 2> $O/proft.err)
find $O/proft -name "*kernel_trace.csv" -exec cp {} $O/train_step_kernel_trace.csv \;
rm -rf $O/proft
python tools/kernel_table.py $O/bench_serial_kernel_trace.csv > $O/bench_serial_kernel_table.txt 2>&1
python tools/kernel_table.py $O/train_step_kernel_trace.csv > $O/train_step_kernel_table.txt 2>&1
rm -f $O/bench_serial_kernel_trace.csv $O/train_step_kernel_trace.csv
# per-stream timeline of the pipelined step
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/proftl -o tl -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra-configs > $O/tl_bench.json 2> $O/tl.err)
find $O/proftl -name "*kernel_trace.csv" -exec cp {} $O/kt.csv \;
rm -rf $O/proftl
gzip -f $O/kt.csv; python tools/step_timeline.py $O/kt.csv.gz 40 > $O/step_timeline.txt 2>&1; rm -f $O/kt.csv.gz
timeout 200 python tools/s3_time.py > $O/s3_time.txt 2>&1
timeout 200 python tools/ms_beside_mfma.py 4 > $O/ms_beside_mfma.txt 2>&1
# round 6: the vote stage's iteration rate by batch shape and segment layout, on synthetic and on the headline's own votes;
# FPS one wave per cloud vs one wave per slot
(timeout 200 python tools/ms_rate.py sgpr; timeout 200 python tools/ms_rate_real2.py) > $O/ms_rate.txt 2>&1
timeout 200 python tools/fps_time.py --frames 1,8,64 > $O/fps_time.txt 2>&1
# split GEMM: LDS-DMA kernel vs the register-staged one (time, bits), its phases in cycles and the shader clock under load
timeout 300 python tools/sg_time.py > $O/sg_time.txt 2>&1
bash tools/sg_variants.sh 8 14 15 > /dev/null 2>&1
(for n in 8 14 15; do echo "== PVN3D_SG_DBG=$n (8: stamps only; 14: + no operand loads, no stores; 15: + no MFMAs either)"; PVN3D_HIP_LIB=tools/sgv/libsg_$n.so timeout 200 python tools/sg_prof.py 2>&1 | grep -v amdgpu.ids; done) > $O/sg_prof.txt 2>&1
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -14 $O/pmc_mfma.log; tail -3 $O/pmc_train.log; head -c 300 $O/bench_default.json; echo; tail -2 $O/bench_default.err
