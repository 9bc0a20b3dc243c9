#!/bin/bash
# round 4: rocprofv3 kernel trace of the headline step with skip_dropped_branches (3 timed steps) + last-step summary
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4skip
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_r4s
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r4s -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-power-probe --skip-dropped > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
KT=$(find /tmp/prof_r4s -name "*kernel_trace.csv" | head -1)
ST=$(find /tmp/prof_r4s -name "*kernel_stats.csv" | head -1)
cp $ST $OUT/bench_kernel_stats.csv
python $GRAFT_REPO_ROOT/tools/trace_summary.py $KT $OUT/bench_last_step.json 1 > $OUT/trace_summary.txt 2>&1
head -50 $OUT/trace_summary.txt
