# Where a small-batch step (16 tuples) spends its time: rocprofv3 kernel trace of the last step.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r2b16; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_b16
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b16 -o bench -- python $R/bench.py --batch 16 --steps 2 --warmup 1 --no-cpu-baseline --no-profile > $OUT/bench.json 2> $OUT/bench.err
python $R/tools/trace_summary.py $(find /tmp/prof_b16 -name "*kernel_trace.csv" | head -1) $OUT/last_step.json 1 > $OUT/trace_summary.txt 2>&1
head -45 $OUT/trace_summary.txt
