#!/bin/bash
# round 4, last call: the default bench line at HEAD (incl. cpu_baseline, power probe, skip leg) and the kernel trace of the headline step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4final; mkdir -p $O
cd $R
timeout 600 python bench.py > $O/bench_default.txt 2>&1
tail -1 $O/bench_default.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['cpu_baseline']['value'], d['skip_dropped_branches'].get('ms_per_step'))" || tail -5 $O/bench_default.txt
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_r4f
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r4f -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-power-probe --no-skip-leg > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
KT=$(find /tmp/prof_r4f -name "*kernel_trace.csv" | head -1); ST=$(find /tmp/prof_r4f -name "*kernel_stats.csv" | head -1)
cp $ST $O/r4_bench_kernel_stats_final_b128.csv
python $R/tools/trace_summary.py $KT $O/r4_bench_last_step_final_b128.json 1 > $O/r4_bench_last_step_final_b128.txt 2>&1
head -3 $O/r4_bench_last_step_final_b128.txt
