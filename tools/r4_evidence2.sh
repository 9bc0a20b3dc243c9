#!/bin/bash
# round 4, final evidence at HEAD: kernel trace + last-step summary of the headline step, GEMM HBM traffic (PMC), attention PMC at S = 257
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4ev2; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_r4
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r4 -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-power-probe --no-skip-leg > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
KT=$(find /tmp/prof_r4 -name "*kernel_trace.csv" | head -1); ST=$(find /tmp/prof_r4 -name "*kernel_stats.csv" | head -1)
cp $ST $O/r4_bench_kernel_stats_head_b128.csv
python $R/tools/trace_summary.py $KT $O/r4_bench_last_step_head_b128.json 1 > $O/r4_bench_last_step_head_b128.txt 2>&1
head -32 $O/r4_bench_last_step_head_b128.txt
timeout 600 bash $R/tools/pmc_bench_traffic.sh $O/r4_gemm_hbm_traffic.json > $O/traffic_log.txt 2>&1; tail -6 $O/traffic_log.txt
timeout 600 bash $R/tools/pmc_attn.sh 257 128 $O/r4_attention_S257_B128_head.txt > /dev/null 2>&1
grep "dkdv_pers\|dq_pers" $O/r4_attention_S257_B128_head.txt | head -30
