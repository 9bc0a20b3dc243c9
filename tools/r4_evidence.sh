#!/bin/bash
# round 4 evidence: kernel trace + last-step summary of the headline step at HEAD, GEMM HBM traffic, bench lines of the other configs
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4ev; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_r4
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r4 -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
KT=$(find /tmp/prof_r4 -name "*kernel_trace.csv" | head -1); ST=$(find /tmp/prof_r4 -name "*kernel_stats.csv" | head -1)
cp $ST $O/r4_bench_kernel_stats_b128.csv
python $R/tools/trace_summary.py $KT $O/r4_bench_last_step_b128.json 1 > $O/r4_bench_last_step_b128.txt 2>&1
head -40 $O/r4_bench_last_step_b128.txt
timeout 600 bash $R/tools/pmc_bench_traffic.sh $O/r4_gemm_hbm_traffic.json > $O/traffic_log.txt 2>&1; tail -25 $O/traffic_log.txt
cd $R
for c in 1 2 4; do
  timeout 400 python bench.py --config $c --steps 4 --warmup 1 --no-cpu-baseline > $O/r4_bench_config${c}_1gpu.json 2> $O/bench_config$c.err
  tail -1 $O/r4_bench_config${c}_1gpu.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config $c', d['ms_per_step'], d['value'], d['unit'], d['roofline']['frac'] if d.get('roofline') else None)" || tail -3 $O/bench_config$c.err
done
timeout 400 python bench.py --config 4 --fp8 --steps 4 --warmup 1 --no-cpu-baseline > $O/r4_bench_config4_fp8_1gpu.json 2> $O/bench_config4fp8.err
tail -1 $O/r4_bench_config4_fp8_1gpu.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config 4 fp8', d['ms_per_step'], d['value'])" || tail -3 $O/bench_config4fp8.err
timeout 400 python bench.py --objective pretrain-vl --steps 4 --warmup 1 --no-cpu-baseline > $O/r4_bench_pretrain_vl_1gpu.json 2> $O/bench_pvl.err
tail -1 $O/r4_bench_pretrain_vl_1gpu.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pretrain-vl', d['ms_per_step'], d['value'])" || tail -3 $O/bench_pvl.err
