#!/bin/bash
# round 5, evidence call at HEAD: whole GPU suite + smoke; the default bench line (cpu_baseline, power probe, skip leg); kernel trace of
# the headline step + last-step summary; PMC traffic of the GEMM family; fp8 kernel rates after the epilogue change and the config-4 pair
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r5final}; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 300 python __graft_entry__.py smoke > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 600 python bench.py > $O/bench_default.txt 2> $O/bench_default.err
tail -1 $O/bench_default.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline'].get('power_limited_peak'), d['cpu_baseline']['value'], d['skip_dropped_branches'].get('ms_per_step'))" || tail -5 $O/bench_default.err
timeout 300 python tools/fp8_bench.py > $O/fp8_bench.txt 2>&1; grep "M=50240" $O/fp8_bench.txt
for v in bf16 fp8; do
  extra=""; [ $v = fp8 ] && extra="--fp8"
  timeout 500 python bench.py --config 4 --steps 8 --warmup 3 --no-cpu-baseline --no-power-probe --no-skip-leg $extra > $O/bench_config4_$v.txt 2>&1
  tail -1 $O/bench_config4_$v.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print('config 4 $v', d['ms_per_step'], d['value'], r.get('fp8_gemm'))" || tail -5 $O/bench_config4_$v.txt
done
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_r5f
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r5f -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-power-probe --no-skip-leg > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
KT=$(find /tmp/prof_r5f -name "*kernel_trace.csv" | head -1); ST=$(find /tmp/prof_r5f -name "*kernel_stats.csv" | head -1)
cp $ST $O/r5_bench_kernel_stats_final_b128.csv
python $R/tools/trace_summary.py $KT $O/r5_bench_last_step_final_b128.json 1 > $O/r5_bench_last_step_final_b128.txt 2>&1
head -3 $O/r5_bench_last_step_final_b128.txt
cd $R
timeout 900 bash tools/pmc_bench_traffic.sh $O/r5_gemm_hbm_traffic.json > $O/pmc_traffic.log 2>&1; python -c "
import json; d=json.load(open('$O/r5_gemm_hbm_traffic.json')); print('traffic per launch', d['bytes_per_launch'], {k: round(v/1e9,2) for k,v in d['by_kernel_read_bytes_per_launch'].items()}, d['by_kernel_launches'])" || tail -5 $O/pmc_traffic.log
