#!/bin/bash
# round 5, evidence call of the second session (HEAD: layer-scale gradient from the weight gradient, cheap recompute level, config 2 at that level):
# the default bench line (cpu_baseline, power probe, skip leg); kernel trace of the headline step + last-step summary; PMC traffic of the GEMM
# family; configs 1 / 2 / 4 (+ fp8); hipBLASLt table
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r5final3}; mkdir -p $O
cd $R
timeout 600 python bench.py > $O/bench_default.txt 2> $O/bench_default.err
tail -1 $O/bench_default.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], d['value'], r['frac'], r.get('power_limited_peak'), r['traffic'], r['algorithmic_bytes_per_launch'], d['cpu_baseline']['value'], d['skip_dropped_branches'].get('ms_per_step'), d['config']['memory'])" || tail -5 $O/bench_default.err
for c in 1 2 4; do
  timeout 500 python bench.py --config $c --steps 8 --warmup 3 --no-cpu-baseline --no-power-probe --no-skip-leg > $O/bench_config$c.txt 2> $O/bench_config$c.err
  tail -1 $O/bench_config$c.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print('config $c', d['ms_per_step'], d['value'], r.get('frac'), d['config'].get('activation_recompute','')[:30], d['config'].get('final_loss'))" || tail -5 $O/bench_config$c.err
done
timeout 500 python bench.py --config 4 --fp8 --steps 8 --warmup 3 --no-cpu-baseline --no-power-probe --no-skip-leg > $O/bench_config4_fp8.txt 2> $O/bench_config4_fp8.err
tail -1 $O/bench_config4_fp8.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print('config 4 fp8', d['ms_per_step'], d['value'], r.get('fp8_gemm'), d['config'].get('final_loss'))" || tail -5 $O/bench_config4_fp8.err
ITERS=30 ROUNDS=3 timeout 400 python tools/blas_compare.py > $O/blas_compare.txt 2>&1; grep -v amdgpu.ids $O/blas_compare.txt | tail -22
Y2=1 ITERS=30 ROUNDS=3 timeout 400 python tools/blas_compare.py 2>&1 | grep "residual" > $O/blas_compare_with_y.txt; cat $O/blas_compare_with_y.txt
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_r5f
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r5f -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-power-probe --no-skip-leg > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
KT=$(find /tmp/prof_r5f -name "*kernel_trace.csv" | head -1); ST=$(find /tmp/prof_r5f -name "*kernel_stats.csv" | head -1)
cp $ST $O/r5_bench_kernel_stats_final2_b128.csv
python $R/tools/trace_summary.py $KT $O/r5_bench_last_step_final2_b128.json 1 > $O/r5_bench_last_step_final2_b128.txt 2>&1
head -40 $O/r5_bench_last_step_final2_b128.txt | cut -c1-150
cd $R
timeout 900 bash tools/pmc_bench_traffic.sh $O/r5_gemm_hbm_traffic.json > $O/pmc_traffic.log 2>&1; python -c "
import json; d=json.load(open('$O/r5_gemm_hbm_traffic.json')); print('traffic per launch', d['bytes_per_launch'], {k: round(v/1e9,2) for k,v in d['by_kernel_read_bytes_per_launch'].items()}, d['by_kernel_launches'])" || tail -5 $O/pmc_traffic.log
