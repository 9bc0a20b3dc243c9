#!/bin/bash
# round 4, call 2: kernel trace of the step with the grouped weight-gradient launch + PMC passes of grouped vs separate launches
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c2; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for v in 1 0; do
  rm -rf /tmp/prof_g$v
  ONEPEACE_GROUPED_WGRAD=$v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_g$v -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile > $O/bench_under_rocprof_g$v.json 2> $O/bench_under_rocprof_g$v.err
  KT=$(find /tmp/prof_g$v -name "*kernel_trace.csv" | head -1)
  ST=$(find /tmp/prof_g$v -name "*kernel_stats.csv" | head -1)
  cp $ST $O/bench_kernel_stats_g$v.csv
  python $R/tools/trace_summary.py $KT $O/bench_last_step_g$v.json 1 > $O/trace_summary_g$v.txt 2>&1
  head -45 $O/trace_summary_g$v.txt
done
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  n=$(echo $set | cut -c1-12 | tr " " "_")
  rm -rf /tmp/pmc_$n
  timeout 300 rocprofv3 --pmc $set --kernel-trace --kernel-include-regex "tn_|splitk" --output-format csv -d /tmp/pmc_$n -o p -- python $R/tools/wgrad_grouped_bench.py --iters 1 > /dev/null 2>&1
  f=$(find /tmp/pmc_$n -name "*counter_collection.csv" | head -1)
  python - "$f" >> $O/pmc_wgrad_grouped_vs_separate.txt <<PY
import csv,sys
from collections import defaultdict
agg=defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"].split("(")[0][-40:]
    agg[(k,r["Counter_Name"])].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()): print("%-42s %-28s n=%d avg=%.4g sum=%.4g" % (k[0], k[1], len(v), sum(v)/len(v), sum(v)))
PY
done
cat $O/pmc_wgrad_grouped_vs_separate.txt
