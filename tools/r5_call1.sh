#!/bin/bash
# round 5, call 1: GPU suite + smoke at the new HEAD; grouped weight-gradient launch A/B (wave queues vs round 4's queues) with PMC fetch
# bytes; whole-step A/B of the two GEMM states (one box)
R=$GRAFT_REPO_ROOT; d=$R/gpurun_out/r5c1; mkdir -p $d
cd $R
R4=$R/one-peace_amd/lib/libonepeace_hip_r4gemm.so
timeout 1500 python -m pytest tests -m gpu -x -q > $d/pytest.txt 2>&1; tail -6 $d/pytest.txt
timeout 300 python __graft_entry__.py smoke > $d/smoke.txt 2>&1; tail -2 $d/smoke.txt
for r in 1 2; do
  timeout 200 python tools/wgrad_grouped_bench.py --nwg 0,1024 --iters 10 > $d/wgrad_new_$r.txt 2>&1
  ONEPEACE_HIP_LIB=$R4 timeout 200 python tools/wgrad_grouped_bench.py --nwg 0 --iters 10 > $d/wgrad_r4_$r.txt 2>&1
done
grep -H "grouped" $d/wgrad_*.txt
cd /tmp; export TMPDIR=/tmp
for lib in new r4; do
  [ $lib = r4 ] && export ONEPEACE_HIP_LIB=$R4 || unset ONEPEACE_HIP_LIB
  out=$d/pmc_wgrad_$lib.txt; : > $out
  for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    n=$(echo $set | cut -c1-12 | tr " " "_")
    rm -rf /tmp/pmcg_$n
    timeout 200 rocprofv3 --pmc $set --kernel-trace --kernel-include-regex "tn_grouped" --output-format csv -d /tmp/pmcg_$n -o p -- python $R/tools/wgrad_grouped_bench.py --iters 2 > /tmp/pmcg_$n.log 2>&1
    f=$(find /tmp/pmcg_$n -name "*counter_collection.csv" | head -1)
    if [ -z "$f" ]; then echo "# pass '$set' produced no counters: $(tail -2 /tmp/pmcg_$n.log | tr '\n' ' ')" >> $out; continue; fi
    python - "$f" >> $out <<PY
import csv,sys
from collections import defaultdict
agg=defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "tn_grouped" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()): print("gemm256w_tn_grouped_kernel  %-30s n=%d avg=%.5g" % (k, len(v), sum(v)/len(v)))
PY
  done
  echo "== $lib"; cat $out
done
unset ONEPEACE_HIP_LIB
cd $R
for lib in new r4 new r4; do
  [ $lib = r4 ] && export ONEPEACE_HIP_LIB=$R4 || unset ONEPEACE_HIP_LIB
  timeout 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-power-probe --no-skip-leg > $d/bench_$lib.txt 2>&1
  tail -1 $d/bench_$lib.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['ms_per_step'], d['value'], d['roofline']['frac'])" || tail -5 $d/bench_$lib.txt
done
