#!/bin/bash
# PMC passes over ONE lock-step layer's grouped weight-gradient launch (run on the GPU box):  tools/pmc_wgrad_grouped.sh <out-file>
out=$1
cd /tmp; export TMPDIR=/tmp
: > $out
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_UNALIGNED_STALL" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU"; do
  n=$(echo $set | cut -c1-12 | tr " " "_")
  rm -rf /tmp/pmcg_$n
  timeout 200 rocprofv3 --pmc $set --kernel-trace --kernel-include-regex "tn_grouped" --output-format csv -d /tmp/pmcg_$n -o p -- python $GRAFT_REPO_ROOT/tools/wgrad_grouped_bench.py --iters 2 > /tmp/pmcg_$n.log 2>&1
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
cat $out
