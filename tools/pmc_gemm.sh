#!/bin/bash
# PMC passes for one GEMM shape (run on the GPU box):  tools/pmc_gemm.sh qkv|geglu|ffn2|wgrad <kernel-substring> <out-file>
which=$1; kern=$2; out=$3
cd /tmp; export TMPDIR=/tmp
: > $out
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_UNALIGNED_STALL"; do
  n=$(echo $set | cut -c1-12 | tr " " "_")
  rm -rf /tmp/pmc_$n
  rocprofv3 --pmc $set --kernel-trace --kernel-include-regex "gemm" --output-format csv -d /tmp/pmc_$n -o p -- python $GRAFT_REPO_ROOT/tools/gemm_probe.py $which 0 3 ${PROBE_B:-64} > /dev/null 2>&1
  f=$(find /tmp/pmc_$n -name "*counter_collection.csv" | head -1)
  python - "$f" "$kern" >> $out <<PY
import csv,sys
from collections import defaultdict
agg=defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()): print("%-28s n=%d avg=%.1f" % (k, len(v), sum(v)/len(v)))
PY
done
cat $out
