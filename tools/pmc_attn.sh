#!/bin/bash
# PMC passes over the attention kernels at one shape (run on the GPU box):  tools/pmc_attn.sh <S> <B> <out-file>
S=$1; B=$2; out=$3
cd /tmp; export TMPDIR=/tmp
: > $out
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_UNALIGNED_STALL" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA"; do
  n=$(echo $set | cut -c1-14 | tr " " "_")
  rm -rf /tmp/pmca_$n
  timeout 120 rocprofv3 --pmc $set --kernel-trace --kernel-include-regex "attn_" --output-format csv -d /tmp/pmca_$n -o p -- python $GRAFT_REPO_ROOT/tools/attn_probe.py $S $B 3 > /tmp/pmca_$n.log 2>&1
  f=$(find /tmp/pmca_$n -name "*counter_collection.csv" | head -1)
  if [ -z "$f" ]; then echo "# pass '$set' produced no counters: $(tail -2 /tmp/pmca_$n.log | tr '\n' ' ')" >> $out; continue; fi
  python - "$f" >> $out <<PY
import csv,sys,re
from collections import defaultdict
agg=defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    m=re.search(r"attn_\w+_kernel(?:ILi\d+E|ILb\dE(?:Lb\dE)?)?", r["Kernel_Name"])
    if m: agg[(m.group(0), r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k,c),v in sorted(agg.items()): print("%-34s %-30s n=%d avg=%.4g" % (k, c, len(v), sum(v)/len(v)))
PY
done
cat $out
