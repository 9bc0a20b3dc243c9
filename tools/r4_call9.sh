#!/bin/bash
# round 4, call 9: grouped weight-gradient launch, epilogue from registers (round 4's first, 8-byte accesses) against through LDS -- alternating processes on one box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c9; mkdir -p $O
cd $R
for rep in 1 2 3; do
  for v in oldepi new; do
    lib=$R/one-peace_amd/lib/libonepeace_hip.so; [ $v = oldepi ] && lib=$R/one-peace_amd/lib/libonepeace_hip_oldepi.so
    echo "== $v (rep $rep)" >> $O/ab.txt
    ONEPEACE_HIP_LIB=$lib timeout 100 python tools/wgrad_grouped_bench.py --iters 100 2>&1 | grep "round" >> $O/ab.txt
  done
done
cat $O/ab.txt
