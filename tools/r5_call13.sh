#!/bin/bash
R=$GRAFT_REPO_ROOT; d=$R/gpurun_out/r5c13; mkdir -p $d
cd $R
for c in full elementwise gemm; do timeout 500 python tools/overlap_probe.py --chain $c 2>&1 | grep -v amdgpu.ids | tee -a $d/overlap.txt; done
