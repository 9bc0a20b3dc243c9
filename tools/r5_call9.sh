#!/bin/bash
R=$GRAFT_REPO_ROOT; d=$R/gpurun_out/r5c9; mkdir -p $d
cd $R
timeout 600 python tools/torch_prof_aten.py 64 > $d/aten.txt 2>&1; grep -v amdgpu.ids $d/aten.txt | tail -75
timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "recompute_cheap" 2>&1 | tail -2
