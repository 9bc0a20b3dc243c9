#!/bin/bash
# round 4, call 21: persistent dK/dV kernel -- attention parity tests, A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c21; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "attention or attn" > $O/pytest.txt 2>&1; tail -12 $O/pytest.txt
timeout 300 python tools/attn_pers_ab.py > $O/ab.txt 2>&1; grep -v amdgpu $O/ab.txt | tail -16
