#!/bin/bash
# round 4, call 17: dK/dV with the lone key block split by queries -- attention parity tests, A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c17; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "attention or attn" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 300 python tools/attn_pers_ab.py > $O/ab.txt 2>&1; grep -v amdgpu $O/ab.txt
