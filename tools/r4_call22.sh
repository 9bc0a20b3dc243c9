#!/bin/bash
# round 4, call 22: final persistent dK/dV kernel -- attention tests, A/B, whole step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c22; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "attention or attn" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 300 python tools/attn_pers_ab.py > $O/ab.txt 2>&1; grep "dK / dV kernel" $O/ab.txt
bash tools/r4_step.sh r4c22 --no-power-probe --no-skip-leg
