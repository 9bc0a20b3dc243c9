#!/bin/bash
# round 4, call 30: 16 grouped problems (audio positional convolution weight gradients) -- parity, step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c30; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -k "tn_grouped or audio or conv or micro_model" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
bash tools/r4_step.sh r4c30 --no-power-probe --no-skip-leg
