#!/bin/bash
# round 4, call 7: LDS-transposed epilogue of the grouped weight-gradient kernel -- parity, per-layer bench, whole step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c7; mkdir -p $O
cd $R
timeout 500 python -m pytest tests/test_ops_gpu.py -x -q -k "tn_grouped or headline_layer_grouped or grouped_gemm_at" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 200 python tools/wgrad_grouped_bench.py > $O/wgrad_bench.txt 2>&1; tail -6 $O/wgrad_bench.txt
bash tools/r4_step.sh r4c7
