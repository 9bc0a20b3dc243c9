#!/bin/bash
# round 4, call 18: dK/dV block skip -- attention tests, A/B, 2-rank step with --skip-dropped
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c18; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_distributed_gpu.py -x -q -k "attention or attn or two_rank_step_keeps" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 300 python tools/attn_pers_ab.py 2>&1 | grep -A2 "S=257" > $O/ab.txt; cat $O/ab.txt
