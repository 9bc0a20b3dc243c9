#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c5; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "attention" > $O/pytest_attn.txt 2>&1; tail -15 $O/pytest_attn.txt
timeout 300 python tools/attn_pers_ab.py 128 > $O/attn_pers_ab.txt 2>&1; cat $O/attn_pers_ab.txt
