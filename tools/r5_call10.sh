#!/bin/bash
# round 5, call 10: the whole GPU suite + smoke at HEAD (layer-scale gradient from the weight gradient on by default)
R=$GRAFT_REPO_ROOT; d=$R/gpurun_out/r5c10; mkdir -p $d
cd $R
( time timeout 2000 python -m pytest tests -m gpu -q ) > $d/pytest.txt 2>&1; tail -12 $d/pytest.txt
timeout 300 python __graft_entry__.py smoke > $d/smoke.txt 2>&1; tail -2 $d/smoke.txt
