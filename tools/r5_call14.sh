#!/bin/bash
# round 5, call 14: PMC traffic of the elementwise kernels at HEAD (resid_bwd no longer reads y) + whole GPU suite at HEAD
R=$GRAFT_REPO_ROOT; d=$R/gpurun_out/r5c14; mkdir -p $d
cd $R
timeout 900 bash tools/pmc_elementwise_traffic.sh $d/r5_elementwise_traffic.txt > $d/pmc_elem.log 2>&1; cat $d/r5_elementwise_traffic.txt || tail -5 $d/pmc_elem.log
( time timeout 2000 python -m pytest tests -m gpu -q ) > $d/pytest.txt 2>&1; tail -6 $d/pytest.txt
