#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c4; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "tn_grouped or headline_layer_grouped" > $O/pytest_ops.txt 2>&1; tail -3 $O/pytest_ops.txt
timeout 300 python tools/wgrad_grouped_timeline.py 0 > $O/timeline_solo.txt 2>&1; cat $O/timeline_solo.txt
timeout 300 python tools/wgrad_grouped_timeline.py 1024 > $O/timeline_nosolo.txt 2>&1; head -4 $O/timeline_nosolo.txt
timeout 300 python tools/wgrad_grouped_bench.py --nwg 0,1024,240 > $O/wgrad_grouped_bench.txt 2>&1; head -12 $O/wgrad_grouped_bench.txt
