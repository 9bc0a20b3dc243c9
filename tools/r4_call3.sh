#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c3; mkdir -p $O
timeout 300 python tools/wgrad_grouped_timeline.py 0 > $O/timeline_nwg256.txt 2>&1; cat $O/timeline_nwg256.txt
timeout 300 python tools/wgrad_grouped_timeline.py 240 > $O/timeline_nwg240.txt 2>&1; cat $O/timeline_nwg240.txt
