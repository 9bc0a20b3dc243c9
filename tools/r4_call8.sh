#!/bin/bash
# round 4, call 8: the clock the shader runs at inside the grouped weight-gradient launch (s_memtime against s_memrealtime) + rocm-smi beside a GEMM loop
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c8; mkdir -p $O
cd $R
timeout 300 python tools/wgrad_grouped_timeline.py > $O/timeline.txt 2>&1; grep -v "start skew" $O/timeline.txt | tail -24
( for i in $(seq 1 12); do /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power\|fclk\|mclk" ; echo ---; sleep 0.5; done ) > $O/smi.txt 2>&1 &
SMI=$!
timeout 200 python tools/wgrad_grouped_bench.py --iters 200 > $O/wgrad_bench_long.txt 2>&1
wait $SMI
head -5 $O/wgrad_bench_long.txt; cat $O/smi.txt | head -60
