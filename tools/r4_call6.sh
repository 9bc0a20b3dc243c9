#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c6; mkdir -p $O
timeout 600 bash $R/tools/pmc_attn.sh 257 128 $O/r4_attention_S257_B128.txt > /dev/null 2>&1
cat $O/r4_attention_S257_B128.txt
