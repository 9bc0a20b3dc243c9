#!/bin/bash
# round 3: the PMC passes of the final code (GPU box).  Outputs under gpurun_out/r3pmc/ (copy to profiles/pmc/ and profiles/).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3pmc; mkdir -p $O
export PROBE_B=128
timeout 600 bash $R/tools/pmc_gemm.sh qkv gemm256v $O/r3_gemm256v_nt_qkv_b128.txt > /dev/null 2>&1
timeout 600 bash $R/tools/pmc_gemm.sh ffn2 gemm256v $O/r3_gemm256v_nt_ffn2_resid_b128.txt > /dev/null 2>&1
timeout 600 bash $R/tools/pmc_gemm.sh wgrad gemm256 $O/r3_gemm256_tn_wgrad_b128.txt > /dev/null 2>&1
timeout 900 bash $R/tools/pmc_bench_traffic.sh $O/r3_gemm_hbm_traffic.json > $O/traffic_log.txt 2>&1
tail -30 $O/traffic_log.txt
for f in $O/*.txt; do echo "== $f"; head -40 $f; done
