#!/bin/bash
# round 4, call 12: register-only MFMA loop, 16x16x32 against 32x32x16, one and two waves per SIMD, random and zero operands
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c12; mkdir -p $O
cd $R
( for i in $(seq 1 64); do /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power (W)" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 0.5; done ) > $O/smi.txt 2>&1 &
SMI=$!
for cfg in "4 0 2.0 16" "4 0 2.0 32" "8 0 2.0 16" "8 0 2.0 32" "4 1 2.0 16" "4 1 2.0 32" "8 1 2.0 16" "8 1 2.0 32"; do
  timeout 60 tools/probes/mfma_power_probe $cfg >> $O/probe.txt 2>&1
  sleep 1
done
wait $SMI
cat $O/probe.txt; cat $O/smi.txt | tr '\n' ';'
