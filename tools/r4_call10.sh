#!/bin/bash
# round 4, call 10: register-only MFMA loop -- what the package sustains at its power limit (tools/probes/mfma_power_probe.hip)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c10; mkdir -p $O
cd $R
( for i in $(seq 1 40); do /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power (W)" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 0.5; done ) > $O/smi.txt 2>&1 &
SMI=$!
for cfg in "4 0 2.0" "4 1 2.0" "4 2 2.0" "8 0 2.0" "4 0 2.0"; do
  timeout 60 tools/probes/mfma_power_probe $cfg >> $O/probe.txt 2>&1
  sleep 1
done
wait $SMI
cat $O/probe.txt; cat $O/smi.txt
