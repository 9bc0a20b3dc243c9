#!/bin/bash
# round 4, call 28: 40 optimiser steps on one fixed batch, reference arithmetic against skip_dropped_branches: loss curves
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c28; mkdir -p $O
cd $R
for mode in "" "--skip-dropped"; do
  timeout 500 python bench.py --steps 40 --warmup 0 --no-cpu-baseline --no-power-probe --no-profile --no-skip-leg --loss-curve $mode > $O/curve_${mode#--}.txt 2>&1
  tail -1 $O/curve_${mode#--}.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']['loss_curve']; print('$mode', d['ms_per_step'], [c[i] for i in (0,1,2,4,9,19,29,39)])" || tail -5 $O/curve_${mode#--}.txt
done
