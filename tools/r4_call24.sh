#!/bin/bash
# round 4, call 24: pair criterions in lock-step -- parity test, configs 2 and 4 with / without
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c24; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -k "pair_criterions or retrieval or micro" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
for cfg in 4 2; do
  for mode in "--no-lock-step" ""; do
    timeout 500 python bench.py --config $cfg --steps 4 --warmup 2 --no-cpu-baseline --no-power-probe $mode > $O/bench_c${cfg}_${mode#--}.txt 2>&1
    tail -1 $O/bench_c${cfg}_${mode#--}.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config $cfg $mode', d['ms_per_step'], d['value'], d['roofline']['launches'])" || tail -5 $O/bench_c${cfg}_${mode#--}.txt
  done
done
