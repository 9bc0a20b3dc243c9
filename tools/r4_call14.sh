#!/bin/bash
# round 4, call 14: configs 2 and 4 (single-modality passes) with and without --skip-dropped; single-stream parity test
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c14; mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_model_gpu.py -x -q -k "skips_dropped or single_stream_skip" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
for cfg in 4 2; do
  for mode in "" "--skip-dropped"; do
    timeout 500 python bench.py --config $cfg --steps 4 --warmup 2 --no-cpu-baseline --no-power-probe $mode > $O/bench_c${cfg}_${mode#--}.txt 2>&1
    tail -1 $O/bench_c${cfg}_${mode#--}.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config $cfg $mode', d['ms_per_step'], d['value'], d['config'].get('stochastic_depth'))" || tail -5 $O/bench_c${cfg}_${mode#--}.txt
  done
done
