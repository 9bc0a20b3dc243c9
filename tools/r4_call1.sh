#!/bin/bash
# round 4, first contact of the grouped weight-gradient launch: parity, micro-benchmark, model tests, whole-step A/B
d=gpurun_out/r4c1
mkdir -p $d
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "tn_grouped or headline_layer_grouped or weight_gradient" > $d/pytest_ops.txt 2>&1; tail -5 $d/pytest_ops.txt
timeout 300 python tools/wgrad_grouped_bench.py --nwg 0,248,240 --subsets > $d/wgrad_grouped_bench.txt 2>&1; cat $d/wgrad_grouped_bench.txt
timeout 900 python -m pytest tests/test_model_gpu.py -x -q > $d/pytest_model.txt 2>&1; tail -5 $d/pytest_model.txt
for v in 0 1 0 1; do
  ONEPEACE_GROUPED_WGRAD=$v timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $d/bench_g$v.txt 2>&1
  tail -1 $d/bench_g$v.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('grouped=$v', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['launches'])" || tail -5 $d/bench_g$v.txt
done
