#!/bin/bash
# round 5, call 3: GPU suite after the flavour removal / persistent-by-default short-K launches; torch profiler view of a step (which
# torch-native ops are left between the HIP kernels); the other BASELINE configs on this code (regression check against profiles/r4_bench_config*)
R=$GRAFT_REPO_ROOT; d=$R/gpurun_out/r5c3; mkdir -p $d
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $d/pytest.txt 2>&1; tail -4 $d/pytest.txt
timeout 300 python tools/torch_prof.py > $d/torch_prof.txt 2>&1; grep -v "^-" $d/torch_prof.txt | head -50
for c in 1 2 4; do
  timeout 500 python bench.py --config $c --steps 4 --warmup 2 --no-cpu-baseline --no-power-probe --no-skip-leg > $d/bench_config$c.txt 2>&1
  tail -1 $d/bench_config$c.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config $c', d['ms_per_step'], d['value'], (d.get('roofline') or {}).get('frac'))" || tail -5 $d/bench_config$c.txt
done
