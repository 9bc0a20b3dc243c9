#!/bin/bash
# round 4, call 13: stochastic depth on the kept samples only -- kernel test, model parity against the multiplier form, bench with the extra leg
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c13; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -k "rows_gather or skips_dropped or lock_step_pass_matches" > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-power-probe > $O/bench.txt 2>&1
tail -1 $O/bench.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['frac']); print(d.get('skip_dropped_branches'))" || tail -20 $O/bench.txt
