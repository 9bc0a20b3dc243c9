#!/bin/bash
# round 4: the whole GPU test suite + smoke + a short bench of the headline config in one call (argument: output sub-directory)
d=gpurun_out/${1:-r4}
mkdir -p $d
timeout 1500 python -m pytest tests -m gpu -x -q > $d/pytest.txt 2>&1; tail -15 $d/pytest.txt
timeout 300 python __graft_entry__.py smoke > $d/smoke.txt 2>&1; tail -2 $d/smoke.txt
timeout 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $d/bench.txt 2>&1
tail -1 $d/bench.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['launches'])" || tail -5 $d/bench.txt
