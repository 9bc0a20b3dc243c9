#!/bin/bash
# round 4: whole-step bench (argument: output sub-directory; extra args go to bench.py)
d=gpurun_out/${1:-r4step}; shift
mkdir -p $d
timeout 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline "$@" > $d/bench.txt 2>&1
tail -1 $d/bench.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['launches'])" || tail -5 $d/bench.txt
