#!/bin/bash
mkdir -p gpurun_out/r3c8
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r3c8/pytest.txt 2>&1; tail -5 gpurun_out/r3c8/pytest.txt
timeout 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/r3c8/bench.txt 2>&1
tail -1 gpurun_out/r3c8/bench.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['frac'])"
