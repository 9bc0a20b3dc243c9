#!/bin/bash
# round 4, call 25: pretrain-vl criterion with its first two passes in lock-step -- fixture parity, bench
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c25; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -k "pretrain" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 500 python bench.py --objective pretrain-vl --steps 4 --warmup 2 --no-cpu-baseline --no-power-probe > $O/bench_pvl.txt 2>&1
tail -1 $O/bench_pvl.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pretrain-vl', d['ms_per_step'], d['value'], d['roofline']['launches'])" || tail -5 $O/bench_pvl.txt
