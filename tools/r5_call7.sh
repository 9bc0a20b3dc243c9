#!/bin/bash
# round 5, call 7: the whole GPU suite at HEAD (timed), smoke, and what the one-tile-per-workgroup rule of world > 1 costs on an idle GPU
R=$GRAFT_REPO_ROOT; d=$R/gpurun_out/r5c7; mkdir -p $d
cd $R
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $d/pytest.txt 2>&1; tail -8 $d/pytest.txt
timeout 300 python __graft_entry__.py smoke > $d/smoke.txt 2>&1; tail -2 $d/smoke.txt
B="--steps 6 --warmup 2 --no-cpu-baseline --no-power-probe --no-skip-leg"
show() { tail -1 $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print('$2', round(d['ms_per_step'],1), round(d['value'],1), 'gemm', round(r.get('frac',0),4))" || tail -5 $1; }
for v in 0 7 0 7; do
  ONEPEACE_TUNE_SCHED=$v timeout 400 python bench.py $B > $d/bench_sched${v}_$(date +%s).txt 2> $d/bench_sched$v.err; show $(ls -t $d/bench_sched${v}_*.txt | head -1) "headline sched $v"
done
