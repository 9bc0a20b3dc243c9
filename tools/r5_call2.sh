#!/bin/bash
# round 5, call 2: hipBLASLt comparison at HEAD (profiles/r5_blas_compare_headline.txt); single-problem NT launches on the persistent
# kernel (sched 6) now that its tile walk has no scratch traffic: per launch and on the whole step
R=$GRAFT_REPO_ROOT; d=$R/gpurun_out/r5c2; mkdir -p $d
cd $R
ITERS=30 ROUNDS=3 timeout 400 python tools/blas_compare.py > $d/blas_default.txt 2>&1; tail -4 $d/blas_default.txt
ONEPEACE_TUNE_SCHED=6 ITERS=30 ROUNDS=3 timeout 400 python tools/blas_compare.py > $d/blas_sched6.txt 2>&1; tail -4 $d/blas_sched6.txt
paste <(grep "ours" $d/blas_default.txt | awk -F'ours' '{print $1}' | cut -c1-58) <(grep "ours" $d/blas_default.txt | sed 's/.*ours \([0-9.]*\) ms.*/\1/') <(grep "ours" $d/blas_sched6.txt | sed 's/.*ours \([0-9.]*\) ms.*/\1/')
for v in 0 6 0 6; do
  ONEPEACE_TUNE_SCHED=$v timeout 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-power-probe --no-skip-leg > $d/bench_sched$v.txt 2>&1
  tail -1 $d/bench_sched$v.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('sched $v', d['ms_per_step'], d['value'], d['roofline']['frac'])" || tail -5 $d/bench_sched$v.txt
done
