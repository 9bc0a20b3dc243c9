#!/bin/bash
# round 5, call 16: the qkv launch split so that its 256 x 256 tiles are whole rounds (ONEPEACE_QKV_ROUND_SPLIT=1): same-box A/B
R=$GRAFT_REPO_ROOT; d=$R/gpurun_out/r5c16; mkdir -p $d
cd $R
B="--steps 8 --warmup 2 --no-cpu-baseline --no-power-probe --no-skip-leg"
for v in 0 1 0 1; do
  ONEPEACE_QKV_ROUND_SPLIT=$v timeout 400 python bench.py $B > $d/bench_${v}_$(date +%s).txt 2> $d/bench_$v.err; tail -1 $(ls -t $d/bench_${v}_*.txt | head -1) | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print('headline round-split=$v', round(d['ms_per_step'],1), round(d['value'],1), 'gemm', round(r.get('frac',0),4), 'launches', r.get('launches'), 'loss', d['config'].get('final_loss'))" || tail -5 $d/bench_$v.err
done
