#!/bin/bash
# round 5, call 12: same-box A/B of the host-side cleanups (torch.cat -> slice copies, one bias-gradient placeholder) against the commit before them
R=$GRAFT_REPO_ROOT; d=$R/gpurun_out/r5c12; mkdir -p $d
B="--steps 8 --warmup 2 --no-cpu-baseline --no-power-probe --no-skip-leg"
for v in prev head prev head; do
  if [ $v = prev ]; then cd $R/ab_prev; else cd $R; fi
  timeout 400 python bench.py $B > $d/bench_${v}_$(date +%s).txt 2> $d/bench_$v.err; tail -1 $(ls -t $d/bench_${v}_*.txt | head -1) | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print('headline $v', round(d['ms_per_step'],1), round(d['value'],1), 'gemm', round(r.get('frac',0),4), 'loss', d['config'].get('final_loss'))" || tail -5 $d/bench_$v.err
done
cd $R; timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "layer_scale_gradient" 2>&1 | tail -2
