#!/bin/bash
# round 5, call 15: op_gemm_nt_batched (the 16 groups of the audio positional convolution as one launch): tests, audio model tests, same-box A/B of the step
R=$GRAFT_REPO_ROOT; d=$R/gpurun_out/r5c15; mkdir -p $d
cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "batched or audio or conv or gemm_nt or gemm_full" > $d/pytest_ops.txt 2>&1; tail -3 $d/pytest_ops.txt
timeout 1200 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "audio or micro or lock_step_pass_matches or deep_text" > $d/pytest_model.txt 2>&1; tail -3 $d/pytest_model.txt
B="--steps 8 --warmup 2 --no-cpu-baseline --no-power-probe --no-skip-leg"
for v in prev head prev head; do
  if [ $v = prev ]; then cd $R/ab_prev; else cd $R; fi
  timeout 400 python bench.py $B > $d/bench_${v}_$(date +%s).txt 2> $d/bench_$v.err; tail -1 $(ls -t $d/bench_${v}_*.txt | head -1) | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print('headline $v', round(d['ms_per_step'],1), round(d['value'],1), 'gemm', round(r.get('frac',0),4), 'launches', r.get('launches'), 'loss', d['config'].get('final_loss'))" || tail -5 $d/bench_$v.err
done
