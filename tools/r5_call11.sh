#!/bin/bash
# round 5, call 11: slice copies instead of torch.cat in the adapters / the lock-step packing, one gradient placeholder per bias handle: model tests, bench
R=$GRAFT_REPO_ROOT; d=$R/gpurun_out/r5c11; mkdir -p $d
cd $R
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_distributed_gpu.py -m gpu -x -q > $d/pytest_model.txt 2>&1; tail -4 $d/pytest_model.txt
timeout 300 python __graft_entry__.py smoke > $d/smoke.txt 2>&1; tail -1 $d/smoke.txt
B="--steps 6 --warmup 2 --no-cpu-baseline --no-power-probe --no-skip-leg"
for i in 1 2; do
  timeout 400 python bench.py $B > $d/bench_$i.txt 2> $d/bench_$i.err; tail -1 $d/bench_$i.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print('headline', round(d['ms_per_step'],1), round(d['value'],1), 'gemm', round(r.get('frac',0),4), 'loss', d['config'].get('final_loss'))" || tail -5 $d/bench_$i.err
done
