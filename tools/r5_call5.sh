#!/bin/bash
# round 5, call 5: fused fp8 outputs of the LayerNorm kernels + fp8 in the lock-step pass: tests, then BASELINE configs[4] with / without --fp8
R=$GRAFT_REPO_ROOT; d=$R/gpurun_out/r5c5; mkdir -p $d
cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -x -q -k "fp8 or layernorm or ln_geglu or graph_replay" > $d/pytest_fp8.txt 2>&1; tail -4 $d/pytest_fp8.txt
for v in bf16 fp8 bf16 fp8; do
  extra=""; [ $v = fp8 ] && extra="--fp8"
  timeout 500 python bench.py --config 4 --steps 20 --warmup 3 --no-cpu-baseline --no-power-probe --no-skip-leg $extra > $d/bench_config4_$v.txt 2>&1
  tail -1 $d/bench_config4_$v.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print('config 4 $v', d['ms_per_step'], d['value'], r.get('frac'), r.get('fp8_gemm'), d['config'].get('final_loss'))" || tail -5 $d/bench_config4_$v.txt
done
for v in bf16 fp8; do
  extra=""; [ $v = fp8 ] && extra="--fp8"
  timeout 500 python bench.py --config 4 --res 512 --steps 8 --warmup 3 --no-cpu-baseline --no-power-probe --no-skip-leg $extra > $d/bench_config4_512_$v.txt 2>&1
  tail -1 $d/bench_config4_512_$v.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print('config 4 512 $v', d['ms_per_step'], d['value'], r.get('frac'), r.get('fp8_gemm'), d['config'].get('final_loss'))" || tail -5 $d/bench_config4_512_$v.txt
done
