#!/bin/bash
# round 5, call 8: layer-scale gradient from the weight gradient (no branch output y): new op tests, model tests that touch the branch functions, A/B of the step
R=$GRAFT_REPO_ROOT; d=$R/gpurun_out/r5c8; mkdir -p $d
cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "row_dot or gamma_grad or resid_bwd or gemm_tn_grouped or weight_gradients_of_a_headline" > $d/pytest_ops.txt 2>&1; tail -5 $d/pytest_ops.txt
timeout 1200 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "layer_scale_gradient or recompute_cheap or lock_step or direct_gradient or fused_layer_with or all_hiddens or stage2 or micro or pair_criterions" > $d/pytest_model.txt 2>&1; tail -5 $d/pytest_model.txt
timeout 300 python __graft_entry__.py smoke > $d/smoke.txt 2>&1; tail -2 $d/smoke.txt
B="--steps 6 --warmup 2 --no-cpu-baseline --no-power-probe --no-skip-leg"
show() { tail -1 $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; m=d['config'].get('memory') or {}; print('$2', round(d['ms_per_step'],1), round(d['value'],1), 'gemm', round(r.get('frac',0),4), 'peak GB', m.get('peak_reserved_gb'), 'loss', d['config'].get('final_loss'))" || tail -5 $1; }
for v in 0 1 0 1; do
  ONEPEACE_DGAMMA_FROM_WGRAD=$v timeout 400 python bench.py $B > $d/bench_dg${v}_$(date +%s).txt 2> $d/bench_dg$v.err; show $(ls -t $d/bench_dg${v}_*.txt | head -1) "headline dgamma-from-wgrad=$v"
done
tail -3 $d/bench_dg1.err | grep -v amdgpu
