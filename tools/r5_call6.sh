#!/bin/bash
# round 5, call 6 (second session): the cheap recompute level, return_all_hiddens / layerdrop on the fused path, ln_geglu_bwd without its second
# exponential -- tests, kernel A/B against the previous layernorm.hip (lib/libonepeace_hip_lnold.so, built by tools/build_variant.py from
# git show 60d4d92:one-peace_amd/csrc/layernorm.hip), whole-step A/B, the cheap level's cost, config 2 at the cheap level
R=$GRAFT_REPO_ROOT; d=$R/gpurun_out/r5c6; mkdir -p $d
cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -x -q -k "recompute_cheap or all_hiddens or ln_geglu or layernorm or geglu or lock_step_pass_matches or fused_layer_with" > $d/pytest_new.txt 2>&1; tail -4 $d/pytest_new.txt
OLD=$R/one-peace_amd/lib/libonepeace_hip_lnold.so
for v in old new old new; do
  lib=""; [ $v = old ] && lib=$OLD
  ONEPEACE_HIP_LIB=$lib timeout 120 python tools/ln_geglu_sweep.py $v 2>&1 | grep -v amdgpu.ids | tee -a $d/ln_geglu_ab.txt
done
B="--steps 6 --warmup 2 --no-cpu-baseline --no-power-probe --no-skip-leg"
show() { tail -1 $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; m=d['config'].get('memory') or {}; print('$2', round(d['ms_per_step'],1), round(d['value'],1), 'gemm', round(r.get('frac',0),4), 'peak GB', m.get('peak_reserved_gb'), d['config']['activation_recompute'][:40], 'loss', d['config'].get('final_loss'))" || tail -5 $1; }
for v in old new old new; do
  lib=""; [ $v = old ] && lib=$OLD
  ONEPEACE_HIP_LIB=$lib timeout 400 python bench.py $B > $d/bench_$v.txt 2> $d/bench_$v.err; show $d/bench_$v.txt "headline $v"
  cp $d/bench_$v.txt $d/bench_${v}_$(date +%s).txt
done
timeout 400 python bench.py $B --recompute-cheap > $d/bench_cheap.txt 2> $d/bench_cheap.err; show $d/bench_cheap.txt "headline cheap"
timeout 500 python bench.py --config 2 --steps 6 --warmup 2 --no-cpu-baseline --no-power-probe > $d/bench_config2.txt 2> $d/bench_config2.err; show $d/bench_config2.txt "config 2"; tail -2 $d/bench_config2.err
timeout 500 python bench.py --config 2 --recompute --steps 6 --warmup 2 --no-cpu-baseline --no-power-probe > $d/bench_config2_full.txt 2> $d/bench_config2_full.err; show $d/bench_config2_full.txt "config 2 full recompute"
