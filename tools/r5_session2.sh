#!/bin/bash
# Round 5, second session: every gpurun call of the session as one function (provenance of profiles/r5_experiments.md sections 8-12).
#   gpurun -- "bash tools/r5_session2.sh <step>"      steps: call6 call7 call8 call9 call10 call11 call12 call13 call14 call15 call16
# A/B steps against another commit of the host side need tools/ab_tree.sh <commit> first (ab_prev/).
R=$GRAFT_REPO_ROOT

call6() {
  # round 5, call 6 (second session): the cheap recompute level, return_all_hiddens / layerdrop on the fused path, ln_geglu_bwd without its second
  # exponential -- tests, kernel A/B against the previous layernorm.hip (lib/libonepeace_hip_lnold.so, built by tools/build_variant.py from
  # git show 60d4d92:one-peace_amd/csrc/layernorm.hip), whole-step A/B, the cheap level's cost, config 2 at the cheap level
  d=$R/gpurun_out/r5c6; mkdir -p $d
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
}

call7() {
  # round 5, call 7: the whole GPU suite at HEAD (timed), smoke, and what the one-tile-per-workgroup rule of world > 1 costs on an idle GPU
  d=$R/gpurun_out/r5c7; mkdir -p $d
  cd $R
  ( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $d/pytest.txt 2>&1; tail -8 $d/pytest.txt
  timeout 300 python __graft_entry__.py smoke > $d/smoke.txt 2>&1; tail -2 $d/smoke.txt
  B="--steps 6 --warmup 2 --no-cpu-baseline --no-power-probe --no-skip-leg"
  show() { tail -1 $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print('$2', round(d['ms_per_step'],1), round(d['value'],1), 'gemm', round(r.get('frac',0),4))" || tail -5 $1; }
  for v in 0 7 0 7; do
    ONEPEACE_TUNE_SCHED=$v timeout 400 python bench.py $B > $d/bench_sched${v}_$(date +%s).txt 2> $d/bench_sched$v.err; show $(ls -t $d/bench_sched${v}_*.txt | head -1) "headline sched $v"
  done
}

call8() {
  # round 5, call 8: layer-scale gradient from the weight gradient (no branch output y): new op tests, model tests that touch the branch functions, A/B of the step
  d=$R/gpurun_out/r5c8; mkdir -p $d
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
}

call9() {
  d=$R/gpurun_out/r5c9; mkdir -p $d
  cd $R
  timeout 600 python tools/torch_prof_aten.py 64 > $d/aten.txt 2>&1; grep -v amdgpu.ids $d/aten.txt | tail -75
  timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "recompute_cheap" 2>&1 | tail -2
}

call10() {
  # round 5, call 10: the whole GPU suite + smoke at HEAD (layer-scale gradient from the weight gradient on by default)
  d=$R/gpurun_out/r5c10; mkdir -p $d
  cd $R
  ( time timeout 2000 python -m pytest tests -m gpu -q ) > $d/pytest.txt 2>&1; tail -12 $d/pytest.txt
  timeout 300 python __graft_entry__.py smoke > $d/smoke.txt 2>&1; tail -2 $d/smoke.txt
}

call11() {
  # round 5, call 11: slice copies instead of torch.cat in the adapters / the lock-step packing, one gradient placeholder per bias handle: model tests, bench
  d=$R/gpurun_out/r5c11; mkdir -p $d
  cd $R
  timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_distributed_gpu.py -m gpu -x -q > $d/pytest_model.txt 2>&1; tail -4 $d/pytest_model.txt
  timeout 300 python __graft_entry__.py smoke > $d/smoke.txt 2>&1; tail -1 $d/smoke.txt
  B="--steps 6 --warmup 2 --no-cpu-baseline --no-power-probe --no-skip-leg"
  for i in 1 2; do
    timeout 400 python bench.py $B > $d/bench_$i.txt 2> $d/bench_$i.err; tail -1 $d/bench_$i.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print('headline', round(d['ms_per_step'],1), round(d['value'],1), 'gemm', round(r.get('frac',0),4), 'loss', d['config'].get('final_loss'))" || tail -5 $d/bench_$i.err
  done
}

call12() {
  # round 5, call 12: same-box A/B of the host-side cleanups (torch.cat -> slice copies, one bias-gradient placeholder) against the commit before them
  d=$R/gpurun_out/r5c12; mkdir -p $d
  B="--steps 8 --warmup 2 --no-cpu-baseline --no-power-probe --no-skip-leg"
  for v in prev head prev head; do
    if [ $v = prev ]; then cd $R/ab_prev; else cd $R; fi
    timeout 400 python bench.py $B > $d/bench_${v}_$(date +%s).txt 2> $d/bench_$v.err; tail -1 $(ls -t $d/bench_${v}_*.txt | head -1) | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print('headline $v', round(d['ms_per_step'],1), round(d['value'],1), 'gemm', round(r.get('frac',0),4), 'loss', d['config'].get('final_loss'))" || tail -5 $d/bench_$v.err
  done
  cd $R; timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "layer_scale_gradient" 2>&1 | tail -2
}

call13() {
  d=$R/gpurun_out/r5c13; mkdir -p $d
  cd $R
  for c in full elementwise gemm; do timeout 500 python tools/overlap_probe.py --chain $c 2>&1 | grep -v amdgpu.ids | tee -a $d/overlap.txt; done
}

call14() {
  # round 5, call 14: PMC traffic of the elementwise kernels at HEAD (resid_bwd no longer reads y) + whole GPU suite at HEAD
  d=$R/gpurun_out/r5c14; mkdir -p $d
  cd $R
  timeout 900 bash tools/pmc_elementwise_traffic.sh $d/r5_elementwise_traffic.txt > $d/pmc_elem.log 2>&1; cat $d/r5_elementwise_traffic.txt || tail -5 $d/pmc_elem.log
  ( time timeout 2000 python -m pytest tests -m gpu -q ) > $d/pytest.txt 2>&1; tail -6 $d/pytest.txt
}

call15() {
  # round 5, call 15: op_gemm_nt_batched (the 16 groups of the audio positional convolution as one launch): tests, audio model tests, same-box A/B of the step
  d=$R/gpurun_out/r5c15; mkdir -p $d
  cd $R
  timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "batched or audio or conv or gemm_nt or gemm_full" > $d/pytest_ops.txt 2>&1; tail -3 $d/pytest_ops.txt
  timeout 1200 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "audio or micro or lock_step_pass_matches or deep_text" > $d/pytest_model.txt 2>&1; tail -3 $d/pytest_model.txt
  B="--steps 8 --warmup 2 --no-cpu-baseline --no-power-probe --no-skip-leg"
  for v in prev head prev head; do
    if [ $v = prev ]; then cd $R/ab_prev; else cd $R; fi
    timeout 400 python bench.py $B > $d/bench_${v}_$(date +%s).txt 2> $d/bench_$v.err; tail -1 $(ls -t $d/bench_${v}_*.txt | head -1) | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print('headline $v', round(d['ms_per_step'],1), round(d['value'],1), 'gemm', round(r.get('frac',0),4), 'launches', r.get('launches'), 'loss', d['config'].get('final_loss'))" || tail -5 $d/bench_$v.err
  done
}

call16() {
  # round 5, call 16: the qkv launch split so that its 256 x 256 tiles are whole rounds (ONEPEACE_QKV_ROUND_SPLIT=1): same-box A/B
  d=$R/gpurun_out/r5c16; mkdir -p $d
  cd $R
  B="--steps 8 --warmup 2 --no-cpu-baseline --no-power-probe --no-skip-leg"
  for v in 0 1 0 1; do
    ONEPEACE_QKV_ROUND_SPLIT=$v timeout 400 python bench.py $B > $d/bench_${v}_$(date +%s).txt 2> $d/bench_$v.err; tail -1 $(ls -t $d/bench_${v}_*.txt | head -1) | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print('headline round-split=$v', round(d['ms_per_step'],1), round(d['value'],1), 'gemm', round(r.get('frac',0),4), 'launches', r.get('launches'), 'loss', d['config'].get('final_loss'))" || tail -5 $d/bench_$v.err
  done
}

call17() {
  # round 5, call 17: first block of the audio feature extractor fused from the waveform: tests, audio model tests, same-box A/B (ONEPEACE_FUSED_CONV1)
  d=$R/gpurun_out/r5c17; mkdir -p $d
  cd $R
  timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "audio" > $d/pytest_ops.txt 2>&1; tail -3 $d/pytest_ops.txt
  timeout 1200 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "audio or micro or deep_text or pretrain_al or stage2" > $d/pytest_model.txt 2>&1; tail -3 $d/pytest_model.txt
  timeout 300 python __graft_entry__.py smoke > $d/smoke.txt 2>&1; tail -1 $d/smoke.txt
  B="--steps 8 --warmup 2 --no-cpu-baseline --no-power-probe --no-skip-leg"
  for v in 0 1 0 1; do
    ONEPEACE_FUSED_CONV1=$v timeout 400 python bench.py $B > $d/bench_${v}_$(date +%s).txt 2> $d/bench_$v.err; tail -1 $(ls -t $d/bench_${v}_*.txt | head -1) | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; m=d['config'].get('memory') or {}; print('headline fused-conv1=$v', round(d['ms_per_step'],1), round(d['value'],1), 'gemm', round(r.get('frac',0),4), 'peak GB', m.get('peak_reserved_gb'), 'loss', d['config'].get('final_loss'))" || tail -5 $d/bench_$v.err
  done
}

"$@"
