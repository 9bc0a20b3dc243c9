#!/bin/bash
# ONE parameterised gpurun call script (replaces the round-stamped r2_* ... r5_* one-shots; their results live in profiles/):
#   gpurun --timeout S -- 'bash tools/gpu_session.sh <tag> <stage> [<stage> ...]'
# Output goes to gpurun_out/<tag>/ (copy what is to be judged into profiles/ afterwards).  Stages:
#   tests-new     the GPU tests selected by -k "$NEW_K" (default: this round's new / changed ones), all of them (no -x) -- fast feedback
#   tests         the whole `-m gpu` suite;  smoke   __graft_entry__.smoke()
#   bench         default `python bench.py` (cpu_baseline, power probe, skip leg);  bench-short  8 steps, no extras
#   bench-extra   --audio-seconds 15, --objective pretrain-vl, --objective pretrain-al lines
#   configs       --config 1 / 2 / 4 / 4 --fp8 lines
#   trace         rocprofv3 --kernel-trace of the headline step + last-step summary (tools/trace_summary.py)
#   trace-cfg4    the same for --config 4;  trace-skip  the same for the headline step with --skip-dropped
#   traffic       FETCH_SIZE / WRITE_SIZE of the GEMM family over a bench run (tools/pmc_bench_traffic.sh)
#   pmc-attn      PMC passes over the attention kernels at S = 257 / B = 128 and S = 785 / B = 64 (tools/pmc_attn.sh)
#   fp8-ab        config 4: bf16 / fp8 forward only / fp8 forward + input gradients, alternating on one box, with loss curves
#   contention    tools/cu_contention_ab.py;  blas   tools/blas_compare.py;  attn-bench  tools/attn_bench.py
#   ab:<lib>      whole-step A/B (2 x 2 alternating runs) of the in-tree library against one-peace_amd/lib/<lib> (tools/build_variant.py)
R=$GRAFT_REPO_ROOT; TAG=${1:-session}; shift
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
NEW_K=${NEW_K:-"row_dot or gamma_grad or flat_parameters or audio_length_15s or all_hiddens or recompute_cheap or layer_scale_gradient or weight_cache_refresh or one_tile_launch_rule or layer_4b"}
line() { tail -1 $1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; c=d['config']
print('$2', round(d['ms_per_step'],1), 'ms', round(d['value'],2), d['unit'], 'frac', r.get('frac'), 'plp', (r.get('power_limited_peak') or {}).get('tflops') if isinstance(r.get('power_limited_peak'), dict) else r.get('power_limited_peak'), 'batch', c.get('per_gpu_batch'), 'loss', c.get('final_loss'), 'mem', (c.get('memory') or {}).get('peak_reserved_gb'), 'skip', (d.get('skip_dropped_branches') or {}).get('ms_per_step'))" 2>/dev/null || tail -5 ${1%.txt}.err; }
for stage in "$@"; do
  echo "=== $stage"
  case $stage in
    tests-new) timeout 1500 python -m pytest tests -m gpu -q -k "$NEW_K" > $O/pytest_new.txt 2>&1; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_new.txt | tail -30 ;;
    tests) timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -8 $O/pytest.txt ;;
    smoke) timeout 300 python __graft_entry__.py smoke > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt ;;
    bench) timeout 700 python bench.py > $O/bench_default.txt 2> $O/bench_default.err; line $O/bench_default.txt default ;;
    bench-short) timeout 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-power-probe --no-skip-leg > $O/bench_short.txt 2> $O/bench_short.err; line $O/bench_short.txt short ;;
    bench-extra)
      timeout 600 python bench.py --audio-seconds 15 --steps 8 --warmup 3 --no-cpu-baseline --no-power-probe --no-skip-leg > $O/bench_audio15.txt 2> $O/bench_audio15.err; line $O/bench_audio15.txt audio15
      timeout 600 python bench.py --objective pretrain-vl --steps 8 --warmup 3 --no-cpu-baseline --no-power-probe > $O/bench_pretrain_vl.txt 2> $O/bench_pretrain_vl.err; line $O/bench_pretrain_vl.txt pretrain-vl
      timeout 600 python bench.py --objective pretrain-al --audio-seconds 15 --steps 8 --warmup 3 --no-cpu-baseline --no-power-probe > $O/bench_pretrain_al.txt 2> $O/bench_pretrain_al.err; line $O/bench_pretrain_al.txt pretrain-al ;;
    configs)
      for c in 1 2 4; do timeout 500 python bench.py --config $c --steps 8 --warmup 3 --no-cpu-baseline --no-power-probe --no-skip-leg > $O/bench_config$c.txt 2> $O/bench_config$c.err; line $O/bench_config$c.txt config$c; done
      timeout 500 python bench.py --config 4 --fp8 --steps 8 --warmup 3 --no-cpu-baseline --no-power-probe --no-skip-leg > $O/bench_config4_fp8.txt 2> $O/bench_config4_fp8.err; line $O/bench_config4_fp8.txt config4-fp8 ;;
    fp8-ab)  # config 4 on ONE box, 20 timed steps with the loss of every step: bf16, fp8 forward only (round 5), fp8 forward + input gradients
      for v in bf16 fp8fwd fp8 bf16 fp8fwd fp8; do
        case $v in bf16) fl="";; fp8fwd) fl="--fp8 --fp8-forward-only";; fp8) fl="--fp8";; esac
        n=$(ls $O | grep -c "^fp8ab_${v}_.*txt")
        timeout 500 python bench.py --config 4 $fl --steps 20 --warmup 3 --loss-curve --no-cpu-baseline --no-power-probe --no-skip-leg > $O/fp8ab_${v}_$n.txt 2> $O/fp8ab_${v}_$n.err
        line $O/fp8ab_${v}_$n.txt "cfg4-$v"
      done ;;
    trace|trace-cfg4|trace-skip)
      extra=""; name=bench_last_step; [ $stage = trace-cfg4 ] && { extra="--config 4"; name=bench_config4_last_step; }
      [ $stage = trace-skip ] && { extra="--skip-dropped"; name=bench_skip_dropped_last_step; }
      ( cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_$TAG
        timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- python $R/bench.py $extra --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-power-probe --no-skip-leg > $O/${name}_under_rocprof.json 2> $O/${name}_under_rocprof.err
        KT=$(find /tmp/prof_$TAG -name "*kernel_trace.csv" | head -1); ST=$(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1)
        cp $ST $O/${name}_kernel_stats.csv
        python $R/tools/trace_summary.py $KT $O/$name.json 1 > $O/$name.txt 2>&1 )
      head -45 $O/$name.txt | cut -c1-160 ;;
    traffic) timeout 900 bash tools/pmc_bench_traffic.sh $O/gemm_hbm_traffic.json > $O/pmc_traffic.log 2>&1
      python -c "
import json; d=json.load(open('$O/gemm_hbm_traffic.json')); print('traffic per launch', d['bytes_per_launch'], {k: round(v/1e9,2) for k,v in d['by_kernel_read_bytes_per_launch'].items()}, d['by_kernel_launches'])" || tail -5 $O/pmc_traffic.log ;;
    pmc-attn) timeout 700 bash tools/pmc_attn.sh 257 128 $O/pmc_attention_S257_B128.txt > /dev/null 2>&1; timeout 700 bash tools/pmc_attn.sh 785 64 $O/pmc_attention_S785_B64.txt > /dev/null 2>&1
      grep -E "SQ_WAIT_ANY|SQ_WAVE_CYCLES|SQ_INSTS_MFMA|SQ_INSTS_VALU |SQ_ACTIVE_INST_VALU" $O/pmc_attention_S257_B128.txt $O/pmc_attention_S785_B64.txt | cut -c1-170 ;;
    contention) timeout 600 python tools/cu_contention_ab.py > $O/cu_contention_ab.txt 2>&1; cat $O/cu_contention_ab.txt ;;
    blas) ITERS=30 ROUNDS=3 timeout 400 python tools/blas_compare.py > $O/blas_compare.txt 2>&1; grep -v amdgpu.ids $O/blas_compare.txt | tail -22 ;;
    attn-bench) timeout 400 python tools/attn_bench.py > $O/attn_bench.txt 2>&1; tail -30 $O/attn_bench.txt ;;
    ab:*) V=$R/one-peace_amd/lib/${stage#ab:}
      for leg in new old new old; do
        [ $leg = old ] && export ONEPEACE_HIP_LIB=$V || unset ONEPEACE_HIP_LIB
        n=$(ls $O | grep -c "^ab_$leg")
        timeout 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-power-probe --no-skip-leg > $O/ab_${leg}_$n.txt 2> $O/ab_${leg}_$n.err; line $O/ab_${leg}_$n.txt "ab-$leg"
      done; unset ONEPEACE_HIP_LIB ;;
    *) echo "unknown stage $stage" ;;
  esac
done
