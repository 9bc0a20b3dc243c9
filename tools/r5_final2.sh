#!/bin/bash
# round 5, second evidence call (after the bit_cast fix of the scaled epilogue and the algorithmic-bytes snapshot in bench.py): whole GPU
# suite + smoke, the default bench line, fp8 kernel rates, the config-4 pair with final losses
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r5final2}; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 300 python __graft_entry__.py smoke > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 600 python bench.py > $O/bench_default.txt 2> $O/bench_default.err
tail -1 $O/bench_default.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], d['value'], r['frac'], r['traffic'], r['algorithmic_bytes_per_launch'], d['cpu_baseline']['value'], d['skip_dropped_branches'].get('ms_per_step'))" || tail -5 $O/bench_default.err
timeout 300 python tools/fp8_bench.py > $O/fp8_bench.txt 2>&1; grep -v amdgpu.ids $O/fp8_bench.txt
for v in bf16 fp8 bf16 fp8; do
  extra=""; [ $v = fp8 ] && extra="--fp8"
  timeout 500 python bench.py --config 4 --steps 20 --warmup 3 --no-cpu-baseline --no-power-probe --no-skip-leg $extra > $O/bench_config4_$v.txt 2>&1
  tail -1 $O/bench_config4_$v.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print('config 4 $v', d['ms_per_step'], d['value'], r.get('fp8_gemm'), d['config'].get('final_loss'))" || tail -5 $O/bench_config4_$v.txt
done
