#!/bin/bash
# round 4, call 23: bench lines of the other BASELINE configurations at HEAD
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c23; mkdir -p $O
cd $R
for c in 1 2 4; do
  timeout 400 python bench.py --config $c --steps 4 --warmup 1 --no-cpu-baseline --no-power-probe > $O/r4_bench_config${c}_1gpu.json 2> $O/bench_config$c.err
  tail -1 $O/r4_bench_config${c}_1gpu.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config $c', d['ms_per_step'], d['value'], d['unit'], d['roofline']['frac'] if d.get('roofline') else None)" || tail -3 $O/bench_config$c.err
done
timeout 400 python bench.py --config 4 --fp8 --steps 4 --warmup 1 --no-cpu-baseline --no-power-probe > $O/r4_bench_config4_fp8_1gpu.json 2> $O/bench_config4fp8.err
tail -1 $O/r4_bench_config4_fp8_1gpu.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config 4 fp8', d['ms_per_step'], d['value'])" || tail -3 $O/bench_config4fp8.err
timeout 400 python bench.py --objective pretrain-vl --steps 4 --warmup 1 --no-cpu-baseline --no-power-probe > $O/r4_bench_pretrain_vl_1gpu.json 2> $O/bench_pvl.err
tail -1 $O/r4_bench_pretrain_vl_1gpu.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pretrain-vl', d['ms_per_step'], d['value'])" || tail -3 $O/bench_pvl.err
