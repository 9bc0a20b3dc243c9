#!/bin/bash
# round 4, call 11: probe test, grouped parity again (final library), default bench line with the power-limited peak + cpu baseline
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c11; mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_probes_gpu.py tests/test_ops_gpu.py -x -q -k "mfma_rate or tn_grouped or headline_layer_grouped" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 900 python bench.py > $O/bench_default.txt 2>&1; tail -1 $O/bench_default.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], json.dumps(d['roofline'])[:1500]); print(d.get('cpu_baseline'))" || tail -5 $O/bench_default.txt
