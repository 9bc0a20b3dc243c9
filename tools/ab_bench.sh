#!/bin/bash
# A/B of op_gemm_set_tile knob values on the whole bench inside one GPU session:  tools/ab_bench.sh 44 40 44 40
for k in "$@"; do
  python tools/bench_knob.py $k -- --steps 4 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 > /tmp/ab_$k.json
  python - $k <<'PY'
import json, sys
d = json.load(open("/tmp/ab_%s.json" % sys.argv[1]))
print("knob", sys.argv[1], "%.2f samples/s" % d["value"], "%.1f ms" % d["ms_per_step"], "GEMM %.0f TF" % d["roofline"]["achieved"])
PY
done
