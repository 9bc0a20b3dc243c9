#!/bin/bash
# HBM-side traffic of the GEMM family at the headline batch from TWO shallow models (4 and 8 layers; every layer issues the
# same launches), one step each, FETCH_SIZE and WRITE_SIZE in separate --pmc passes restricted to the GEMM kernels:
#   per-layer traffic = (8-layer - 4-layer) / 4;   40-layer step = 4-layer + 36 x per-layer.
# (The full 40-layer command under --pmc takes > 5 box-minutes per pass.)   tools/r2_pmc_slice.sh <out.json>
out=$1; R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  for L in 4 8; do
    rm -rf /tmp/pmcs_${c}_$L
    timeout 300 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "gemm|splitk" --output-format csv -d /tmp/pmcs_${c}_$L -o p -- \
      python $R/bench.py --layers $L --steps 1 --warmup 1 --no-profile --no-cpu-baseline > /tmp/pmcs_${c}_$L.log 2>&1
  done
done
python - "$out" <<'PY'
import csv, glob, json, sys
def load(d):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    tot, n = 0.0, 0
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm" in k or "splitk" in k:
            tot += float(r["Counter_Value"]); n += ("gemm" in k)
    return tot, n
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    t4, n4 = load("/tmp/pmcs_%s_4" % c); t8, n8 = load("/tmp/pmcs_%s_8" % c)
    res[c] = {"kb_4_layers_2_steps": t4, "gemm_launches_4": n4, "kb_8_layers_2_steps": t8, "gemm_launches_8": n8}
json.dump(res, open(sys.argv[1], "w"), indent=1)
print(json.dumps(res, indent=1))
PY
